cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02e
timeout -k 10 600 python -m pytest tests/test_gpu_host_mirror.py -x -q -m gpu -s -k "destroyed or exhaustion or pools or checkpoint" > gpurun_out/r02e/tests_a.log 2>&1; echo "tests_a rc=$?"
tail -5 gpurun_out/r02e/tests_a.log
timeout -k 10 900 python -m pytest tests/test_gpu_host_mirror.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r02e/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02e/tests.log
