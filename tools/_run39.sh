cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ad
timeout -k 10 400 python -m pytest tests/test_gpu_select.py -x -q -m gpu > gpurun_out/r02ad/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/r02ad/tests.log
