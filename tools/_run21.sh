cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02l
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mixes_its_flows" > gpurun_out/r02l/mixed.log 2>&1; rc=$?; echo "mixed rc=$rc"; tail -15 gpurun_out/r02l/mixed.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02l/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r02l/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/soak_fuzz.py 500000 3000 > gpurun_out/r02l/soak_fuzz.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r02l/soak_fuzz.log
