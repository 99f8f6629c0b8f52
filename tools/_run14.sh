cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02c
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c2_c4.py tests/test_gpu_persistence.py -x -q -m gpu > gpurun_out/r02c/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 gpurun_out/r02c/tests.log
[ $rc -eq 0 ] && timeout -k 10 300 python3 tools/pmc_probe.py encode u64:32,u64:16,u64:8,u64:13,u32:16,u32:8,u32:24,u32:13 100000000 20 > gpurun_out/r02c/encode.json 2> gpurun_out/r02c/encode.err
cat gpurun_out/r02c/encode.json
echo done
