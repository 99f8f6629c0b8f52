cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02d
timeout -k 10 600 python -m pytest tests/test_gpu_select.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r02d/tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r02d/tests.log
python3 tools/pmc_probe.py select,count u64:8,u64:13,u64:16,u64:20,u64:32,u32:8,u32:16,u32:24 0 10 > gpurun_out/r02d/sel_asm.json 2> gpurun_out/r02d/sel_asm.err
python3 tools/pmc_probe.py select,count u16:5,u16:8,u16:12,u8:3,u8:4,u8:6 0 10 > gpurun_out/r02d/sel_narrow.json 2> gpurun_out/r02d/sel_narrow.err
echo done
