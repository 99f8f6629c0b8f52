cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02k
timeout -k 10 600 python -m pytest tests/test_gpu_repack.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02k/tests.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r02k/tests.log
timeout -k 10 300 python3 tools/pmc_probe.py repack u64:8,u64:13,u64:16,u64:20,u64:24,u64:32,u32:8,u32:13,u32:16,u32:20,u32:24,u16:8,u16:12,u8:4,u8:6 0 10 > gpurun_out/r02k/repack.json 2> gpurun_out/r02k/repack.err
echo done
