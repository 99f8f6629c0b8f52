cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02h
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_repack.py tests/test_gpu_c2_c4.py tests/test_gpu_persistence.py -x -q -m gpu > gpurun_out/r02h/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02h/tests.log
for k in 1 0; do ADAC_TUNING=single_pass_encode=$k timeout -k 10 300 python3 tools/pmc_probe.py encode u64:32,u64:16,u64:8,u32:16,u32:8 100000000 20 > gpurun_out/r02h/enc_$k.json 2> gpurun_out/r02h/enc_$k.err; done
timeout -k 10 300 python3 tools/pmc_probe.py repack u64:8,u64:13,u64:16,u64:20,u64:24,u32:8,u32:13,u32:16,u32:20,u32:24 0 10 > gpurun_out/r02h/repack.json 2> gpurun_out/r02h/repack.err
echo done
