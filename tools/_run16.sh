cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02e
timeout -k 10 120 python3 tools/encode_stamps.py u64:32 100000000 2 > gpurun_out/r02e/stamps_u64_32_p2.json 2> gpurun_out/r02e/err.txt || { tail -5 gpurun_out/r02e/err.txt; exit 1; }
timeout -k 10 120 python3 tools/encode_stamps.py u64:32 100000000 1 > gpurun_out/r02e/stamps_u64_32_p1.json 2>> gpurun_out/r02e/err.txt || exit 1
timeout -k 10 120 python3 tools/encode_stamps.py u64:8 100000000 1 > gpurun_out/r02e/stamps_u64_8_p1.json 2>> gpurun_out/r02e/err.txt || exit 1
cat gpurun_out/r02e/stamps_u64_32_p2.json gpurun_out/r02e/stamps_u64_32_p1.json
