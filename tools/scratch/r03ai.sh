#!/bin/bash
# (scratch: masked SUM g12 vs g11)
O=gpurun_out/r03ai
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
G12=$PWD/duckdb-adaptive-compression_amd/build/libadacodec_g12.so
ADAC_LIB=$G12 timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_select.py tests/test_gpu_configs.py -x -q > $O/tests.log 2>&1; tail -n 3 $O/tests.log
ADAC_LIB=$G12 AB_VALID=0.9 AB_KNOBS="scan_cells=1" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u64:8,u32:13,u16:12,u8:6 3 10 > $O/g12.json 2> $O/g12.err
AB_VALID=0.9 AB_KNOBS="scan_cells=1" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u64:8,u32:13,u16:12,u8:6 3 10 > $O/g11.json 2> $O/g11.err
ADAC_LIB=$G12 timeout -k 10 300 python3 tools/soak_fuzz.py 5000000 2000 > $O/soak.log 2>&1; tail -n 1 $O/soak.log
tail -n 3 $O/g12.err
