#!/bin/bash
# (scratch: one GPU session of round 3 — gather g7 vs g6, SQ counters)
set -o pipefail
O=gpurun_out/r03ac
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_select.py tests/test_gpu_fuzz.py tests/test_gpu_c2_c4.py -x -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
AB_KNOBS="gather_compact=0|gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u32:24,u32:13,u16:12,u8:6 4 20 > $O/g7.json 2> $O/g7.err
ADAC_LIB=$PWD/duckdb-adaptive-compression_amd/build/libadacodec_g6.so AB_KNOBS="gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u32:24,u32:13,u16:12,u8:6 4 20 > $O/g6.json 2> $O/g6.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $O/pmc1 --output-format csv -- python3 tools/pmc_probe.py unpack,gather u64:32 100000000 3 > $O/pmc1.json 2> $O/pmc1.err
python3 tools/pmc_sum.py $O/pmc1 k_ > $O/pmc1.txt; rm -rf $O/pmc1
tail -n 3 $O/g7.err
