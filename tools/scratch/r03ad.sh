#!/bin/bash
# (scratch: one GPU session of round 3 — whole GPU suite, gather g8 vs g7, driver bench)
set -o pipefail
O=gpurun_out/r03ad
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
AB_KNOBS="gather_compact=0|gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u32:24,u32:13,u16:12,u8:6 4 20 > $O/g8.json 2> $O/g8.err
ADAC_LIB=$PWD/duckdb-adaptive-compression_amd/build/libadacodec_g7.so AB_KNOBS="gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u16:12,u8:6 4 20 > $O/g7.json 2> $O/g7.err
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err
tail -n 3 $O/g8.err $O/bench.err
