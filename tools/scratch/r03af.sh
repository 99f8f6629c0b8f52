#!/bin/bash
# (scratch: masked scans g9 vs g8 + whole GPU suite)
set -o pipefail
O=gpurun_out/r03af
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
AB_VALID=0.9 AB_KNOBS="scan_cells=1" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u64:8,u32:13,u16:12,u8:3 3 10 > $O/g9.json 2> $O/g9.err
ADAC_LIB=$PWD/duckdb-adaptive-compression_amd/build/libadacodec_g8.so AB_VALID=0.9 AB_KNOBS="scan_cells=1" timeout -k 10 400 python3 tools/ab_select.py c2,u64:16,u64:8,u32:13,u16:12,u8:3 3 10 > $O/g8.json 2> $O/g8.err
tail -n 3 $O/g9.err
