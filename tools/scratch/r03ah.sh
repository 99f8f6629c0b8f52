#!/bin/bash
# (scratch: whole GPU suite on g11 (empty-tile exit in the gather, new BITPACKING fixtures) + clustered selection A/B)
O=gpurun_out/r03ah
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -n 5 $O/tests.log
AB_CLUSTERED=0.1 AB_KNOBS="gather_compact=0|gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u32:16,u16:12 3 20 > $O/g11.json 2> $O/g11.err
ADAC_LIB=$PWD/duckdb-adaptive-compression_amd/build/libadacodec_g10.so AB_CLUSTERED=0.1 AB_KNOBS="gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2,u32:16,u16:12 3 20 > $O/g10.json 2> $O/g10.err
AB_SELECTIVITY=0.01 AB_KNOBS="gather_compact=0|gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2 3 20 > $O/g11_sel001.json 2> $O/g11_sel001.err
AB_SELECTIVITY=0.9 AB_KNOBS="gather_compact=0|gather_compact=3" timeout -k 10 400 python3 tools/ab_select.py c2 3 20 > $O/g11_sel09.json 2> $O/g11_sel09.err
tail -n 3 $O/g11.err $O/g11_sel001.err
