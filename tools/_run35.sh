cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02z
for dbg in 0 3 4; do
ADAC_TUNING=sel_debug=$dbg timeout -k 10 200 python3 tools/pmc_probe.py select u64:8,u64:16,u64:32,u32:8 0 10 > gpurun_out/r02z/sel2_dbg$dbg.json 2>> gpurun_out/r02z/err.txt
python3 - <<PY
import json
for x in json.load(open('gpurun_out/r02z/sel2_dbg$dbg.json')):
    print('dbg$dbg', x['dtype'],x['width'],'select %.0f'%(x['select_read_GBps']))
PY
done
