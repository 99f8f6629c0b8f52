cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ab
for dbg in 0 5 0 5; do
ADAC_TUNING=sel_debug=$dbg timeout -k 10 200 python3 tools/pmc_probe.py select u64:8,u32:8,u16:8,u16:12,u8:4,u8:6 0 10 > gpurun_out/r02ab/sel_dbg$dbg.json 2>> gpurun_out/r02ab/err.txt
python3 - <<PY
import json
print('dbg$dbg', ' '.join('%s:%d=%.0f'%(x['dtype'],x['width'],x['select_read_GBps']) for x in json.load(open('gpurun_out/r02ab/sel_dbg$dbg.json'))))
PY
done
