cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ae
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02ae/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02ae/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/pmc_probe.py select,count,sum u64:8,u64:13,u64:16,u64:24,u64:32,u32:8,u32:13,u32:16,u32:24,u16:8,u8:4,u8:6 0 10 > gpurun_out/r02ae/scan.json 2> gpurun_out/r02ae/scan.err || exit 1
python3 - <<'PY'
import json
for x in json.load(open('gpurun_out/r02ae/scan.json')):
    pb=x['packed_bytes']
    print(x['dtype'],x['width'],'select %.0f  count %.0f  sum %.0f'%(x['select_read_GBps'], pb/x['ms']['count']/1e6, pb/x['ms']['sum']/1e6))
PY
