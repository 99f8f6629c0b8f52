cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02i
CASES=u64:32,u64:16,u64:8,u32:16,u32:8,u32:13
ADAC_TUNING=encode_stamps=2 timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02i/encode_firstcome.json 2> gpurun_out/r02i/encode.err || exit 1
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r02i/encode_firstcome.json'))
for x in a: print(x['dtype'],x['width'],'%.4f ms'%(x['ms']['encode']))
PY
