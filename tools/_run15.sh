cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02d
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c2_c4.py tests/test_gpu_persistence.py -x -q -m gpu > gpurun_out/r02d/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 gpurun_out/r02d/tests.log
[ $rc -eq 0 ] || exit 1
CASES=u64:32,u64:16,u64:8,u64:13,u32:16,u32:8,u32:24,u32:13
timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02d/encode_parts2.json 2> gpurun_out/r02d/encode.err || exit 1
ADAC_TUNING=encode_parts=1 timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02d/encode_parts1.json 2>> gpurun_out/r02d/encode.err || exit 1
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r02d/encode_parts2.json')); b=json.load(open('gpurun_out/r02d/encode_parts1.json'))
for x,y in zip(a,b): print(x['dtype'],x['width'],'parts2 %.4f ms  parts1 %.4f ms'%(x['ms']['encode'],y['ms']['encode']))
PY
echo done
