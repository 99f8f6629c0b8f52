cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02g
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c2_c4.py tests/test_gpu_persistence.py -x -q -m gpu > gpurun_out/r02g/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 gpurun_out/r02g/tests.log
[ $rc -eq 0 ] || exit 1
CASES=u64:32,u64:16,u64:8,u64:13,u32:16,u32:8,u32:24,u32:13
timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02g/encode.json 2> gpurun_out/r02g/encode.err || exit 1
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r02g/encode.json'))
for x in a: print(x['dtype'],x['width'],'%.4f ms'%(x['ms']['encode']))
PY
timeout -k 10 120 python3 tools/encode_stamps.py u64:32 100000000 > gpurun_out/r02g/stamps_u64_32.json 2>> gpurun_out/r02g/encode.err
cat gpurun_out/r02g/stamps_u64_32.json
echo done
