cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02k2
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_c2_c4.py tests/test_gpu_persistence.py -x -q -m gpu > gpurun_out/r02k2/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 gpurun_out/r02k2/tests.log
[ $rc -eq 0 ] || exit 1
CASES=u64:32,u64:16,u64:8,u64:13,u32:16,u32:8,u32:24,u32:13
timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02k2/encode_ordered.json 2> gpurun_out/r02k2/encode.err || exit 1
ADAC_TUNING=encode_placement=1 timeout -k 10 200 python3 tools/pmc_probe.py encode $CASES 100000000 20 > gpurun_out/r02k2/encode_firstcome.json 2>> gpurun_out/r02k2/encode.err || exit 1
python3 - <<'PY'
import json
a=json.load(open('gpurun_out/r02k2/encode_ordered.json')); b=json.load(open('gpurun_out/r02k2/encode_firstcome.json'))
for x,y in zip(a,b): print(x['dtype'],x['width'],'ordered %.4f ms  first-come %.4f ms'%(x['ms']['encode'],y['ms']['encode']))
PY
timeout -k 10 120 python3 tools/encode_stamps.py u64:32 100000000 > gpurun_out/r02k2/stamps_u64_32.json 2>> gpurun_out/r02k2/encode.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r02k2/stamps_u64_32.json'))
print(d['kernel_span_us'], d['wg_lifetime_us']['mean'], d['phase_us_mean'], d['lookback_only_us_mean'], d['barrier_after_lookback_us_mean'])
PY
echo done
