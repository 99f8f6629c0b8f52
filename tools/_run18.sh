cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02h
timeout -k 10 120 python3 tools/encode_stamps.py u64:32 100000000 > gpurun_out/r02h/stamps_u64_32.json 2> gpurun_out/r02h/err.txt || { tail -3 gpurun_out/r02h/err.txt; exit 1; }
timeout -k 10 120 python3 tools/encode_stamps.py u32:8 100000000 > gpurun_out/r02h/stamps_u32_8.json 2>> gpurun_out/r02h/err.txt || exit 1
python3 - <<'PY'
import json
for f in ['stamps_u64_32','stamps_u32_8']:
    d=json.load(open('gpurun_out/r02h/%s.json'%f))
    print(f, d['kernel_span_us'], d['wg_lifetime_us']['mean'], d['phase_us_mean'], d['lookback_only_us_mean'], d['barrier_after_lookback_us_mean'])
PY
