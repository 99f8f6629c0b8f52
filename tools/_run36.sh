cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02aa
timeout -k 10 400 python -m pytest tests/test_gpu_select.py tests/test_gpu_parity.py tests/test_gpu_c2_c4.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r02aa/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02aa/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/pmc_probe.py select u64:8,u64:13,u64:16,u64:20,u64:24,u64:32,u32:8,u32:13,u32:16,u32:24,u16:8,u16:12,u8:4,u8:6 0 10 > gpurun_out/r02aa/select.json 2> gpurun_out/r02aa/select.err || exit 1
python3 - <<'PY'
import json
for x in json.load(open('gpurun_out/r02aa/select.json')):
    print(x['dtype'],x['width'],'select %.0f GB/s'%(x['select_read_GBps']))
PY
