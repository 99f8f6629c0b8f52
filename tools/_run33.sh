cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02x
timeout -k 10 300 python -m pytest tests/test_gpu_group_sum.py -x -q -m gpu > gpurun_out/r02x/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02x/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/soak_group_sum.py > gpurun_out/r02x/soak.log 2>&1; tail -3 gpurun_out/r02x/soak.log
timeout -k 10 500 python3 bench_configs.py q1_packed 2> gpurun_out/r02x/q1.err | tail -1 > gpurun_out/r02x/q1.json
python3 -c "
import json
d=json.load(open('gpurun_out/r02x/q1.json'))['q1_packed']
print({k:v for k,v in d.items() if k!='columns' and k!='note'})
for c in d['columns']: print(c)
"
