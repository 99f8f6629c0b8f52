cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02q
timeout -k 10 400 python -m pytest tests/test_gpu_group_sum.py -x -q -m gpu > gpurun_out/r02q/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02q/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 500 python3 bench_configs.py q1_packed 2> gpurun_out/r02q/q1.err | tail -1 > gpurun_out/r02q/q1.json; tail -3 gpurun_out/r02q/q1.err
python3 -c "
import json
d=json.load(open('gpurun_out/r02q/q1.json'))['q1_packed']
print({k:v for k,v in d.items() if k!='columns' and k!='note'})
for c in d['columns']: print(c)
"
