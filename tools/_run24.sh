cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02p
timeout -k 10 400 python -m pytest tests/test_gpu_group_sum.py -x -q -m gpu > gpurun_out/r02p/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -25 gpurun_out/r02n/tests.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02p
timeout -k 10 500 python3 bench_configs.py q1_packed > gpurun_out/r02p/q1.json 2> gpurun_out/r02p/q1.err; echo rc=$?; tail -3 gpurun_out/r02p/q1.err
python3 -c "
import json
d=json.load(open('gpurun_out/r02p/q1.json'))['q1_packed']
print({k:v for k,v in d.items() if k!='columns' and k!='note'})
for c in d['columns']: print(c)
"
