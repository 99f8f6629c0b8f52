cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02t
timeout -k 10 500 python3 tools/_bisect_gs.py > gpurun_out/r02t/soak.log 2>&1; tail -8 gpurun_out/r02t/soak.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02t/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02t/tests.log
timeout -k 10 500 python3 bench_configs.py q1_packed 2> gpurun_out/r02t/q1.err | tail -1 > gpurun_out/r02t/q1.json
python3 -c "
import json
d=json.load(open('gpurun_out/r02t/q1.json'))['q1_packed']
print({k:v for k,v in d.items() if k!='columns' and k!='note'})
for c in d['columns']: print(c)
"
