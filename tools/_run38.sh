cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02ac
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02ac/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02ac/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python3 tools/soak_fuzz.py 800000 3000 > gpurun_out/r02ac/soak_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r02ac/soak_fuzz.log
timeout -k 10 300 python3 bench_configs.py q6_packed 2> gpurun_out/r02ac/q6.err | tail -1 > gpurun_out/r02ac/q6.json; python3 -c "
import json; d=json.load(open('gpurun_out/r02ac/q6.json'))['q6_packed']; print('q6 ms', d['q6_on_packed_ms'])"
