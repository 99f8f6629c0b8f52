O=gpurun_out/r03ak
mkdir -p $O && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -n 3 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -n 1 $O/smoke.log
timeout -k 10 500 python3 tools/soak_fuzz.py 6000000 4000 > $O/soak_fuzz.log 2>&1; tail -n 1 $O/soak_fuzz.log
timeout -k 10 200 python3 tools/soak_host_mirror.py 60000 400 > $O/soak_host.log 2>&1; tail -n 1 $O/soak_host.log
timeout -k 10 200 python3 tools/soak_group_sum.py > $O/soak_group.log 2>&1; tail -n 2 $O/soak_group.log
timeout -k 10 200 python3 tools/soak_bitpacking.py 9000 1500 > $O/soak_bp.log 2>&1; tail -n 1 $O/soak_bp.log
timeout -k 10 100 python3 tools/soak_threads.py 20 8 > $O/soak_threads.log 2>&1; tail -n 2 $O/soak_threads.log
