cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02as
timeout -k 10 420 python3 tools/soak_fuzz.py 1000000 12000 > gpurun_out/r02as/soak_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r02as/soak_fuzz.log
timeout -k 10 240 python3 tools/soak_host_mirror.py 20000 1500 > gpurun_out/r02as/soak_host.log 2>&1; echo "host rc=$?"; tail -1 gpurun_out/r02as/soak_host.log
timeout -k 10 120 python3 tools/soak_threads.py > gpurun_out/r02as/soak_threads.log 2>&1; echo "threads rc=$?"; tail -1 gpurun_out/r02as/soak_threads.log
timeout -k 10 200 python3 tools/soak_bitpacking.py > gpurun_out/r02as/soak_bp.log 2>&1; echo "bp rc=$?"; tail -1 gpurun_out/r02as/soak_bp.log
timeout -k 10 200 python3 tools/soak_group_sum.py > gpurun_out/r02as/soak_gs.log 2>&1; echo "gs rc=$?"; tail -2 gpurun_out/r02as/soak_gs.log
