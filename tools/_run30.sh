cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02v
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r02v/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r02v/tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python3 tools/soak_fuzz.py 700000 2500 > gpurun_out/r02v/soak_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r02v/soak_fuzz.log
timeout -k 10 200 python3 tools/soak_host_mirror.py 9000 300 > gpurun_out/r02v/soak_host.log 2>&1; echo "host rc=$?"; tail -2 gpurun_out/r02v/soak_host.log
timeout -k 10 200 python3 tools/soak_threads.py > gpurun_out/r02v/soak_threads.log 2>&1; echo "threads rc=$?"; tail -3 gpurun_out/r02v/soak_threads.log
timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --rows 25000000 --steps 20 --warmup 3 --single-device-rehearsal --no-sweep --no-cpu-baseline > gpurun_out/r02v/bench_n2.json 2> gpurun_out/r02v/bench_n2.err; echo "n2 rc=$?"; tail -c 600 gpurun_out/r02v/bench_n2.json
