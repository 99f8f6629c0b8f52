cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02af
timeout -k 10 900 python3 bench.py --rows 1000000000 --steps 10 --warmup 2 --no-sweep --no-cpu-baseline > gpurun_out/r02af/bench_1B.json 2> gpurun_out/r02af/bench_1B.err; echo "rc=$?"; tail -3 gpurun_out/r02af/bench_1B.err | cut -c1-300; tail -c 1500 gpurun_out/r02af/bench_1B.json
