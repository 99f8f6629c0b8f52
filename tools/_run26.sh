cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02r
rocprofv3 --kernel-trace --stats -d gpurun_out/r02r/trace --output-format csv -- python3 bench_configs.py q1_packed > gpurun_out/r02r/q1.out 2> gpurun_out/r02r/q1.err; echo rc=$?
f=$(ls gpurun_out/r02r/trace/*/*kernel_stats.csv | head -1); head -8 $f | cut -c1-200
