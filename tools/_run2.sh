cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_persistence.py tests/test_gpu_c2_c4.py -x -q -m gpu > gpurun_out/r02b/tests.log 2>&1; echo "tests rc=$?" 
for dbg in 0 1 2; do
  ADAC_TUNING=sel_debug=$dbg python3 tools/pmc_probe.py select,count u64:13,u64:16,u64:32,u32:8 0 10 > gpurun_out/r02b/sel_debug$dbg.json 2> gpurun_out/r02b/sel_debug$dbg.err
done
rocprofv3 --kernel-trace --stats -d gpurun_out/r02b/trace --output-format csv -- python3 tools/pmc_probe.py select,count u64:13,u64:16 0 10 > gpurun_out/r02b/trace.json 2> gpurun_out/r02b/trace.err
python bench.py --steps 20 --warmup 3 --no-sweep --cpu-seconds 3 > gpurun_out/r02b/bench.json 2> gpurun_out/r02b/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r02b/tests.log
