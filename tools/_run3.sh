cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02c
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02c/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02c/tests.log
python3 tools/pmc_probe.py select,count,repack u64:8,u64:13,u64:16,u64:20,u64:32,u32:8,u32:16,u32:24 0 10 > gpurun_out/r02c/new_times.json 2> gpurun_out/r02c/new_times.err
ADAC_TUNING=grouped_repack=0 python3 tools/pmc_probe.py repack u64:8,u32:16 0 10 > gpurun_out/r02c/old_repack.json 2> gpurun_out/r02c/old_repack.err
echo done
