cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02i
./tools/repro_validity_lookup > gpurun_out/r02i/repro_validity.txt 2>&1; tail -2 gpurun_out/r02i/repro_validity.txt
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02i/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02i/tests.log
SOAK_POOLS=3 timeout -k 10 200 python tools/soak_threads.py 60 10 > gpurun_out/r02i/soak_threads.txt 2>&1; echo "soak_threads rc=$?"; tail -3 gpurun_out/r02i/soak_threads.txt
timeout -k 10 200 python tools/soak_host_mirror.py 60 > gpurun_out/r02i/soak_host.txt 2>&1; echo "soak_host rc=$?"; tail -2 gpurun_out/r02i/soak_host.txt
timeout -k 10 200 python tools/soak_fuzz.py 60 > gpurun_out/r02i/soak_fuzz.txt 2>&1; echo "soak_fuzz rc=$?"; tail -2 gpurun_out/r02i/soak_fuzz.txt
timeout -k 10 600 python bench_configs.py > gpurun_out/r02i/configs.json 2> gpurun_out/r02i/configs.err; echo "configs rc=$?"
