cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02w
timeout -k 10 900 python3 bench_configs.py 2> gpurun_out/r02w/configs.err | tail -1 > gpurun_out/r02w/configs.json; echo "configs rc=$?"; tail -2 gpurun_out/r02w/configs.err
bash tools/profile_round.sh r02d
echo done
