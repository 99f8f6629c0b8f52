cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02b
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r02b/tests.log 2>&1; rc=$?; echo "tests rc=$rc"
tail -4 gpurun_out/r02b/tests.log
[ $rc -eq 0 ] && bash tools/profile_round.sh r02b
echo done
