cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02y
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r02y/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/r02y/smoke.log
