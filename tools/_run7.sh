cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02f
for z in 1; do ADACH_ZERO_COPY=$z timeout -k 10 500 python bench_configs.py plugin_scan > gpurun_out/r02f/plugin_z$z.json 2> gpurun_out/r02f/plugin_z$z.err; echo "rc=$?"; done
tail -3 gpurun_out/r02f/plugin_z1.err
