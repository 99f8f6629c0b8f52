cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02au
for v in base ntg base ntg; do
cp build_variants/libadac_$v.so duckdb-adaptive-compression_amd/libadacodec.so; touch duckdb-adaptive-compression_amd/libadacodec.so
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-sweep --no-cpu-baseline 2> gpurun_out/r02au/err_$v.txt | tail -1 > gpurun_out/r02au/b_$v.json
python3 -c "
import json
d=json.load(open('gpurun_out/r02au/b_$v.json'))
print('$v', 'unpack_selected ms', d['fused_scan']['unpack_selected']['ms'], 'decode ms', d['ms_per_step'])"
done
