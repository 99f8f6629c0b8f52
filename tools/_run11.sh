cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02j
# N = 2 and N = 4 rehearsal on the ONE GPU of this box: every rank on cuda:0, gloo for the scalar reductions
for n in 2 4; do
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600+n)) bench.py --gpus $n --steps 20 --warmup 3 --rows 25000000 --single-device-rehearsal --backend gloo --no-sweep --no-cpu-baseline > gpurun_out/r02j/bench_n$n.json 2> gpurun_out/r02j/bench_n$n.err; echo "n=$n rc=$?"
done
# C4 shape: 1 B rows over 8 ranks is 125 M rows per rank; rehearse rank slices with 4 ranks x 50 M rows of a 200 M-row column
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 4 --steps 10 --warmup 2 --total-rows 200000000 --single-device-rehearsal --backend gloo --no-sweep --no-cpu-baseline > gpurun_out/r02j/bench_total200M_n4.json 2> gpurun_out/r02j/bench_total200M_n4.err; echo "total rc=$?"
# the direct form: python bench.py --gpus 2 spawns its ranks itself (torchrun child) -- here it must fail cleanly: 2 GPUs are not there
timeout -k 10 200 python bench.py --gpus 2 --steps 5 --warmup 1 --rows 1000000 --no-sweep --no-cpu-baseline > gpurun_out/r02j/direct.json 2> gpurun_out/r02j/direct.err; echo "direct rc=$?"
tail -3 gpurun_out/r02j/direct.err
