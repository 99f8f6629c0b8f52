"""The N>1 path on CPU: world_size-2 gloo rehearsal of bench.py's control flow (rendezvous over 127.0.0.1, ONE
global column partitioned by segment id with each rank generating only its slice, barriers, max-over-ranks time,
sum-over-ranks rows, the checksum of checksums, rank-0 JSON) and the segment partition.  The data path has no
collective by design (segments are independent), so only the device codec itself is not rehearsed."""
import importlib
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_segment_partition(adac):
    sh = importlib.import_module(adac.__name__ + ".sharding")
    for nseg in (0, 1, 7, 8, 3907, 39063):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(world):
                lo, hi = sh.segment_range(nseg, r, world)
                assert lo == seen and hi >= lo
                seen = hi
                for s in {lo, (lo + hi) // 2, hi - 1} if hi > lo else ():
                    assert sh.owner_of(s, nseg, world) == r
                assert hi - lo in (nseg // world, nseg // world + 1)
            assert seen == nseg


def test_two_rank_gloo_plumbing(adac):
    adac.build()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--plumbing-only", "--backend", "gloo", "--rows", "500000"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()  # exactly one JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["data"] == "plumbing-only" and r["value"] is None and r["scaling"] == "weak"
    assert r["total_rows"] == 2 * 500000          # --rows is per GPU: the global column has rows x world rows
    assert abs(r["max_elapsed"] - 0.002) < 1e-9   # MAX over ranks, not rank 0's own 0.001
    counts = adac.appender_segment_counts(2 * 500000, 8)   # ONE column of 1 M rows, not two of 500 k
    assert r["total_segments"] == len(counts)
    assert r["rank0_segments"] == [0, len(counts) // 2]
    assert r["rank0_rows"] == int(counts[:len(counts) // 2].sum())
    # the ranks' slices are disjoint, cover the column and hold the global column's values
    assert r["checksum_of_checksums"] == r["column_checksum"]


def test_eight_rank_gloo_rehearsal_of_the_c4_launch_line(adac):
    """The driver's own N = 8 command (python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py
    --gpus 8) with C4's flag (--total-rows: ONE column cut eight ways = strong scaling), at reduced rows and without
    any device work: rendezvous of eight ranks, the eight-way partition, the per-rank block of the JSON line."""
    adac.build()
    total = 2_400_000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--total-rows", str(total), "--plumbing-only",
           "--backend", "gloo"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout.decode()
    r = json.loads(lines[0])
    counts = adac.appender_segment_counts(total, 8)
    assert r["n_gpus"] == 8 and r["scaling"] == "strong" and r["total_rows"] == total
    assert r["total_segments"] == len(counts) and r["checksum_of_checksums"] == r["column_checksum"]
    pr = r["per_rank"]
    assert [p["rank"] for p in pr] == list(range(8)) and [p["device"] for p in pr] == list(range(8))
    assert sum(p["rows"] for p in pr) == total and sum(p["segments"] for p in pr) == len(counts)
    assert max(p["segments"] for p in pr) - min(p["segments"] for p in pr) <= 1   # balanced to within one segment
    assert abs(r["max_elapsed"] - 0.008) < 1e-9


def test_column_shards_are_slices_of_one_column(adac):
    import numpy as np
    adac.build()
    sh = importlib.import_module(adac.__name__ + ".sharding")
    wl = importlib.import_module(adac.__name__ + ".workload")
    total = 3_300_000   # a few 2^20-row generator blocks, shard boundaries inside blocks
    whole = wl.zipf_column(total, np.uint64, seed=42, threads=4)
    counts = adac.appender_segment_counts(total, 8)
    for world in (1, 3, 8):
        rows_seen, segs_seen = 0, 0
        for rank in range(world):
            seg_lo, seg_hi, row_lo, row_hi, c = sh.column_shard(total, 8, rank, world)
            assert (seg_lo, row_lo) == (segs_seen, rows_seen) and np.array_equal(c, counts[seg_lo:seg_hi])
            part = wl.zipf_column_range(row_lo, row_hi, np.uint64, seed=42, threads=3)
            assert np.array_equal(part, whole[row_lo:row_hi])
            rows_seen, segs_seen = row_hi, seg_hi
        assert (rows_seen, segs_seen) == (total, len(counts))


def test_bench_refuses_a_gpu_count_that_is_not_the_world_size():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")   # as if launched as one rank of a too-small torchrun
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plumbing-only"], env=env,
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert out.returncode != 0 and b"--gpus 8 but WORLD_SIZE is 1" in out.stderr
