"""The N>1 path on CPU: world_size-2 gloo rehearsal of bench.py's control flow (rendezvous over 127.0.0.1,
barriers, max-over-ranks time, sum-over-ranks rows, rank-0 JSON) and the segment partition.  The data path has
no collective by design (segments are independent), so there is nothing else to rehearse."""
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
    assert r["n_gpus"] == 2 and r["data"] == "plumbing-only" and r["value"] is None
    assert r["total_rows"] == 2 * 500000
    assert abs(r["max_elapsed"] - 0.002) < 1e-9   # MAX over ranks, not rank 0's own 0.001
    nseg = len(adac.appender_segment_counts(500000, 8))
    assert r["total_segments"] == 2 * nseg
