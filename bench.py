#!/usr/bin/env python3
"""bench.py — decoded values/s + achieved HBM GB/s of the succinct column codec on MI355X.

Metric (BASELINE.json): decoded values/sec + achieved HBM GB/s, Zipf uint64 column, 1/2/4/8 GPU.
Workload at N=1 (config C2): 100 M-row uint64 column, values Zipf(n = 2^32-1, s = 1.0) drawn with the
reference's rejection-inversion sampler on mt19937 (seed 42 + rank), cut into segments the way the reference's
Appender does (2048 / 32767 / ... rows), encoded on the device (analyze + plan + pack, timed separately) and
then fully scanned.  One STEP = one full scan: adac_unpack over every segment of the rank's column (the
whole-column form of SuccinctScanPartial, src/storage/compression/succinct.cpp:123-144), inputs and outputs
resident in HBM.  N > 1: one process per GPU; the workload is ONE global column of --total-rows rows (default
--rows x N: weak scaling; config C4 is `--gpus 8 --total-rows 1000000000`) whose Appender segment list is
partitioned by segment id into N contiguous ranges (sharding.column_shard): rank k generates, encodes and scans
only the segments of its range in its own per-GPU pool, no data-path collective; value = total rows decoded by all
ranks / max-over-ranks time, and rank 0 checks the sum of the per-rank checksums against the column's.

One JSON line on rank 0.  Extra objects: "roofline" (dominant kernel k_unpack, algorithmic bytes / HIP-event
launch time vs the 8 TB/s HBM peak), "cpu_baseline" (the oracle's port of the reference scan loop on the host
cores, same packed words), "encode", "fused_scan", "sweep".
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PKG = "duckdb-adaptive-compression_amd"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable copy)


def kernel_source_sha256():
    """Hash of the device sources a PMC traffic record belongs to: the .hip, the internal header and every .inl
    (one helper, in the package, shared with profiles/summarize.py)."""
    return importlib.import_module(PKG).kernel_source_sha256()


def pmc_traffic_for(kernel, rows):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json) — used ONLY when they
    were measured on these very kernel sources and this row count; otherwise (None, reason).  The PMC passes cannot
    run inside this process (separate rocprofv3 --pmc runs, tools/profile_round.sh)."""
    src = {"measured_in_this_run": False, "file": "profiles/pmc_traffic.json"}
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        src.update(round=pmc.get("round"), kernel_source_sha256=pmc.get("kernel_source_sha256"))
        rec = pmc.get("kernels", {}).get(kernel)
        if pmc.get("kernel_source_sha256") != kernel_source_sha256():
            src["dropped"] = "stale: kernel sources changed since the PMC passes of that round"
        elif pmc.get("rows") != rows or rec is None:
            src["dropped"] = "measured on a different workload (rows %s) or kernel not recorded" % pmc.get("rows")
        else:
            return rec, src
    except Exception as e:  # noqa: BLE001
        src["dropped"] = "unreadable: %s" % e
    return None, src


def host_cpu():
    """(model name, logical cores of the box, cores this process may use)"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, os.cpu_count() or usable, usable


def spawn_ranks(n):
    """`python bench.py --gpus N` launched directly: start the N ranks as a torchrun CHILD process before anything
    in this process touches a GPU (never an exec after GPU init) and leave with its return code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def algorithmic_bytes(descs, type_size):
    """w/8 read (as whole packed words) + sizeof(T) written per value + 32 B descriptor/segment."""
    counts = descs["count"].astype(np.uint64)
    widths = descs["width"].astype(np.uint64)
    read = int((((counts * widths + np.uint64(63)) >> np.uint64(6)) << np.uint64(3)).sum())
    write = int(counts.sum()) * type_size
    meta = 32 * len(descs)
    return read, write, meta


class DeviceColumn:
    """One rank's shard: raw values, packed arena and decoded output resident in HBM."""

    def __init__(self, adac, torch, ctx, vals, counts, dtype):
        self.adac, self.torch, self.ctx = adac, torch, ctx
        self.dtype = np.dtype(dtype)
        self.n = len(vals)
        self.layout = adac.Layout(ctx, self.dtype, counts)
        dev = "cuda:%d" % ctx.device
        signed = np.dtype("i%d" % self.dtype.itemsize)
        tdt = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[self.dtype.itemsize]
        self.d_vals = torch.from_numpy(vals.view(signed)).to(dev)
        self.d_words = torch.zeros(self.layout.max_arena_words + 16, dtype=torch.int64, device=dev)
        self.d_out = torch.empty(self.n + 16, dtype=tdt, device=dev)
        torch.cuda.synchronize()  # allocations/fills ran on torch's stream; the codec uses its own
        self.descs = None

    def encode(self, rule, padded=False):
        self.layout.encode(self.d_vals, self.d_words, None, rule, padded)

    def unpack(self):
        self.layout.unpack(self.d_words, self.d_out)

    def fetch_descs(self):
        self.descs = self.layout.get_descs()
        return self.descs

    def verify_roundtrip(self):
        return bool(self.torch.equal(self.d_out[:self.n], self.d_vals))


def time_launches(ctx, fn, reps, rounds=1):
    """Average duration of one launch of fn over `reps` back-to-back launches, HIP events on the ctx stream; with
    rounds > 1 the median of that many such averages (the sweep: one cell of round 3's final run read half its rate
    once — u32 w 8 decode 2712 GB/s between 5738 and 5771 on the runs before — while every other cell was normal)."""
    out = []
    for _ in range(rounds):
        ctx.sync()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        out.append(ctx.timer_stop() / reps)
    return float(np.median(out))


def cpu_baseline(orc, col, vals, seconds, threads, all_threads):
    """The oracle's port of the reference scan loop (per-value read_int + min add, 2048 rows per call) on the
    host cores, over the same packed words the GPU produced; also proves at full size that the oracle decodes
    the device-packed column back to the input."""
    descs = col.descs
    arena = col.d_words.cpu().numpy().view(np.uint64)
    n = col.n
    seg_words, adds = [], []
    for d in descs:
        wo, c, w = int(d["word_off"]), int(d["count"]), int(d["width"])
        nw = (c * w + 63) // 64
        seg_words.append(arena[wo:wo + nw + 1])
        packed = bool(d["flags"] & 1) and int(d["min"]) != 0xFFFFFFFFFFFFFFFF
        adds.append(int(d["min"]) if packed else 0)
    counts = descs["count"].astype(np.uint64)
    widths = descs["width"]
    out_offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint64)
    out = np.empty(n, dtype=col.dtype)

    def run(th, nseg=None, with_copy=False):
        k = len(seg_words) if nseg is None else nseg
        t0 = time.perf_counter()
        orc.scan_segments_mt(seg_words[:k], counts[:k], widths[:k], adds[:k], out_offs[:k], col.dtype, out,
                             with_copy=with_copy, threads=th)
        return time.perf_counter() - t0, int(counts[:k].sum())

    t, rows = run(threads)  # warm + verification pass
    ok = bool(np.array_equal(out, vals))
    reps, total_t = 0, 0.0
    while total_t < seconds and reps < 200:
        t, rows = run(threads)
        total_t += t
        reps += 1
    mt_rate = rows * reps / total_t
    # the same loop on every core this process may use ("all cores"), bounded to a third of the budget
    all_rate = mt_rate
    if all_threads != threads:
        run(all_threads)
        areps, atotal = 0, 0.0
        while atotal < seconds / 3 and areps < 100:
            t, rows = run(all_threads)
            atotal += t
            areps += 1
        all_rate = rows * areps / atotal
    model, box_cores, usable = host_cpu()
    # single thread on a bounded sample of segments: copy-free and faithful (whole-vector copy per call)
    k = min(len(seg_words), 96)
    t1, r1 = run(1, k)
    t1, r1 = run(1, k)
    t2, r2 = run(1, min(k, 24), with_copy=True)
    return {
        "value": mt_rate, "unit": "values/s", "cores": threads, "kind": "port",
        "sample": "full column (%d rows, %d segments) x %d passes, copy-free scan loop, %d host threads, "
                  "one contiguous segment range per thread" % (n, len(seg_words), reps, threads),
        "cpu_model": model, "host_cores": box_cores, "usable_cores": usable,
        "all_cores": {"value": all_rate, "cores": all_threads,
                      "note": "the same loop with one segment range per usable core (the box's whole CPU, not the "
                              "per-GPU share)"},
        "single_thread_value": r1 / t1,
        "single_thread_faithful_copy_value": r2 / t2,
        "faithful_note": "faithful = with the reference's per-call deep copy of the segment's int_vector "
                         "(succinct.cpp:127); sample = first %d segments" % min(k, 24),
        "oracle_decodes_device_words_to_input": ok,
    }


def run_sweep(adac, torch, ctx, base_rows, steps):
    """Decode + fused-scan throughput for uniformly distributed values at widths 8..32 (the north star's
    '8-32-bit unpack' range), uint64 and uint32 outputs."""
    out = []
    rng = np.random.default_rng(7)
    for dtype, widths in ((np.uint64, (8, 13, 16, 20, 24, 32)), (np.uint32, (8, 13, 16, 20, 24)),
                          (np.uint16, (5, 8, 12)), (np.uint8, (3, 4, 6))):
        dtype = np.dtype(dtype)
        for w in widths:
            # the packed column must not fit the 256 MiB Infinity Cache, or the fused scan reads it from there
            rows = max(base_rows, int(400e6 * 8 / w))
            counts = adac.appender_segment_counts(rows, dtype.itemsize)
            vals = rng.integers(0, 2 ** w, size=rows, dtype=np.uint32 if w <= 32 else np.uint64).astype(dtype)
            col = DeviceColumn(adac, torch, ctx, vals, counts, dtype)
            col.encode(adac.RULE_APPEND)
            col.unpack()
            ctx.sync()
            descs = col.fetch_descs()
            assert col.verify_roundtrip()
            rd, wr, meta = algorithmic_bytes(descs, dtype.itemsize)
            ms = time_launches(ctx, col.unpack, steps, 3)
            del col.d_vals  # the raw column is not needed by the scans
            d_sums = torch.zeros(len(counts), dtype=torch.int64, device="cuda:%d" % ctx.device)
            torch.cuda.synchronize()
            col.layout.scan_sum(col.d_words, d_sums)
            ms_sum = time_launches(ctx, lambda: col.layout.scan_sum(col.d_words, d_sums), steps, 3)
            # re-compaction packed -> packed: the column re-encoded with byte-padded widths, then back to exact
            # widths (n (old_w + new_w) / 8 bytes)
            pad = adac.Layout(ctx, dtype, counts)
            d_pad = torch.zeros(pad.max_arena_words + 16, dtype=torch.int64, device="cuda:%d" % ctx.device)
            exact = adac.Layout(ctx, dtype, counts)
            d_exact = torch.zeros(exact.max_arena_words + 16, dtype=torch.int64, device="cuda:%d" % ctx.device)
            torch.cuda.synchronize()
            col.layout.reencode(col.d_words, pad, d_pad, None, adac.RULE_APPEND, True)
            pd = pad.get_descs()
            rd_pad = int(((pd["count"].astype(np.uint64) * pd["width"] + 63) // 64 * 8).sum())
            rep = lambda: pad.reencode(d_pad, exact, d_exact, None, adac.RULE_APPEND, False)
            rep()
            ctx.sync()
            ed = exact.get_descs()
            assert np.array_equal(ed["width"], descs["width"]) and torch.equal(d_exact[:16384], col.d_words[:16384])
            ms_rep = time_launches(ctx, rep, steps, 3)
            rp = lambda: pad.repack(d_pad, exact, d_exact)
            ms_rp = time_launches(ctx, rp, steps, 3)
            repack = {"old_widths": sorted(set(pd["width"].tolist())), "reencode_ms": ms_rep, "repack_kernel_ms": ms_rp,
                      "repack_GBps": (rd_pad + rd) / (ms_rp * 1e-3) / 1e9,
                      "reencode_values_per_s": rows / (ms_rep * 1e-3)}
            del pad, d_pad, exact, d_exact
            d_bm = torch.zeros((rows + 63) // 64 + 1, dtype=torch.int64, device="cuda:%d" % ctx.device)
            torch.cuda.synchronize()
            sel = lambda: col.layout.scan_select_between(col.d_words, 0, 2 ** (w - 1), d_bm, d_sums)
            sel()
            ms_sel = time_launches(ctx, sel, steps, 3)
            out.append({
                "select_read_GBps": rd / (ms_sel * 1e-3) / 1e9, "recompaction": repack,
                "dtype": "u%d" % (8 * dtype.itemsize), "width": w, "rows": rows,
                "widths_seen": sorted(set(descs["width"].tolist())),
                "decode_values_per_s": rows / (ms * 1e-3),
                "decode_total_GBps": (rd + wr + meta) / (ms * 1e-3) / 1e9,
                "decode_read_GBps": rd / (ms * 1e-3) / 1e9,
                "fused_sum_values_per_s": rows / (ms_sum * 1e-3),
                "fused_sum_read_GBps": rd / (ms_sum * 1e-3) / 1e9,
                "fused_sum_read_frac_of_peak": rd / (ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBS,
            })
            del col, d_sums, d_bm
            torch.cuda.empty_cache()
    return out


def plumbing_only(args, comm):
    """CPU rehearsal of the N>1 path with NO device work and NO codec work (the gloo world_size-2 test): rendezvous,
    the partition of ONE global column by segment id, each rank generating only its slice, barrier, max/sum
    reductions, the checksum of checksums against the whole column, rank-0 JSON.  Never reports a throughput claim."""
    sh = importlib.import_module(PKG + ".sharding")
    wl = importlib.import_module(PKG + ".workload")
    total = args.total_rows or args.rows * comm.world
    seg_lo, seg_hi, row_lo, row_hi, counts = sh.column_shard(total, 8, comm.rank, comm.world)
    vals = wl.zipf_column_range(row_lo, row_hi, np.uint64, domain=args.domain, skew=args.skew, seed=42, threads=2)
    assert len(vals) == int(counts.sum()) == row_hi - row_lo
    comm.barrier()
    elapsed = comm.max(0.001 * (comm.rank + 1))
    total_rows = comm.sum(row_hi - row_lo)
    nseg_total = comm.sum(seg_hi - seg_lo)
    checksum = comm.sum_u64(int(vals.sum(dtype=np.uint64)))
    # the per-GPU block of the real line: every rank's rows / segments (device = the local rank it would bind)
    per_rank = [{"rank": int(r[0]), "device": int(r[1]), "rows": int(r[2]), "segments": int(r[3])}
                for r in comm.gather_rows([comm.rank, comm.local_rank, row_hi - row_lo, seg_hi - seg_lo])]
    comm.barrier()
    if comm.rank == 0:
        whole = wl.zipf_column(total, np.uint64, domain=args.domain, skew=args.skew, seed=42, threads=2)
        print(json.dumps({"metric": "plumbing-only", "value": None, "n_gpus": comm.world, "data": "plumbing-only",
                          "scaling": "strong" if args.total_rows else "weak", "per_rank": per_rank,
                          "max_elapsed": elapsed, "total_rows": total_rows, "total_segments": nseg_total,
                          "rank0_rows": row_hi - row_lo, "rank0_segments": [seg_lo, seg_hi],
                          "checksum_of_checksums": "%016x" % checksum,
                          "column_checksum": "%016x" % int(whole.sum(dtype=np.uint64))}), flush=True)
    comm.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=100_000_000, help="rows per GPU (weak scaling) when --total-rows is not given")
    ap.add_argument("--total-rows", type=int, default=0,
                    help="rows of the ONE global column partitioned across the ranks (C4: 1000000000 with --gpus 8)")
    ap.add_argument("--skew", type=float, default=1.0)
    ap.add_argument("--domain", type=int, default=2 ** 32 - 1)
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads of the CPU baseline (the box's CPU share per GPU)")
    ap.add_argument("--sweep-rows", type=int, default=200_000_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--backend", default=None)
    ap.add_argument("--plumbing-only", action="store_true")
    ap.add_argument("--single-device-rehearsal", action="store_true",
                    help="map every rank to cuda:0 (rehearse N>1 on a one-GPU box; use with --backend gloo)")
    args = ap.parse_args()

    sh = importlib.import_module(PKG + ".sharding")
    rank, local_rank, world = sh.dist_env()
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            sys.exit(spawn_ranks(args.gpus))
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE is %d (launch with torchrun --nproc-per-node %d, or run "
                         "`python bench.py --gpus %d` directly and let it start the ranks)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if args.plumbing_only:
        return plumbing_only(args, sh.Comm(backend=args.backend or "gloo"))

    import torch
    adac = importlib.import_module(PKG)
    if not torch.cuda.is_available():
        raise adac.AdacError(5, "bench.py needs an MI355X (no CPU fallback)")
    if args.single_device_rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    comm = sh.Comm(backend=args.backend, device=local_rank)
    # the native libraries normally arrive prebuilt with the tree; if not, rank 0 builds and the rest wait
    if rank == 0 and not (os.path.exists(adac.LIB_PATH) and os.path.exists(os.path.join(ROOT, PKG, "libadacworkload.so"))):
        importlib.import_module("__graft_entry__").build()
    comm.barrier()
    wl = importlib.import_module(PKG + ".workload")
    stream = torch.cuda.Stream(device=local_rank)
    ctx = adac.Context(local_rank, stream.cuda_stream)

    dtype = np.dtype(np.uint64)
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    gen_threads = max(1, min(32, ncpu // max(1, min(world, 8))))
    t0 = time.perf_counter()
    # ONE global column, partitioned by segment id: this rank owns segments [seg_lo, seg_hi) = rows [row_lo, row_hi)
    total_rows_cfg = args.total_rows or args.rows * world
    seg_lo, seg_hi, row_lo, row_hi, counts = sh.column_shard(total_rows_cfg, dtype.itemsize, rank, world)
    my_rows = row_hi - row_lo
    vals = wl.zipf_column_range(row_lo, row_hi, dtype, domain=args.domain, skew=args.skew, seed=42, threads=gen_threads)
    log("[rank %d] segments [%d, %d) = rows [%d, %d) of the %d-row column generated in %.1f s"
        % (rank, seg_lo, seg_hi, row_lo, row_hi, total_rows_cfg, time.perf_counter() - t0))
    col = DeviceColumn(adac, torch, ctx, vals, counts, dtype)

    # ---- encode (timed separately; not part of the step) ----
    col.encode(adac.RULE_APPEND)
    ctx.sync()
    enc_ms = time_launches(ctx, lambda: col.encode(adac.RULE_APPEND), 10)
    # the same kernel with first-come arena placement (offsets in order of completion instead of segment order)
    adac.set_tuning("encode_placement", 1)
    col.encode(adac.RULE_APPEND)
    ctx.sync()
    enc_fc_ms = time_launches(ctx, lambda: col.encode(adac.RULE_APPEND), 10)
    adac.set_tuning("encode_placement", 0)
    col.encode(adac.RULE_APPEND)
    ctx.sync()
    # A2 alone: the append path carries min/max (succinct.cpp:286-299), so Compact() is width decision + one pack pass

    def plan_pack():
        col.layout.plan(adac.RULE_APPEND, False)
        col.layout.pack(col.d_vals, col.d_words)
    pack_ms = time_launches(ctx, plan_pack, 10)
    descs = col.fetch_descs()
    rd, wr, meta = algorithmic_bytes(descs, dtype.itemsize)
    wh = {}
    for w, c in zip(descs["width"].tolist(), descs["count"].tolist()):
        wh[w] = wh.get(w, 0) + c

    # ---- correctness at full size (not timed): round trip + checksum ----
    col.unpack()
    ctx.sync()
    roundtrip_ok = col.verify_roundtrip()
    d_sums = torch.zeros(len(counts), dtype=torch.int64, device=col.d_vals.device)
    torch.cuda.synchronize()
    col.layout.scan_sum(col.d_words, d_sums)
    ctx.sync()
    dev_checksum = int(d_sums.sum().item()) & (2 ** 64 - 1)   # sum mod 2^64 of the rank's per-segment device SUMs
    host_checksum = int(vals.sum(dtype=np.uint64))
    checksum_ok = dev_checksum == host_checksum
    if not (roundtrip_ok and checksum_ok):
        raise RuntimeError("parity failure at full size: roundtrip=%s checksum=%s" % (roundtrip_ok, checksum_ok))

    # ---- the timed region: W warm-up scans, then exactly K scans between barriers ----
    for _ in range(args.warmup):
        col.unpack()
    ctx.sync()
    torch.cuda.synchronize()
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    for _ in range(args.steps):
        col.unpack()
    ev_ms = ctx.timer_stop()  # HIP events on the stream k_unpack is launched on (synchronises)
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = comm.max(elapsed)
    total_rows = comm.sum(my_rows)
    # checksum of checksums: the ranks' device SUMs add up to the whole column's (each rank checked its own above)
    global_dev_checksum = comm.sum_u64(dev_checksum)
    global_host_checksum = comm.sum_u64(host_checksum)
    total_segments = comm.sum(len(counts))
    if total_rows != total_rows_cfg or global_dev_checksum != global_host_checksum:
        raise RuntimeError("parity failure across ranks: rows %d/%d checksum %x/%x"
                           % (total_rows, total_rows_cfg, global_dev_checksum, global_host_checksum))
    value = total_rows * args.steps / elapsed
    launch_ms = ev_ms / args.steps
    launch_ms_max = comm.max(launch_ms)
    # per GPU (north star: "aggregate and per-GPU"): every rank's own rows / its own HIP-event launch time
    per_rank = [{"rank": int(r[0]), "device": int(r[1]), "rows": int(r[2]), "segments": int(r[3]),
                 "launch_ms": r[4], "values_per_s": r[2] / (r[4] * 1e-3), "hbm_GBps": r[5] / (r[4] * 1e-3) / 1e9}
                for r in comm.gather_rows([rank, local_rank, my_rows, len(counts), launch_ms, rd + wr + meta])]

    result = {
        "metric": "decoded values/sec + achieved HBM GB/s, Zipf uint64 column",
        "value": value,
        "unit": "values/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        # --total-rows fixes the whole job (ONE column cut N ways: strong); otherwise every rank brings --rows (weak)
        "scaling": "strong" if args.total_rows else "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": "%s: ONE %d-row uint64 Zipf(s=%.1f, n=%d) column (mt19937 seed 42 + 2^20-row block id), "
                        "Appender segment layout, full scan (decode to HBM); encode timed separately"
                        % ("C2" if world == 1 and total_rows_cfg == 100_000_000 else
                           "C4" if total_rows_cfg == 1_000_000_000 else "C2-shaped", total_rows_cfg, args.skew,
                           args.domain),
            "total_rows": total_rows_cfg,
            "total_segments": total_segments,
            "rank0_rows": my_rows,
            "rank0_segments": int(len(counts)),
            "rank0_rows_by_width": {str(k): int(v) for k, v in sorted(wh.items())},
            "sharding": "the column's segment list partitioned by segment id into %d contiguous ranges, one pool per "
                        "GPU, no collective on the data path" % world,
        },
        "per_rank": per_rank,
        "per_gpu_values_per_s": {"min": min(r["values_per_s"] for r in per_rank),
                                 "max": max(r["values_per_s"] for r in per_rank)},
        "achieved_HBM_GBps_aggregate": comm.sum(rd + wr + meta) / (launch_ms_max * 1e-3) / 1e9,
        "parity": {"roundtrip_full_size": roundtrip_ok, "checksum_full_size": checksum_ok,
                   "checksum_of_checksums": "%016x" % global_dev_checksum,
                   "oracle_pin": "unpinned: the reference holds no tests, golden vectors or fixtures for this path and "
                                 "its SDSL library is absent; the oracle is anchored by the known answers recorded in "
                                 "SURVEY.md §8c (tests/golden/survey_known_answers.json)"},
    }

    # roofline of the dominant kernel (k_unpack<u64>) on this rank
    ach = (rd + wr + meta) / (launch_ms * 1e-3) / 1e9
    rec, traffic_source = pmc_traffic_for("k_unpack<u64>", my_rows)
    traffic = rec["hbm_bytes"] if rec else None
    result["roofline"] = {
        "kernel": "k_unpack<u64>", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
        "algorithmic_bytes_per_launch": rd + wr + meta, "read_bytes": rd, "write_bytes": wr,
        "launch_ms": launch_ms,
        "read_GBps": rd / (launch_ms * 1e-3) / 1e9, "read_frac_of_peak": rd / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
    }
    result["encode"] = {
        "values_per_s": my_rows / (enc_ms * 1e-3), "ms": enc_ms,
        "algorithmic_GBps": (wr + rd) / (enc_ms * 1e-3) / 1e9,
        "note": "adac_encode: single-pass kernel (k_encode_1p, persistent: a segment stays in the registers of one "
                "workgroup across min/max, width, arena placement and pack; the raw column is read once: "
                "algorithmic bytes = raw + packed) for 4- and 8-byte types; BitCompressFromUncompressed's two passes "
                "(column_segment.cpp:385-456) in one; arena offsets = exclusive prefix in segment order (deterministic)",
        "first_come_placement": {
            "ms": enc_fc_ms, "values_per_s": my_rows / (enc_fc_ms * 1e-3),
            "algorithmic_GBps": (wr + rd) / (enc_fc_ms * 1e-3) / 1e9,
            "note": "adac_set_tuning(\"encode_placement\", 1): a segment takes its arena space from a cursor when its "
                    "width is known, no ordered look-back; same words, widths and mins, offsets differ run to run"},
        "compact_after_append": {
            "ms": pack_ms, "values_per_s": my_rows / (pack_ms * 1e-3),
            "algorithmic_GBps": (wr + rd) / (pack_ms * 1e-3) / 1e9,
            "note": "plan + pack with the min/max the append path carries: BitCompressFromSuccinct "
                    "(column_segment.cpp:348-383)"},
    }

    if world == 1 and rank == 0:
        # fused scan+sum (no materialisation): the read-roofline variant
        ms_sum = time_launches(ctx, lambda: col.layout.scan_sum(col.d_words, d_sums), args.steps)
        result["fused_scan"] = {
            "kernel": "k_scan_agg<u64,sum>", "values_per_s": my_rows / (ms_sum * 1e-3),
            "read_GBps": rd / (ms_sum * 1e-3) / 1e9, "read_frac_of_peak": rd / (ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBS,
        }
        # the north star's own figure — fraction of the HBM-READ roofline — is reachable only by a scan that does not
        # materialise (a decode to u64 writes 2 bytes per byte read at w = 32): the fused SUM over the same column
        rec_r, src_r = pmc_traffic_for("k_scan_agg<u64,sum>", my_rows)
        result["roofline_read"] = {
            "kernel": "k_scan_agg<u64,sum>", "bound": "hbm", "achieved": rd / (ms_sum * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rd / (ms_sum * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic": rec_r["hbm_bytes"] if rec_r else None, "traffic_source": src_r,
            "algorithmic_bytes_per_launch": rd, "launch_ms": ms_sum,
            "note": "algorithmic bytes = the packed words of every segment (w/8 per value), nothing written but one "
                    "64-bit sum per segment",
        }
        # filter scan with a selection-bitmap result (FilterSelection on packed bytes): value <= median
        d_bm = torch.zeros((my_rows + 63) // 64 + 1, dtype=torch.int64, device=col.d_vals.device)
        median = int(np.median(vals[:1_000_000]))
        torch.cuda.synchronize()
        sel = lambda: col.layout.scan_select_between(col.d_words, 0, median, d_bm, d_sums)
        sel()
        ctx.sync()
        if int(d_sums.sum().item()) != int((vals <= median).sum()):
            raise RuntimeError("parity failure: selection count")
        ms_sel = time_launches(ctx, sel, args.steps)
        result["fused_scan"]["select_bitmap"] = {
            "kernel": "k_scan_agg<u64,select>", "values_per_s": my_rows / (ms_sel * 1e-3), "ms": ms_sel,
            "read_GBps": rd / (ms_sel * 1e-3) / 1e9, "bitmap_bytes": (my_rows + 7) // 8,
            "note": "includes the tiny kernel that merges the bitmap words two scan groups share (no clearing pass, "
                    "no global atomics: the scan writes every other word whole)",
        }
        # scan-with-selection: decode only the rows the bitmap keeps (dense output + element ids)
        nsel = int((vals <= median).sum())
        d_gout = torch.empty(nsel + 16, dtype=torch.int64, device=col.d_vals.device)
        d_gids = torch.empty(nsel + 16, dtype=torch.int64, device=col.d_vals.device)
        torch.cuda.synchronize()
        got = col.layout.unpack_selected(col.d_words, d_bm, d_gout, d_gids)
        keep = np.flatnonzero(vals <= median)
        if got != nsel or not torch.equal(d_gout[:nsel].cpu(), torch.from_numpy(vals[keep].view(np.int64))) \
                or not torch.equal(d_gids[:nsel].cpu(), torch.from_numpy(keep.astype(np.int64))):
            raise RuntimeError("parity failure: unpack_selected")
        ms_g = time_launches(ctx, lambda: col.layout.unpack_selected(col.d_words, d_bm, d_gout, d_gids, False), args.steps)
        result["fused_scan"]["unpack_selected"] = {
            "selected_rows": nsel, "ms": ms_g, "selected_values_per_s": nsel / (ms_g * 1e-3),
            "scanned_values_per_s": my_rows / (ms_g * 1e-3),
            "note": "values + element ids of the selected rows, dense, row order (popcount, prefix, gather kernels)"}
        del d_bm, d_gout, d_gids
        # A6: point fetch (SuccinctFetchRow) — 16 M uniformly random (segment, row) look-ups in one launch
        nf = 1 << 24
        frng = np.random.default_rng(99)
        glob = frng.integers(0, my_rows, size=nf, dtype=np.int64)
        starts = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
        fseg = (np.searchsorted(starts, glob, side="right") - 1).astype(np.uint32)
        frow = (glob - starts[fseg]).astype(np.uint32)
        d_fseg = torch.from_numpy(fseg.view(np.int32)).to(col.d_vals.device)
        d_frow = torch.from_numpy(frow.view(np.int32)).to(col.d_vals.device)
        d_fout = torch.empty(nf, dtype=torch.int64, device=col.d_vals.device)
        torch.cuda.synchronize()
        fetch = lambda: col.layout.fetch_rows(col.d_words, d_fseg, d_frow, nf, d_fout)
        fetch()
        ctx.sync()
        if not torch.equal(d_fout.cpu(), torch.from_numpy(vals[glob].view(np.int64))):
            raise RuntimeError("parity failure: point fetch")
        ms_f = time_launches(ctx, fetch, 10)
        result["point_fetch"] = {"kernel": "k_fetch<u64>", "lookups": nf, "ms": ms_f,
                                 "lookups_per_s": nf / (ms_f * 1e-3),
                                 "note": "uniformly random rows of the C2 column, ids and results resident in HBM"}
        del d_fseg, d_frow, d_fout
        # measured device copy rate, for "fraction of achievable" next to "fraction of spec peak"
        src = col.d_vals
        dst = torch.empty_like(src)
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            dst.copy_(src)
            ms_copy = time_launches(ctx, lambda: dst.copy_(src), 10)
        copy_gbps = 2 * src.numel() * 8 / (ms_copy * 1e-3) / 1e9
        result["device_copy_GBps"] = copy_gbps
        result["roofline"]["frac_of_measured_copy"] = ach / copy_gbps
        del dst
        if not args.no_cpu_baseline:
            import oracle as orc
            orc.build()
            threads = max(1, min(ncpu, args.cpu_threads))
            result["cpu_baseline"] = cpu_baseline(orc, col, vals, args.cpu_seconds, threads, ncpu)
        if not args.no_sweep:
            del col.d_out
            torch.cuda.empty_cache()
            result["sweep"] = run_sweep(adac, torch, ctx, args.sweep_rows, min(50, max(5, args.steps // 2)))

    if rank == 0:
        print(json.dumps(result), flush=True)
    comm.close()


if __name__ == "__main__":
    main()
