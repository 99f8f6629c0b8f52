"""BASELINE.json's headline configurations at FULL size on the HIP path (through the C ABI):
  C2  100 M-row uint64 Zipf(s = 1.0, n = 2^32 - 1) column, encode + full scan on one GPU
      (data shape: benchmark/micro/succinct/zipf_distribution.cpp:13-48; sampler zipf.cpp:12-100; Appender segment
      layout: duckdb-adaptive-compression_amd/layout.py)
  C4  1 B-row uint64 column whose segments are partitioned by id across 8 GPUs — run here as rank 0's and rank 7's
      shards (~125 M rows each) of the ONE global segment list, on the one GPU of the test box
      (unit of independence: row_group_collection.cpp:119-155).
The oracle checks a seeded sample of segments bit for bit (min, width, packed words, decode); every segment's min and
width are checked against numpy; the whole column through size-independent properties: encode -> decode round trip,
checksum (fused SUM), per-segment COUNT(range), the selection bitmap and the scan-with-selection."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U64 = 0xFFFFFFFFFFFFFFFF


def hi_bit(x):
    """sdsl::bits::hi on a numpy uint64 array (hi(0) == 0)."""
    x = x.astype(np.uint64)
    out = np.zeros(len(x), dtype=np.int64)
    for sh in (32, 16, 8, 4, 2, 1):
        big = (x >> np.uint64(sh)) != 0
        out[big] += sh
        x = np.where(big, x >> np.uint64(sh), x)
    return out


def check_column(adac, orc, ctx, vals, counts, sample_rng, nsample=72):
    n = len(vals)
    nseg = len(counts)
    assert int(counts.sum()) == n
    starts = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    lay = adac.Layout(ctx, np.uint64, counts)
    d_vals = ctx.upload(vals)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 128).zero()
    lay.encode(d_vals, d_words)
    ctx.sync()
    descs = lay.get_descs()
    mm = lay.get_minmax()

    # every segment: min / max / width against numpy (column_segment.cpp:351-363, rule APPEND, unsigned T)
    seg_min = np.minimum.reduceat(vals, starts[:-1])
    seg_max = np.maximum.reduceat(vals, starts[:-1])
    assert np.array_equal(mm[:, 0], seg_min) and np.array_equal(mm[:, 1], seg_max)
    w_exp = hi_bit(seg_max - seg_min) + 1
    packed_exp = w_exp < 64
    assert np.array_equal(descs["width"], np.where(packed_exp, w_exp, 64).astype(np.uint8))
    assert np.array_equal((descs["flags"] & adac.SEG_PACKED) != 0, packed_exp)
    assert np.array_equal(descs["min"][packed_exp], seg_min[packed_exp])
    assert np.array_equal(descs["count"], counts)
    arena = np.array([adac.arena_words(int(c), int(w)) for c, w in zip(counts, descs["width"])], dtype=np.uint64)
    assert np.array_equal(descs["word_off"], np.concatenate([[0], np.cumsum(arena)[:-1]]).astype(np.uint64))

    # a seeded sample of segments (head and tail ones included) bit for bit against the oracle
    pick = sorted(set([0, 1, 2, 7, 8, nseg - 2, nseg - 1] + sample_rng.choice(nseg, nsample, replace=False).tolist()))
    for s in pick:
        d = descs[s]
        v = vals[starts[s]:starts[s + 1]]
        mn, mx = orc.analyze_flat(v, 0)
        w = orc.width_from_succinct(mn, mx)
        assert (int(d["min"]), int(d["width"])) == (mn, w), "segment %d" % s
        exp = orc.pack_flat(v, mn, w)
        got = d_words.download(np.uint64, len(exp), int(d["word_off"]) * 8)
        assert np.array_equal(got, exp), "packed words of segment %d" % s
        assert np.array_equal(orc.unpack_flat(got, 0, len(v), w, mn, np.uint64), v)
        assert adac.size_in_bytes(len(v), w) == orc.size_in_bytes(len(v) * w)

    # full scan: the decode equals the input (round trip at full size)
    d_out = ctx.alloc(n * 8 + 64)
    lay.unpack(d_words, d_out)
    ctx.sync()
    assert np.array_equal(d_out.download(np.uint64, n), vals)
    del d_out

    # fused SUM: per segment and the column checksum
    d_res = ctx.alloc(nseg * 8)
    lay.scan_sum(d_words, d_res)
    sums = d_res.download(np.uint64, nseg)
    assert np.array_equal(sums, np.add.reduceat(vals, starts[:-1]))
    assert int(sums.sum(dtype=np.uint64)) == int(vals.sum(dtype=np.uint64))

    # fused COUNT(lo <= v <= hi) per segment, and COUNT(== k) for a value that exists
    median = int(np.median(vals[:2_000_000]))
    for lo, hi in ((0, median), (median, U64), (1000, 100000), (int(vals[n // 3]), int(vals[n // 3]))):
        lay.scan_count_between(d_words, lo, hi, d_res)
        hit = (vals >= np.uint64(lo)) & (vals <= np.uint64(hi))
        assert np.array_equal(d_res.download(np.uint64, nseg), np.add.reduceat(hit.astype(np.uint64), starts[:-1]))

    # selection bitmap (FilterSelection on the packed bytes) and the scan-with-selection over it
    d_bm = ctx.alloc((n + 63) // 64 * 8 + 8)
    lay.scan_select_between(d_words, 0, median, d_bm, d_res)
    hit = vals <= np.uint64(median)
    bm = d_bm.download(np.uint8, (n + 7) // 8)
    assert np.array_equal(bm, np.packbits(hit, bitorder="little"))
    assert int(d_res.download(np.uint64, nseg).sum()) == int(hit.sum())
    keep = np.flatnonzero(hit)
    d_sel = ctx.alloc(len(keep) * 8 + 64)
    d_ids = ctx.alloc(len(keep) * 8 + 64)
    assert lay.unpack_selected(d_words, d_bm, d_sel, d_ids) == len(keep)
    assert np.array_equal(d_sel.download(np.uint64, len(keep)), vals[keep])
    assert np.array_equal(d_ids.download(np.uint64, len(keep)), keep.astype(np.uint64))
    return descs


def test_c2_zipf_u64_100m(adac, oracle, gpu_ctx):
    wl = importlib.import_module(adac.__name__ + ".workload")
    n = 100_000_000
    vals = wl.zipf_column(n, np.uint64, domain=2 ** 32 - 1, skew=1.0, seed=42, threads=16)
    counts = adac.appender_segment_counts(n, 8)
    assert len(counts) == 3907  # ~488 flushes x 8 segments
    assert counts[:8].tolist() == [2048, 32767, 32767, 32767, 22531, 32767, 32767, 16386]  # SURVEY.md §3.1 [probe]
    descs = check_column(adac, oracle, gpu_ctx, vals, counts, np.random.default_rng(2))
    # the heavy tail: essentially every full segment needs 32 bits
    assert (descs["width"][counts > 30000] == 32).mean() > 0.99


@pytest.mark.parametrize("rank", [0, 7])
def test_c4_shard_of_1b(adac, oracle, gpu_ctx, rank):
    wl = importlib.import_module(adac.__name__ + ".workload")
    sh = importlib.import_module(adac.__name__ + ".sharding")
    total, world = 1_000_000_000, 8
    seg_lo, seg_hi, row_lo, row_hi, counts = sh.column_shard(total, 8, rank, world)
    assert (seg_lo, seg_hi) == sh.segment_range(39063, rank, world)  # ~39 063 segments
    assert 124_900_000 < row_hi - row_lo < 125_100_000
    if rank == world - 1:
        assert row_hi == total
    vals = wl.zipf_column_range(row_lo, row_hi, np.uint64, domain=2 ** 32 - 1, skew=1.0, seed=42, threads=16)
    # the shard is a slice of the ONE global column, not a column of its own: its first block equals the global one's
    head = wl.zipf_column(min(total, row_lo + 4096), np.uint64, domain=2 ** 32 - 1, skew=1.0, seed=42,
                          threads=16)[row_lo:row_lo + 4096] if rank == 0 else None
    if head is not None:
        assert np.array_equal(vals[:4096], head)
    check_column(adac, oracle, gpu_ctx, vals, counts, np.random.default_rng(40 + rank), nsample=64)
