"""GPU decode of DuckDB's on-disk BITPACKING segments against the oracle's restatement of the
reference's compress + scan: blocks are produced by the oracle exactly as BitpackingCompressState writes them,
decoded on the device through the C ABI, and compared bit for bit with the oracle's scan and the original rows."""
import numpy as np
import pytest

from oracle import bitpacking as bp

pytestmark = pytest.mark.gpu
ALL = [np.uint8, np.int8, np.uint16, np.int16, np.uint32, np.int32, np.uint64, np.int64]
STRIDE = 262144  # blocks are 262136 bytes; 16-byte aligned slots with slack


def mixed_column(dtype, rng, groups=9):
    """A column whose 2048-row groups exercise every mode: constant, constant delta, delta_for, for, ragged tail."""
    dtype = np.dtype(dtype)
    info = np.iinfo(dtype)
    half = int(info.max) // 2
    parts = []
    for g in range(groups):
        n = 2048
        kind = g % 5
        if kind == 0:
            parts.append(np.full(n, half // 3 + g, dtype=np.int64))
        elif kind == 1:
            parts.append(7 + (3 * np.arange(n, dtype=np.int64)) % max(half - 7, 1))
        elif kind == 2:
            steps = rng.integers(0, 3, size=n)
            parts.append(half // 2 + np.cumsum(steps) % (half // 4 + 1))
            parts[-1].sort()
        elif kind == 3:
            span = min(8 * dtype.itemsize - 2, 13)
            parts.append(half // 3 + rng.integers(0, 1 << span, size=n))
        else:
            lo = int(info.min) // 2 if dtype.kind == "i" else 0
            parts.append(rng.integers(lo, half, size=n, dtype=np.int64))
    parts.append(half // 5 + rng.integers(0, 50, size=777))  # ragged last group
    return np.concatenate(parts).astype(dtype)


def upload_blocks(ctx, comp):
    buf = np.zeros(comp.nseg * STRIDE + 64, dtype=np.uint8)
    offs, counts = [], []
    for i in range(comp.nseg):
        buf[i * STRIDE:i * STRIDE + bp.BLOCK_SIZE] = comp.block(i)
        offs.append(i * STRIDE)
        counts.append(comp.count(i))
    return ctx.upload(buf), np.array(offs, dtype=np.uint64), np.array(counts, dtype=np.uint32)


@pytest.mark.parametrize("dtype", ALL)
def test_full_scan_all_modes(adac, gpu_ctx, dtype):
    rng = np.random.default_rng(100 + np.dtype(dtype).itemsize)
    v = mixed_column(dtype, rng)
    comp = bp.Compressed(v)
    modes = comp.groups_by_mode()
    assert modes["constant"] >= 1 and modes["for"] >= 1, modes
    if np.dtype(dtype).itemsize > 1:   # 8-bit columns wrap inside a group, so their arithmetic runs are not constant-delta
        assert modes["constant_delta"] >= 1 and modes["delta_for"] >= 1, modes
    d_blocks, offs, counts = upload_blocks(gpu_ctx, comp)
    lay = adac.BitpackingLayout(gpu_ctx, dtype, offs, counts)
    assert lay.total_values == len(v) and lay.ngroups == sum((int(c) + 2047) // 2048 for c in counts)
    d_out = gpu_ctx.alloc(len(v) * v.dtype.itemsize + 64)
    d_out.upload(np.full(len(v) + 8, 0x33, dtype=np.uint8).repeat(v.dtype.itemsize)[:len(v) * v.dtype.itemsize + 8])
    lay.unpack(d_blocks, d_out)
    got = d_out.download(v.dtype, len(v))
    ref = np.concatenate([comp.scan(i) for i in range(comp.nseg)])
    assert np.array_equal(ref, v)
    assert np.array_equal(got, ref)
    # point fetches, including DELTA_FOR rows deep inside a group
    k = 3000
    rows_abs = rng.integers(0, len(v), size=k)
    starts = np.array([comp.start(i) for i in range(comp.nseg)])
    segs = (np.searchsorted(starts, rows_abs, side="right") - 1).astype(np.uint32)
    rows = (rows_abs - starts[segs]).astype(np.uint32)
    d_f = gpu_ctx.alloc(k * v.dtype.itemsize)
    lay.fetch_rows(d_blocks, gpu_ctx.upload(segs), gpu_ctx.upload(rows), k, d_f)
    assert np.array_equal(d_f.download(v.dtype, k), v[rows_abs])


def test_forced_modes_and_widths(adac, gpu_ctx):
    """force_bitpacking_mode (bitpacking.cpp:128, test-only knob of the reference): every width a type can take
    in FOR and DELTA_FOR mode, including zero-width groups and the GetEffectiveWidth jumps."""
    rng = np.random.default_rng(9)
    for dtype in (np.uint16, np.uint32, np.uint64, np.int32):
        dtype = np.dtype(dtype)
        bits = 8 * dtype.itemsize
        for mode in (bp.MODE_FOR, bp.MODE_DELTA_FOR):
            cols = []
            for w in range(0, bits - 1):
                span = rng.integers(0, 1 << w, size=2048, dtype=np.uint64) if w else np.zeros(2048, dtype=np.uint64)
                if w:
                    span[3], span[9] = 0, (1 << w) - 1
                base = 1000
                if mode == bp.MODE_DELTA_FOR:   # increments of up to w bits -> delta width w (cumsum kept in range)
                    step = np.minimum(span, np.uint64((1 << max(bits - 13, 1)) - 1))
                    col = base + np.cumsum(step.astype(object))
                    if int(col[-1]) >= (1 << (bits - 1)) - 1:
                        continue
                    cols.append(np.array(col, dtype=np.uint64))
                else:
                    if (1 << w) + base >= (1 << (bits - 1)):
                        continue
                    cols.append(span + np.uint64(base))
            v = np.concatenate(cols).astype(dtype)
            comp = bp.Compressed(v, force_mode=mode)
            d_blocks, offs, counts = upload_blocks(gpu_ctx, comp)
            lay = adac.BitpackingLayout(gpu_ctx, dtype, offs, counts)
            d_out = gpu_ctx.alloc(len(v) * dtype.itemsize + 64)
            lay.unpack(d_blocks, d_out)
            assert np.array_equal(d_out.download(dtype, len(v)), v), (dtype, mode)
            widths = {comp.group_info(0, g)[2] for g in range(min(comp.count(0) // 2048, 40))}
            assert len(widths) >= (4 if bits == 16 else 12), (dtype, mode, sorted(widths))


def test_nulls_multi_segment_and_output_placement(adac, gpu_ctx):
    rng = np.random.default_rng(17)
    n = 700_000
    v = (5_000_000 + rng.integers(0, 1 << 22, size=n)).astype(np.int32)
    valid = rng.random(n) > 0.15
    comp = bp.Compressed(v, valid)
    assert comp.nseg >= 3
    d_blocks, offs, counts = upload_blocks(gpu_ctx, comp)
    # place every segment at an odd element offset of the output: stores stay 16-byte aligned internally
    out_offs, run = [], 0
    for c in counts:
        run += 3
        out_offs.append(run)
        run += int(c)
    lay = adac.BitpackingLayout(gpu_ctx, np.int32, offs, counts, np.array(out_offs, dtype=np.uint64))
    d_out = gpu_ctx.alloc((run + 16) * 4)
    d_out.upload(np.full(run + 16, -7, dtype=np.int32))
    lay.unpack(d_blocks, d_out)
    out = d_out.download(np.int32, run + 16)
    row = 0
    for i, (o, c) in enumerate(zip(out_offs, counts)):
        ref = comp.scan(i)                       # NULL slots included: bit for bit what the reference scan yields
        assert np.array_equal(out[o:o + int(c)], ref)
        ok = valid[row:row + int(c)]
        assert np.array_equal(ref[ok], v[row:row + int(c)][ok])
        assert np.all(out[o - 3:o] == -7)
        row += int(c)
    with pytest.raises(adac.AdacError):
        adac.BitpackingLayout(gpu_ctx, np.int32, np.array([8], dtype=np.uint64), np.array([10], dtype=np.uint32))
    with pytest.raises(adac.AdacError):
        adac.BitpackingLayout(gpu_ctx, np.float32, offs, counts)


def gpu_compress(adac, ctx, v, valid=None, force_mode=0):
    d_vals = ctx.upload(v)
    d_valid = None
    if valid is not None:
        bits = np.packbits(valid, bitorder="little")
        bits = np.concatenate([bits, np.zeros((-len(bits)) % 8 + 8, np.uint8)]).view(np.uint64)
        d_valid = ctx.upload(bits)
    plan = adac.BitpackingPlan(ctx, v.dtype, d_vals, len(v), d_valid, force_mode)
    if not plan.encodable:
        return plan, None, d_vals
    d_blocks = ctx.alloc(max(plan.nseg, 1) * plan.BLOCK_STRIDE + 64)
    plan.write(d_vals, d_blocks, d_valid)
    ctx.sync()
    return plan, d_blocks, d_vals


def assert_blocks_equal_oracle(plan, d_blocks, comp):
    assert plan.nseg == comp.nseg
    assert plan.groups_by_mode() == comp.groups_by_mode()
    img = d_blocks.download(np.uint8, plan.nseg * plan.BLOCK_STRIDE)
    for i in range(comp.nseg):
        start, count, size = plan.segment(i)
        assert (start, count, size) == (comp.start(i), comp.count(i), comp.size(i)), i
        got = img[i * plan.BLOCK_STRIDE:i * plan.BLOCK_STRIDE + size]
        exp = comp.block(i)[:size]
        if not np.array_equal(got, exp):
            bad = np.nonzero(got != exp)[0]
            raise AssertionError("segment %d differs at bytes %s" % (i, bad[:10]))


@pytest.mark.parametrize("dtype", ALL)
def test_gpu_compress_writes_the_reference_block_image(adac, gpu_ctx, dtype):
    """adac_bp_plan_create + adac_bp_write against the oracle's restatement of BitpackingCompress: identical
    segment boundaries, mode per group and block bytes; then the device decodes its own blocks."""
    rng = np.random.default_rng(200 + np.dtype(dtype).itemsize)
    v = mixed_column(dtype, rng, groups=14)
    plan, d_blocks, _ = gpu_compress(adac, gpu_ctx, v)
    comp = bp.Compressed(v)
    assert_blocks_equal_oracle(plan, d_blocks, comp)
    counts = np.array([plan.segment(i)[1] for i in range(plan.nseg)], dtype=np.uint32)
    lay = adac.BitpackingLayout(gpu_ctx, dtype, np.arange(plan.nseg, dtype=np.uint64) * plan.BLOCK_STRIDE, counts)
    d_out = gpu_ctx.alloc(len(v) * v.dtype.itemsize + 64)
    lay.unpack(d_blocks, d_out)
    assert np.array_equal(d_out.download(v.dtype, len(v)), v)


def test_gpu_compress_forced_modes_nulls_and_many_segments(adac, gpu_ctx):
    rng = np.random.default_rng(77)
    # forced modes (the reference's force_bitpacking_mode test knob)
    v = (10_000 + np.cumsum(rng.integers(0, 900, size=40_000))).astype(np.int64)
    for mode in (bp.MODE_FOR, bp.MODE_DELTA_FOR, bp.MODE_CONSTANT_DELTA, bp.MODE_CONSTANT):
        plan, d_blocks, _ = gpu_compress(adac, gpu_ctx, v, force_mode=mode)
        assert_blocks_equal_oracle(plan, d_blocks, bp.Compressed(v, force_mode=mode))
    # NULL rows (stored as the value 0; oracle in the same convention) over several 256 KiB blocks
    n = 900_000
    v = (3_000_000 + rng.integers(0, 1 << 25, size=n)).astype(np.uint32)
    valid = rng.random(n) > 0.2
    valid[4096:8192] = False
    plan, d_blocks, _ = gpu_compress(adac, gpu_ctx, v, valid)
    comp = bp.Compressed(v, valid, null_zero=True)
    assert comp.nseg >= 4
    assert_blocks_equal_oracle(plan, d_blocks, comp)
    # 8- and 16-bit columns: group payloads start at odd byte offsets
    for dtype in (np.uint8, np.int16):
        info = np.iinfo(dtype)
        parts = []
        for g in range(40):
            if g % 3 == 0:
                parts.append(np.full(2048, g % 50, dtype=np.int64))
            else:
                parts.append(rng.integers(0, max(2, int(info.max) >> (g % 5 + 1)), size=2048))
        v = np.concatenate(parts).astype(dtype)
        plan, d_blocks, _ = gpu_compress(adac, gpu_ctx, v)
        assert_blocks_equal_oracle(plan, d_blocks, bp.Compressed(v))
    # a column no mode can hold: Flush() == false
    bad = np.array([np.iinfo(np.int64).min, np.iinfo(np.int64).max] * 2048, dtype=np.int64)
    plan, d_blocks, _ = gpu_compress(adac, gpu_ctx, bad)
    assert not plan.encodable and d_blocks is None and plan.nseg == 0
    with pytest.raises(ValueError):
        bp.Compressed(bad)


@pytest.mark.parametrize("dtype", ALL)
def test_range_scans_from_any_start(adac, gpu_ctx, dtype):
    """adac_bp_unpack_range = BitpackingScanPartial for arbitrary (start, count): ranges that begin and end inside
    groups of every mode (a DELTA_FOR group entered in the middle still needs its prefix), single rows, whole
    segments, unaligned output offsets — against the oracle's scan of the same block."""
    dtype = np.dtype(dtype)
    rng = np.random.default_rng(400 + dtype.itemsize + (dtype.kind == "i"))
    v = mixed_column(dtype, rng, groups=11)
    comp = bp.Compressed(v)
    d_blocks, offs, counts = upload_blocks(gpu_ctx, comp)
    lay = adac.BitpackingLayout(gpu_ctx, dtype, offs, counts)
    row0 = 0
    for seg in range(comp.nseg):
        c = int(counts[seg])
        cases = [(0, c), (0, 1), (c - 1, 1), (2047, 2), (2048, 2048), (1, c - 1)]
        for _ in range(12):
            s = int(rng.integers(0, c))
            cases.append((s, int(rng.integers(1, c - s + 1))))
        for start, cnt in cases:
            if start + cnt > c:
                continue
            shift = int(rng.integers(0, 9))
            d_out = gpu_ctx.alloc((cnt + shift) * dtype.itemsize + 64)
            d_out.upload(np.full(cnt + shift + 8, 0x5A, dtype=np.uint8).repeat(dtype.itemsize).view(dtype)[:cnt + shift])
            lay.unpack_range(d_blocks, seg, start, cnt, d_out, shift)
            got = d_out.download(dtype, cnt + shift)
            exp = comp.scan(seg, start, cnt)
            assert np.array_equal(got[shift:], exp), (dtype.name, seg, start, cnt, shift)
            assert np.array_equal(exp, v[row0 + start:row0 + start + cnt])
            if shift:
                assert (got[:shift].view(np.uint8) == 0x5A).all()      # nothing before the range is touched
        row0 += c
    with pytest.raises(adac.AdacError):
        lay.unpack_range(d_blocks, 0, int(counts[0]), 1, d_out)       # past the segment
