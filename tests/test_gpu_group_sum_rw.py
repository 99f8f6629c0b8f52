"""The register-walk form of the grouped scan (k_group_sum_rw: value column walked in registers at a templated width,
keys staged as a byte per row) and its hand-over to the staged-LDS kernel: every value width 4..32 against every key
width 1..8, frames of reference on both columns, segments the register walk does not take mixed into the same column
(value widths 1..3 and > 32, key widths > 8, signed segments that wrap) — against numpy's GROUP BY and against the
staged-LDS kernel alone (adac_set_tuning("group_sum_rw", 0)).  tests/test_gpu_group_sum.py is the unchanged round-2
suite for the entry point; the reference has no grouped scan of its own (parity as there: numpy over the decoded rows)."""
import numpy as np
import pytest

from test_gpu_group_sum import encode_column, reference_groups

pytestmark = pytest.mark.gpu


def both_ways(adac, ctx, vals, keys, counts, ngroups):
    vlay, vwords = encode_column(adac, ctx, vals, counts)
    klay, kwords = encode_column(adac, ctx, keys, counts)
    d_sums, d_cnts = ctx.alloc((ngroups + 1) * 8), ctx.alloc((ngroups + 1) * 8)
    exp_s, exp_c = reference_groups(vals, keys, ngroups)
    res = []
    for rw in (1, 0, 1):   # the second register-walk pass checks that the hand-over word was left at zero
        adac.set_tuning("group_sum_rw", rw)
        vlay.scan_group_sum(vwords, klay, kwords, ngroups, d_sums, d_cnts)
        res.append((d_sums.download(np.uint64, ngroups + 1).tolist(), d_cnts.download(np.uint64, ngroups + 1).tolist()))
    adac.set_tuning("group_sum_rw", 1)
    for s, c in res:
        assert c == exp_c and s == exp_s
    return vlay.get_descs(), klay.get_descs()


@pytest.mark.parametrize("vdtype", [np.uint32, np.int32, np.uint64, np.uint16])
def test_every_value_width_against_every_key_width(adac, gpu_ctx, vdtype):
    vdtype = np.dtype(vdtype)
    rng = np.random.default_rng(31 + vdtype.itemsize)
    tb = 8 * vdtype.itemsize
    widths = [w for w in range(4, 33) if w < tb]
    # one segment per value width (ragged sizes, some spanning several scan groups), all in ONE column
    counts = np.array([int(rng.integers(3000, 70000)) for _ in widths], dtype=np.uint32)
    parts = []
    for w, c in zip(widths, counts):
        # unsigned: anywhere in the type's range; signed: a negative frame of reference, values stay below zero
        base = int(rng.integers(0, 2 ** tb - 2 ** w + 1, dtype=np.uint64)) if vdtype.kind == "u" else \
            -int(rng.integers(2 ** w, 2 ** (tb - 1) + 1, dtype=np.uint64))
        span = rng.integers(0, 2 ** w, size=int(c), dtype=np.uint64)
        span[:2] = (0, 2 ** w - 1)
        parts.append(((span.astype(object) + base) % (2 ** tb)).astype(np.dtype("u%d" % vdtype.itemsize)).view(vdtype))
    vals = np.concatenate(parts)
    for wk in range(1, 9):
        for ngroups, kbase in ((7, 0), (min(7, 2 ** wk), 0), (7, 2)):
            keys = (rng.integers(0, 2 ** wk, size=len(vals)) + kbase).astype(np.uint8 if wk + kbase.bit_length() < 8 else np.uint16)
            vd, kd = both_ways(adac, gpu_ctx, vals, keys, counts, ngroups)
            assert sorted(set(vd["width"].tolist())) == widths and set(kd["width"].tolist()) <= {wk, wk + 1}


def test_segments_the_register_walk_leaves_to_the_staged_kernel(adac, gpu_ctx):
    rng = np.random.default_rng(5150)
    counts = np.array([20000, 4096, 33000, 2048, 50000, 7, 12345, 9000], dtype=np.uint32)
    n = int(counts.sum())
    starts = [int(x) for x in np.concatenate([[0], np.cumsum(counts)[:-1]])]
    counts_i = [int(c) for c in counts]
    # int64 values: widths 2, 40, 13 (taken), 1, 24 (taken), raw 64, a signed segment that wraps (INT64 range), 6 (taken)
    vals = np.zeros(n, dtype=np.int64)
    spec = [(3, 2), (1 << 50, 40), (-50000, 13), (77, 1), (1 << 33, 24), (None, 64), ("wrap", 1), (-300, 6)]
    for (base, w), s, c in zip(spec, starts, counts_i):
        if base is None:
            vals[s:s + c] = rng.integers(np.iinfo(np.int64).min, np.iinfo(np.int64).max, size=c)
        elif base == "wrap":
            vals[s:s + c] = np.where(rng.random(c) < 0.5, np.iinfo(np.int64).max, np.iinfo(np.int64).min)
        else:
            span = rng.integers(0, 2 ** w, size=c, dtype=np.int64)
            span[:2] = (0, 2 ** w - 1)
            vals[s:s + c] = span + base
    # uint16 keys: small codes, but two segments at key width 10 (keys up to 1000: mostly overflow) and one with a base
    keys = rng.integers(0, 5, size=n).astype(np.uint16)
    keys[starts[2]:starts[2] + counts_i[2]] = rng.integers(0, 1000, size=counts_i[2])
    keys[starts[4]:starts[4] + counts_i[4]] += 300          # all of them overflow: the constant-byte form
    keys[starts[7]:starts[7] + counts_i[7]] = rng.integers(2, 6, size=counts_i[7])   # frame of reference 2 on the keys
    vd, kd = both_ways(adac, gpu_ctx, vals, keys, counts, 6)
    assert vd["width"].tolist()[:5] == [2, 40, 13, 1, 24] and kd["width"].tolist()[2] == 10
    # the rule-RECOMPACT encode of the wrapping segment is what makes it non-linear; with the append rule it stays raw
    both_ways(adac, gpu_ctx, vals.view(np.uint64), keys, counts, 6)
    both_ways(adac, gpu_ctx, (vals & 0x7fffffff).astype(np.uint32), keys.astype(np.uint8), counts, 3)
