"""f-3 — persistence of packed segments, on the device.  The reference never persists a SUCCINCT
segment (ColumnSegment::ConvertToPersistent returns early, src/storage/table/column_segment.cpp:529-533); the block
image defined in include/adacodec.h is sdsl::int_vector<0>::serialize (third_party/sdsl/include/sdsl/
int_vector.hpp:602-609,1565-1578) + a 16-byte trailer.  Round trips, both ways of producing the image:
  host path    encode on the GPU -> D2H words -> adac_block_write -> adac_block_read -> H2D -> adac_unpack
  device path  adac_blocks_write (images built in HBM, one kernel for the batch) -> D2H -> byte-identical to the
               host path's images -> adac_block_peek -> H2D -> adac_blocks_read -> adac_unpack
and the header against the oracle's restatement of SDSL's serialize / load."""
import numpy as np
import pytest

from test_gpu_parity import ALL_DTYPES, make_values

pytestmark = pytest.mark.gpu

U64 = 0xFFFFFFFFFFFFFFFF


def roundtrip(adac, orc, ctx, dtype, counts, seg_vals, rule=0, padded=False, validity=None):
    dtype = np.dtype(dtype)
    nseg = len(counts)
    lay = adac.Layout(ctx, dtype, counts)
    n = int(counts.sum())
    host_vals = np.concatenate(seg_vals) if n else np.zeros(0, dtype=dtype)
    d_vals = ctx.upload(host_vals if n else np.zeros(2, dtype=dtype))
    d_valid = None if validity is None else ctx.upload(validity)
    d_words = ctx.alloc(lay.max_arena_words * 8 + 128).zero()
    lay.encode(d_vals, d_words, d_valid, rule, padded)
    ctx.sync()
    descs = lay.get_descs()
    arena = d_words.download(np.uint64, lay.max_arena_words)
    d_ref = ctx.alloc(max(n, 2) * dtype.itemsize + 64)
    lay.unpack(d_words, d_ref)
    ctx.sync()
    decoded = d_ref.download(dtype, n)

    # ---- host path: one image per segment from the downloaded words ----
    images = []
    for d in descs:
        nw = adac.packed_words(int(d["count"]), int(d["width"]))
        words = arena[int(d["word_off"]):int(d["word_off"]) + nw]
        img = adac.block_write(d, dtype, words)
        assert len(img) == adac.size_in_bytes(int(d["count"]), int(d["width"])) + 16
        # the first size_in_bytes bytes ARE sdsl::int_vector<0>::serialize of the packed vector (oracle restatement)
        assert img[:len(img) - 16] == orc.serialize(words, int(d["count"]) * int(d["width"]), int(d["width"]))
        bs, w, back = orc.load(img[:len(img) - 16])
        assert (bs, w) == (int(d["count"]) * int(d["width"]), int(d["width"])) and np.array_equal(back, words)
        images.append(img)

    # ---- device path: all images in one kernel, then one copy down ----
    strides = np.array([adac.block_stride(int(d["count"]), int(d["width"])) for d in descs], dtype=np.uint64)
    offs = np.concatenate([[0], np.cumsum(strides)[:-1]]).astype(np.uint64)
    total = int(strides.sum())
    d_blocks = ctx.alloc(total + 64)
    d_blocks.upload(np.full(total + 64, 0xA5, dtype=np.uint8))  # every byte of every image must be written
    adac.blocks_write(ctx, dtype, descs, offs, d_words, d_blocks)
    blob = d_blocks.download(np.uint8, total)
    for i, img in enumerate(images):
        o, s = int(offs[i]), int(strides[i])
        assert blob[o:o + len(img)].tobytes() == img, "device image of segment %d" % i
        assert not blob[o + len(img):o + s].any()   # zero padding up to the 8-byte unit

    # ---- load both ways into fresh arenas (different placement: reversed segment order) ----
    new_descs = np.zeros(nseg, dtype=adac.SEGMENT_DESC_DTYPE)
    woff = 0
    order = list(range(nseg))[::-1]
    for i in order:
        d, t = adac.block_peek(blob[int(offs[i]):int(offs[i]) + int(strides[i])].tobytes())
        d2, t2, words = adac.block_read(images[i])
        assert t == t2 == dtype and d.tobytes() == d2.tobytes()
        assert (int(d["count"]), int(d["width"]), int(d["min"]), int(d["flags"])) == \
            (int(descs[i]["count"]), int(descs[i]["width"]), int(descs[i]["min"]), int(descs[i]["flags"]))
        new_descs[i] = d
        new_descs[i]["word_off"] = woff
        new_descs[i]["val_off"] = descs[i]["val_off"]
        woff += adac.arena_words(int(d["count"]), int(d["width"]))
    arena_words = max(woff, 16)
    # host path: words placed by the host, one upload
    host_arena = np.zeros(arena_words, dtype=np.uint64)
    for i in range(nseg):
        _, _, words = adac.block_read(images[i])
        o = int(new_descs[i]["word_off"])
        host_arena[o:o + len(words)] = words
    d_w1 = ctx.upload(host_arena)
    # device path: the block buffer goes up as it came down, the words move on the device
    d_w2 = ctx.alloc(arena_words * 8 + 128)
    d_w2.upload(np.full(arena_words * 8 + 128, 0x5A, dtype=np.uint8))   # tail words must be zeroed by the call
    d_up = ctx.upload(blob) if total else ctx.alloc(64)
    adac.blocks_read(ctx, dtype, new_descs, offs, d_up, d_w2)
    assert np.array_equal(d_w2.download(np.uint64, arena_words), host_arena)
    for d_w in (d_w1, d_w2):
        lay2 = adac.Layout(ctx, dtype, counts)
        lay2.set_descs(new_descs)
        d_out = ctx.alloc(max(n, 2) * dtype.itemsize + 64)
        lay2.unpack(d_w, d_out)
        ctx.sync()
        out = d_out.download(dtype, n)
        assert np.array_equal(out, decoded)
        if validity is None:
            assert np.array_equal(out, host_vals)
    return descs


@pytest.mark.parametrize("dtype", ALL_DTYPES)
def test_block_images_round_trip_every_type_and_width(adac, oracle, gpu_ctx, dtype):
    rng = np.random.default_rng(31 + np.dtype(dtype).itemsize)
    tb = 8 * np.dtype(dtype).itemsize
    tile = adac.tile_values(dtype)
    counts, seg_vals = [], []
    for w in range(1, tb + 1):   # every width; w == tb stays unpacked (raw slots in the image, flags 0)
        n = int(rng.integers(1, 2 * tile + 77))
        counts.append(n)
        seg_vals.append(make_values(rng, dtype, n, w))
    counts += [1, 63, 64, 65]
    seg_vals += [make_values(rng, dtype, c, 3) for c in (1, 63, 64, 65)]
    counts = np.array(counts, dtype=np.uint32)
    for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
        for padded in (False, True):
            descs = roundtrip(adac, oracle, gpu_ctx, dtype, counts, seg_vals, rule, padded)
    assert not (descs["flags"][tb - 1] & adac.SEG_PACKED)


def test_block_images_with_null_slots_and_all_ones_segment(adac, oracle, gpu_ctx):
    rng = np.random.default_rng(5)
    for dtype in (np.int32, np.uint64, np.int8):
        dtype = np.dtype(dtype)
        counts = np.array([5000, 70, 2049, 300], dtype=np.uint32)
        seg_vals = [make_values(rng, dtype, int(c), 9 if dtype.itemsize > 1 else 5) for c in counts]
        seg_vals[3] = np.full(300, -1, dtype=np.int64).astype(dtype)   # every row all-ones: the sentinel collision
        n = int(counts.sum())
        validity = np.full((n + 63) // 64 + 1, U64, dtype=np.uint64)
        for e in rng.choice(n - 300, 400, replace=False):    # NULL slots everywhere but in the all-ones segment
            validity[e >> 6] &= np.uint64(~(1 << (int(e) & 63)) & U64)
        for rule in (adac.RULE_APPEND, adac.RULE_RECOMPACT):
            descs = roundtrip(adac, oracle, gpu_ctx, dtype, counts, seg_vals, rule, False, validity)
            if rule == adac.RULE_APPEND and (dtype.kind == "i" or dtype.itemsize == 8):
                # defect 7: min == max == UINT64_MAX (sign-extended -1); the stored min decodes to -1
                assert int(descs[3]["width"]) == 1 and int(descs[3]["min"]) == U64 - 1
        roundtrip(adac, oracle, gpu_ctx, dtype, counts, seg_vals)   # and without NULLs


def test_block_image_batch_of_a_whole_column(adac, oracle, gpu_ctx):
    """3.3 M rows in the Appender layout (66 segments): the batched device path at a size where chunking matters."""
    rng = np.random.default_rng(11)
    n = 3_300_000
    counts = adac.appender_segment_counts(n, 4)
    vals = rng.integers(0, 1 << 21, size=n, dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    roundtrip(adac, oracle, gpu_ctx, np.uint32, counts, [vals[starts[i]:starts[i + 1]] for i in range(len(counts))])


def test_blocks_read_rejects_an_image_that_does_not_match_its_descriptor(adac, gpu_ctx):
    vals = np.arange(1000, dtype=np.uint32) * 3 + 1000
    lay = adac.Layout(gpu_ctx, np.uint32, np.array([1000], dtype=np.uint32))
    d_words = gpu_ctx.alloc(lay.max_arena_words * 8 + 128).zero()
    lay.encode(gpu_ctx.upload(vals), d_words)
    descs = lay.get_descs()
    assert int(descs[0]["width"]) == 12   # 1000 values 1000 + 3 i -> w 12, size_in_bytes 1513
    assert adac.size_in_bytes(1000, 12) == 1513
    d_blocks = gpu_ctx.alloc(adac.block_stride(1000, 12) + 64).zero()
    adac.blocks_write(gpu_ctx, np.uint32, descs, [0], d_words, d_blocks)
    wrong = descs.copy()
    wrong[0]["min"] = 999
    d_w = gpu_ctx.alloc(lay.max_arena_words * 8 + 128)
    with pytest.raises(adac.AdacError):
        adac.blocks_read(gpu_ctx, np.uint32, wrong, [0], d_blocks, d_w)
    adac.blocks_read(gpu_ctx, np.uint32, descs, [0], d_blocks, d_w)
    with pytest.raises(adac.AdacError):   # misaligned block offset
        adac.blocks_write(gpu_ctx, np.uint32, descs, [4], d_words, d_blocks)
