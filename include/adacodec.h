/*
 * adacodec.h — C ABI of libadacodec: the MI355X (gfx950) succinct column-segment codec.
 *
 * This is the drop-in boundary for ONE path of leonwind/duckdb-adaptive-compression: the per-segment
 * bit-packed integer codec behind DuckDB's COMPRESSION_SUCCINCT function table.  Every entry point names
 * the reference interface it replaces (paths relative to the reference checkout).  Plain C types only:
 * pointers named d_* are DEVICE pointers (HBM), everything else is host memory.  No entry point falls back
 * to a CPU implementation: without a usable gfx950 device the device functions return ADAC_ERR_NO_DEVICE.
 *
 * Packed format (identical to sdsl::int_vector<0>, third_party/sdsl/include/sdsl/int_vector.hpp:289-327,
 * bits.hpp:456-529): value i of a segment occupies bits [i*w, (i+1)*w) of a little-endian uint64 stream,
 * LSB first, may straddle one word boundary; stored value = (x - min) mod 2^w; bits past count*w in the last
 * word are zero.  The integration glue (C++ adapter a DuckDB maintainer would add) is in INTEGRATION.md.
 */
#ifndef ADACODEC_H
#define ADACODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAC_ABI_VERSION 1

typedef enum adac_status {
	ADAC_OK = 0,
	ADAC_ERR_INVALID_ARGUMENT = 1,
	ADAC_ERR_UNSUPPORTED_TYPE = 2, /* InternalException("Unsupported type ...") succinct.cpp:364 */
	ADAC_ERR_DEVICE = 3,           /* a HIP call failed; adac_last_error() has the text */
	ADAC_ERR_OUT_OF_MEMORY = 4,
	ADAC_ERR_NO_DEVICE = 5
} adac_status;

/* duckdb::PhysicalType codes of the eight supported types (src/include/duckdb/common/types.hpp:117-138;
 * supported set: SuccinctFun::TypeIsSupported, src/storage/compression/succinct.cpp:368-382). */
typedef enum adac_type {
	ADAC_UINT8 = 2,
	ADAC_INT8 = 3,
	ADAC_UINT16 = 4,
	ADAC_INT16 = 5,
	ADAC_UINT32 = 6,
	ADAC_INT32 = 7,
	ADAC_UINT64 = 8,
	ADAC_INT64 = 9
} adac_type;

/* Which of the reference's two encode paths produced / should produce a segment's min, max and width. */
typedef enum adac_rule {
	/* SuccinctAppendLoop + BitCompressFromSuccinct (succinct.cpp:271-306, column_segment.cpp:348-383):
	 * min/max over uint64_t(T x) (sign-extended for signed T), NULL rows excluded; w = hi(max-min)+1. */
	ADAC_RULE_APPEND = 0,
	/* BitCompressFromUncompressed (column_segment.cpp:385-456): min/max over the raw block zero-extended,
	 * NULL slots (NullValue<T>) included; max reduced by min only when max > min, so a constant non-zero
	 * segment keeps w = hi(value)+1. */
	ADAC_RULE_RECOMPACT = 1,
	/* Not an encode rule: typed min/max of the valid rows in T's own order — the segment zonemap
	 * (NumericStatistics::Update<T>, src/include/duckdb/storage/statistics/numeric_statistics.hpp:54-67, read by
	 * RowGroup::CheckZonemapSegments, src/storage/table/row_group.cpp:287-322).  adac_zonemap only. */
	ADAC_RULE_ZONEMAP = 2
} adac_rule;

#define ADAC_NO_MIN UINT64_MAX /* ColumnSegment::min_factor initial value: no frame of reference */
#define ADAC_SEG_PACKED 0x1u   /* width < 8*sizeof(T) and (x - min) stored; clear = raw slots at 8*sizeof(T) */

/* One column segment as the device sees it (32 bytes).  Mirrors the succinct members of
 * duckdb::ColumnSegment (src/include/duckdb/storage/table/column_segment.hpp:60-64,190-214):
 * succinct_vec.{data,width}, count, min_factor, compacted. */
typedef struct adac_segment_desc {
	uint64_t word_off; /* first uint64 word of the segment in the packed arena; multiple of 16 (128 B) */
	uint64_t val_off;  /* first element of the segment in the raw / decoded value buffer */
	uint64_t min;      /* min_factor, ADAC_NO_MIN if none */
	uint32_t count;    /* values in the segment */
	uint8_t width;     /* bits per stored value, 1..64 */
	uint8_t flags;     /* ADAC_SEG_* */
	uint16_t reserved;
} adac_segment_desc;

typedef struct adac_ctx adac_ctx;       /* one device + one HIP stream */
typedef struct adac_layout adac_layout; /* a batch of segments: counts, placement, device tables */

/* ---------------------------------------------------------------------------------------------
 * Host-only helpers (never touch the device).
 * ------------------------------------------------------------------------------------------- */

int adac_abi_version(void);
const char *adac_status_string(adac_status s);
const char *adac_last_error(void); /* thread-local text of the last ADAC_ERR_DEVICE */

/* SuccinctFun::TypeIsSupported (succinct.cpp:368-382) */
int adac_type_is_supported(int physical_type);
/* GetTypeIdSize for the supported types; 0 if unsupported */
uint32_t adac_type_size(int physical_type);
/* sdsl::bits::hi (bits.hpp:392-397): index of the highest set bit, hi(0) == 0 */
uint32_t adac_hi(uint64_t x);
/* Minimal width for a segment: column_segment.cpp:351-359 (rule APPEND) / :404-417 (rule RECOMPACT);
 * pad_to_byte = DBConfig::succinct_padded_to_next_byte_enabled (config.hpp:193). */
uint8_t adac_width(uint64_t min, uint64_t max, int rule, int pad_to_byte);
/* The min a PACKED descriptor stores for a segment analysed to (min, max) and planned at `width`: min itself,
 * except for the sentinel collision min == max == UINT64_MAX (every valid row is all-ones, -1 for the INT types):
 * the reference packs such a segment without subtracting and scans it without adding (UINT64_MAX doubles as "no
 * min": column_segment.cpp:371-373, succinct.cpp:138-140), returning 2^width - 1 instead of -1.  The product
 * keeps those packed bits and stores min = UINT64_MAX - (2^width - 1), which decodes them to the original value.
 * adac_plan applies this on the device; hosts that build descriptors themselves (adac_layout_set_descs) call it. */
uint64_t adac_stored_min(uint64_t min, uint64_t max, uint8_t width);

/* ceil(count*width/64): logical uint64 words of a packed segment (int_vector::capacity()/64) */
uint64_t adac_packed_words(uint64_t count, uint8_t width);
/* sdsl::size_in_bytes(succinct_vec) = 9 + 8*ceil(count*width/64) (io.hpp:636-640,
 * int_vector.hpp:602-609,1565-1578): what ColumnSegment::GetDataSize reports (column_segment.cpp:204-214) */
uint64_t adac_size_in_bytes(uint64_t count, uint8_t width);
/* words a segment occupies in the packed arena: the SDSL allocation ((bits+64)>>6 words,
 * memory_management.hpp:353) rounded up to 128 B */
uint64_t adac_arena_words(uint64_t count, uint8_t width);
/* Persistent block image of one packed segment (the reference never defined one: ConvertToPersistent is a
 * no-op for SUCCINCT, src/storage/table/column_segment.cpp:531-533).  Layout, little-endian:
 *   [0, 9 + 8*W)   the sdsl::int_vector<0> serialisation of the packed vector — uint64 bit_size, uint8 width,
 *                  W = ceil(bit_size/64) words (int_vector.hpp:602-609,1546-1578): sdsl::load() reads it back
 *   then 16 bytes  uint64 min_factor, uint8 flags (ADAC_SEG_*), uint8 physical type, 6 zero bytes
 * adac_size_in_bytes(count, width) + 16 bytes in total.  Host memory only. */
#define ADAC_BLOCK_TRAILER_BYTES 16
uint64_t adac_block_bytes(uint64_t count, uint8_t width);
/* words: the segment's ceil(count*width/64) packed words (host copy).  Returns bytes written, 0 if cap is short. */
uint64_t adac_block_write(const adac_segment_desc *desc, int physical_type, const uint64_t *words, void *out,
                          uint64_t cap);
/* Parses a block: fills desc (count, width, min, flags; word_off/val_off left 0), *physical_type, and copies
 * the packed words to words_out (capacity cap_words).  ADAC_ERR_INVALID_ARGUMENT on a malformed block. */
adac_status adac_block_read(const void *block, uint64_t len, adac_segment_desc *desc, int *physical_type,
                            uint64_t *words_out, uint64_t cap_words);

/* Header + trailer of a block image without touching its words (len: adac_block_bytes or adac_block_stride of
 * the image): what a loader needs to size the arena before adac_blocks_read moves the words. */
adac_status adac_block_peek(const void *block, uint64_t len, adac_segment_desc *desc, int *physical_type);
/* Bytes an image occupies in a block buffer: adac_block_bytes rounded up to whole 8-byte units (zero padded),
 * 8 * (packed_words + 4).  Images are placed at byte offsets that are multiples of 8. */
uint64_t adac_block_stride(uint64_t count, uint8_t width);

/* values per device tile for a type (16 KiB of decoded output) */
uint32_t adac_tile_values(int physical_type);
/* Launch-shape knobs for in-process A/B measurement: "persistent_unpack", "templated_scan", "scan_probe" (0/1),
 * "scan_tiles_per_wg" (tiles per fused-scan workgroup; 0 = chosen by type), "blocks_per_cu",
 * "num_cus", "single_pass_encode" (0 = analyze + plan + pack as three kernels).  Decoded values, packed words, widths
 * and mins never depend on them.  One knob changes WHERE adac_encode puts a segment in the arena: "encode_placement" 1
 * hands out arena space in order of completion (a cursor, no ordered look-back) instead of the exclusive prefix in
 * segment order that adac_plan computes: descriptors then carry offsets that differ from run to run (disjoint,
 * 128-byte aligned, inside max_arena_words).  "encode_stamps" 1 records diagnostic time stamps.  Returns 0 if the name
 * is known. */
int adac_set_tuning(const char *name, int value);
/* Diagnostic, not part of the drop-in boundary: the phase time stamps (8 x uint64 per segment) the single-pass encode
 * recorded while "encode_stamps" was set (tools/encode_stamps.py).  Returns 0 on success. */
int adac_debug_encode_stamps(void *host, uint64_t bytes);

/* ---------------------------------------------------------------------------------------------
 * Context and device memory.
 *
 * Threading: a context is one HIP stream and a layout caches launch tables on it, so calls that share a context
 * (or a layout) are serialised by the caller — the host mirror does it with its pool lock, the way the
 * reference serialises representation flips with bit_compression_lock (column_segment.cpp:162-167).  Different
 * contexts (one per engine worker, or one per GPU) are independent; the host-only helpers are re-entrant.
 * ------------------------------------------------------------------------------------------- */

/* gfx950 devices visible to this process (hipGetDeviceCount); 0 when there is none or the runtime is unusable.
 * A host builds one segment pool per device from it. */
int adac_device_count(void);

/* Binds to HIP device `device`.  external_stream: a hipStream_t owned by the caller (e.g. the engine's or
 * torch's current stream) or NULL to create a private non-blocking stream. */
adac_status adac_ctx_create(int device, void *external_stream, adac_ctx **out);
/* Layouts, plans and graphs hold a reference on their context: adac_ctx_destroy drops the creator's reference and
 * the stream goes away with the last object made on it, so destruction order does not matter.  Raw device memory
 * (adac_dev_alloc) is not tracked: free it before the context. */
void adac_ctx_destroy(adac_ctx *ctx);
adac_status adac_ctx_sync(adac_ctx *ctx);
void *adac_ctx_stream(adac_ctx *ctx); /* the hipStream_t every launch of this ctx goes to */
int adac_ctx_device(adac_ctx *ctx);

adac_status adac_dev_alloc(adac_ctx *ctx, size_t bytes, void **d_ptr);
adac_status adac_dev_free(adac_ctx *ctx, void *d_ptr);
adac_status adac_dev_memset(adac_ctx *ctx, void *d_ptr, int byte, size_t bytes);   /* stream-ordered */
adac_status adac_memcpy_h2d(adac_ctx *ctx, void *d_dst, const void *src, size_t bytes); /* blocking */
adac_status adac_memcpy_d2h(adac_ctx *ctx, void *dst, const void *d_src, size_t bytes); /* blocking */

/* Device-to-host copy enqueued on the context's stream without waiting for it (dst should be page-locked,
 * adac_host_alloc_pinned, for the copy to overlap host work); adac_ctx_sync makes the bytes visible. */
adac_status adac_memcpy_d2h_async(adac_ctx *ctx, void *dst, const void *d_src, size_t bytes);
/* Page-locked host memory for staging (hipHostMalloc): D2H/H2D copies of it run at PCIe line rate. */
adac_status adac_host_alloc_pinned(adac_ctx *ctx, size_t bytes, void **ptr);
adac_status adac_host_free_pinned(adac_ctx *ctx, void *ptr);

/* A marker in a context's stream: adac_event_record enqueues it, adac_event_wait blocks the calling host thread until
 * everything enqueued before it has finished (and only that: later work on the same stream is not waited for, which
 * is what lets a host keep several decode + copy batches in flight on one stream).  Any thread may wait. */
typedef struct adac_event adac_event;
adac_status adac_event_record(adac_ctx *ctx, adac_event **out);
adac_status adac_event_wait(adac_event *ev);
int adac_event_done(adac_event *ev); /* 1 finished, 0 still running (or error) */
void adac_event_destroy(adac_event *ev);

/* HIP-event stopwatch on the ctx stream (for measuring kernels where they are launched). */
adac_status adac_timer_start(adac_ctx *ctx);
adac_status adac_timer_stop(adac_ctx *ctx, float *elapsed_ms); /* records, synchronises, reads */

/* ---------------------------------------------------------------------------------------------
 * Layout: a batch of segments of one physical type.
 * ------------------------------------------------------------------------------------------- */

/* counts[i]: values in segment i (ColumnSegment::count).  val_offs[i]: element offset of segment i in the
 * value buffer handed to analyze/pack/unpack, or NULL for back-to-back placement.  Builds the device tile
 * table; descriptors start with width = 8*sizeof(T), min = ADAC_NO_MIN, not packed (the state of a fresh
 * transient SUCCINCT segment, column_segment.cpp:101-105). */
adac_status adac_layout_create(adac_ctx *ctx, int physical_type, const uint32_t *counts, const uint64_t *val_offs,
                               uint64_t nseg, adac_layout **out);
void adac_layout_destroy(adac_layout *l);
uint64_t adac_layout_nseg(const adac_layout *l);
uint64_t adac_layout_ntiles(const adac_layout *l);
uint64_t adac_layout_total_values(const adac_layout *l);
/* elements the value buffer must hold: max(val_off + count) */
uint64_t adac_layout_value_span(const adac_layout *l);
/* upper bound of the packed arena in uint64 words for ANY widths (every segment unpacked) */
uint64_t adac_layout_max_arena_words(const adac_layout *l);
/* Upload host descriptors (decode of segments encoded elsewhere). word_off must be a multiple of 16. */
adac_status adac_layout_set_descs(adac_layout *l, const adac_segment_desc *descs);
/* Download the device descriptors (after adac_plan / adac_encode).  Synchronises the stream. */
adac_status adac_layout_get_descs(adac_layout *l, adac_segment_desc *descs);
/* Download per-segment {min,max} pairs (2*nseg uint64) left by adac_analyze.  Synchronises. */
adac_status adac_layout_get_minmax(adac_layout *l, uint64_t *minmax);
/* The device copy of the descriptor table (nseg entries), for callers that chain their own kernels. */
const adac_segment_desc *adac_layout_device_descs(const adac_layout *l);

/* ---------------------------------------------------------------------------------------------
 * Hot path.  All calls enqueue on the ctx stream and return without synchronising.
 * ------------------------------------------------------------------------------------------- */

/* Per-segment min/max.  Replaces the running min/max of SuccinctAppendLoop (succinct.cpp:271-306; rule
 * APPEND) and pass 1 of BitCompressFromUncompressed (column_segment.cpp:390-400; rule RECOMPACT).
 * d_vals: raw values of type T at element offsets val_off.  d_validity: DuckDB validity mask over the same
 * element index space (bit e of word e/64 set = row valid) or NULL = all valid. */
adac_status adac_analyze(adac_layout *l, const void *d_vals, const uint64_t *d_validity, int rule);

/* Segment zonemaps: zonemap[2*s] / [2*s+1] = minimum / maximum of segment s over its valid rows, as bit patterns
 * of T (zero-extended), ordered as T orders them (signed for the INT types).  A segment without a valid row
 * reports min = T's maximum and max = T's minimum (an empty interval).  Overwrites the layout's min/max
 * scratch (run it before or after an encode, not between adac_analyze and adac_plan).  Synchronises. */
adac_status adac_zonemap(adac_layout *l, const void *d_vals, const uint64_t *d_validity, uint64_t *zonemap);

/* From the min/max left by adac_analyze compute every segment's width, flags, stored min and arena offset
 * on the device (no host round trip).  Replaces the width decision of column_segment.cpp:351-363 /
 * :404-420.  A segment whose width would not shrink (8*sizeof(T) <= w) stays unpacked. */
adac_status adac_plan(adac_layout *l, int rule, int pad_to_byte);

/* Bit-pack every segment: words[word_off + ...] <- (x - min) mod 2^w.  Replaces the pack loops of
 * BitCompressFromSuccinct (column_segment.cpp:365-376) and BitCompressFromUncompressed (:426-443).
 * NULL slots are stored as NullValue<T> - min (succinct.cpp:288-291, null_value.hpp:26-28). */
adac_status adac_pack(adac_layout *l, const void *d_vals, const uint64_t *d_validity, uint64_t *d_words);

/* analyze + plan + pack */
adac_status adac_encode(adac_layout *l, const void *d_vals, const uint64_t *d_validity, int rule, int pad_to_byte,
                        uint64_t *d_words);

/* Re-compaction packed -> packed (the `adac_repack(words, n, old_w, new_w, min)` of SURVEY.md §8b; traffic
 * n (old_w + new_w) / 8 bytes per segment, §8d).  The reference's BitCompressFromSuccinct
 * (column_segment.cpp:348-383) re-packs in place from full-width slots; here the source is any encoded form
 * (padded, wider than needed after the value range shrank, or unpacked slots) and the destination is a second
 * layout over the SAME segments (type, counts, value offsets), out of place:
 *   adac_analyze_packed  min/max of the decoded values under `rule`, left in dst for adac_plan(dst, ...);
 *   adac_repack          decode at src's width/min, subtract dst's min, pack at dst's width into d_dst_words;
 *   adac_reencode        analyze_packed + adac_plan(dst) + repack.
 * The result is bit-identical to adac_encode of the decoded values.  d_validity as for adac_analyze / adac_pack. */
adac_status adac_analyze_packed(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity, int rule,
                                adac_layout *dst);
adac_status adac_repack(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity, adac_layout *dst,
                        uint64_t *d_dst_words);
adac_status adac_reencode(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity, int rule,
                          int pad_to_byte, adac_layout *dst, uint64_t *d_dst_words);

/* Decode every segment: out[val_off + i] = T(read_int(words, i*w, w) + min).  Replaces SuccinctScanPartial
 * over a whole column (succinct.cpp:123-144) and ColumnSegment::UncompressSuccinct (column_segment.cpp:458-506).
 * Unpacked segments are copied without the min add (SURVEY.md §8a parity domain (iii)).
 * d_out must be 16-byte aligned. */
adac_status adac_unpack(adac_layout *l, const uint64_t *d_words, void *d_out);

/* Decode `count` values of ONE segment starting at row `start` to d_out[out_off ...]: the scan_vector /
 * scan_partial slots (compression_function.hpp:84-88; succinct.cpp:123-144,232-240). */
adac_status adac_unpack_range(adac_layout *l, const uint64_t *d_words, uint64_t seg, uint64_t start, uint64_t count,
                              void *d_out, uint64_t out_off);

/* The same slots for SEVERAL segments in one launch and WITHOUT a layout: every job names its segment inline (where
 * its words start in d_words, width, min, flags — the fields of adac_segment_desc) and a row range; rows
 * [start, start + count) are decoded to d_out[out_off ...).  The jobs travel in the kernel arguments, so the call
 * allocates nothing and uploads nothing: it is what a host mirror of ColumnData::ScanVector
 * (src/storage/table/column_data.cpp:92-139) uses to serve one vector, and to decode the next few segments of a scan
 * ahead of the consumer.  Any njobs (launched in groups of 48).  out_off * sizeof(T) need not be aligned. */
typedef struct adac_unpack_job {
	uint64_t word_off; /* first uint64 word of the segment in d_words (multiple of 16) */
	uint64_t min;      /* the descriptor's min (ADAC_NO_MIN if none) */
	uint64_t out_off;  /* element offset of the first decoded row in d_out */
	uint32_t start;    /* first row */
	uint32_t count;    /* rows to decode */
	uint8_t width;     /* 1..8*sizeof(T) */
	uint8_t flags;     /* ADAC_SEG_* */
	uint16_t reserved;
	uint32_t reserved2;
} adac_unpack_job;
adac_status adac_unpack_jobs(adac_ctx *ctx, int physical_type, const adac_unpack_job *jobs, uint64_t njobs,
                             const uint64_t *d_words, void *d_out);

/* Point fetch: d_out[k] = value at row d_rows[k] of segment d_segs[k] — the intended semantics of
 * SuccinctFetchRow (succinct.cpp:244-260; the reference implementation ignores row_id). */
adac_status adac_fetch_rows(adac_layout *l, const uint64_t *d_words, const uint32_t *d_segs, const uint32_t *d_rows,
                            uint64_t n, void *d_out);

/* Fused scan + aggregate without materialising: d_sums[seg] = sum of the decoded values, each widened to 64 bits
 * according to T's signedness (sign-extended for the INT types, zero-extended for the UINT types), mod 2^64 —
 * the low 64 bits of SQL SUM over the segment.  (reads packed bytes only.) */
adac_status adac_scan_sum(adac_layout *l, const uint64_t *d_words, uint64_t *d_sums);

/* Fused scan + equality filter: d_counts[seg] = number of rows whose decoded value == key (key given as the
 * bit pattern of T zero-extended) — the `SELECT i FROM t1 WHERE i == k` look-ups of
 * benchmark/micro/succinct/zipf_distribution.cpp:40-48 without materialising the column. */
adac_status adac_scan_count_eq(adac_layout *l, const uint64_t *d_words, uint64_t key, uint64_t *d_counts);

/* Fused scan + range filter: d_counts[seg] = rows with lo <= value <= hi in T's own order (signed for the INT
 * types); lo / hi are bit patterns of T, zero-extended.  With lo = T's minimum or hi = T's maximum this is
 * `<=` / `>=`, with lo == hi it is `==`: the comparison kinds ColumnSegment::FilterSelection pushes down
 * (src/storage/table/column_segment.cpp:575-844).  Evaluated on the packed fields (lo - min, hi - min); a
 * segment whose [min, min + 2^w) cannot intersect the range is skipped without reading it (zonemap skip). */
adac_status adac_scan_count_between(adac_layout *l, const uint64_t *d_words, uint64_t lo, uint64_t hi,
                                    uint64_t *d_counts);

/* Grouped aggregate over TWO packed columns of one table — the shape of TPC-H Q1 on the reference's config 3
 * (`SELECT key, SUM(value), COUNT(*) ... GROUP BY key`, benchmark log TPCH_runtime.txt:2-6; the reference decodes both
 * columns vector by vector, succinct.cpp:123-144, and feeds a hash aggregate).  `values` and `keys` are layouts on the
 * same context with the same row count per segment (types, widths, placements may differ); nothing is materialised.
 * key = the key column's value as an unsigned number of its own width.  d_sums[g], d_counts[g] for g < ngroups are the
 * sum (values widened to 64 bits by the value type's signedness, mod 2^64 — adac_scan_sum's rule) and the number of
 * the rows whose key is g; entry [ngroups] collects the rows whose key is >= ngroups, so both arrays hold ngroups + 1
 * entries.  1 <= ngroups <= 256. */
adac_status adac_scan_group_sum(adac_layout *values, const uint64_t *d_value_words, adac_layout *keys,
                                const uint64_t *d_key_words, uint32_t ngroups, uint64_t *d_sums, uint64_t *d_counts);

/* The same two scans with a DuckDB validity mask over the element index space (bit e of word e/64 set = row e
 * valid, as for adac_analyze): NULL rows take no part in the aggregate — what SUM / COUNT over a nullable
 * column mean.  d_validity == NULL is the unmasked scan. */
adac_status adac_scan_sum_valid(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity, uint64_t *d_sums);
adac_status adac_scan_count_between_valid(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity,
                                          uint64_t lo, uint64_t hi, uint64_t *d_counts);

/* Filter push-down with a selection result — ColumnSegment::FilterSelection (column_segment.cpp:575-844) on the
 * packed bytes: bit e of d_bitmap (element index e = val_off + row, the same index space as the validity mask) is
 * set iff the row is valid (d_validity, NULL = every row) and lo <= value <= hi in T's own order; d_counts[seg] =
 * rows selected.  d_bitmap has ceil(value_span / 64) words, is cleared by the call and must not alias
 * d_validity.  Because the result has the validity mask's layout it can be handed to the next column's scan
 * (adac_scan_select_between again for a conjunction, adac_scan_sum_valid / adac_scan_count_between_valid for the
 * aggregate) when the columns share their value offsets: a multi-column filter + aggregate (TPC-H Q6's shape)
 * that never materialises a value. */
adac_status adac_scan_select_between(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity, uint64_t lo,
                                     uint64_t hi, uint64_t *d_bitmap, uint64_t *d_counts);

/* Materialise only the selected rows: d_out receives, densely and in row order (segment by segment), the values of
 * the rows whose bit is set in d_bitmap (the result of adac_scan_select_between, or any mask over the element index
 * space); d_out_ids, if not NULL, receives their element indices (val_off + row).  *total_out, if not NULL, is set
 * to the number of rows written (this makes the call synchronous); d_out must hold at least that many values —
 * the sum of the select's d_counts, or total_values in the worst case.  The scan-with-selection half of
 * ColumnSegment::FilterSelection (column_segment.cpp:575-844): a tile is decoded once, rows that did not pass are
 * never written. */
adac_status adac_unpack_selected(adac_layout *l, const uint64_t *d_words, const uint64_t *d_bitmap, void *d_out,
                                 uint64_t *d_out_ids, uint64_t *total_out);

/* ---------------------------------------------------------------------------------------------
 * Persistence in HBM: the block images of MANY packed segments built / parsed by one kernel, so
 * that a checkpoint is one device pass + one device-to-host copy (and a load one host-to-device copy + one pass)
 * instead of a copy per segment.  This is the ConvertToPersistent the reference leaves empty for SUCCINCT
 * (src/storage/table/column_segment.cpp:529-533; the checkpoint-side Compress / FinalizeCompress slots,
 * src/storage/compression/succinct.cpp:91-119, never reach a block).  descs[i]: the segment (word_off into d_words,
 * count, width, min, flags); block_offs[i]: byte offset (multiple of 8) of its image in d_blocks, which spans
 * adac_block_stride(count, width) bytes; the first adac_block_bytes of them are exactly what adac_block_write
 * produces (sdsl::int_vector<0>::serialize + trailer).  Both calls synchronise the stream.
 * adac_blocks_read also zeroes the segment's arena words past its packed words and returns
 * ADAC_ERR_INVALID_ARGUMENT when an image's header does not match its descriptor (build descs with adac_block_peek).
 * ------------------------------------------------------------------------------------------- */
adac_status adac_blocks_write(adac_ctx *ctx, int physical_type, const adac_segment_desc *descs,
                              const uint64_t *block_offs, uint64_t nseg, const uint64_t *d_words, void *d_blocks);
adac_status adac_blocks_read(adac_ctx *ctx, int physical_type, const adac_segment_desc *descs,
                             const uint64_t *block_offs, uint64_t nseg, const void *d_blocks, uint64_t *d_words);

/* ---------------------------------------------------------------------------------------------
 * DuckDB BITPACKING segments — the persistent counterpart of the succinct codec, decode side.
 * A segment is the block image DuckDB's checkpoint writes (src/storage/compression/bitpacking.cpp:357-538):
 * metadata groups of 2048 rows in CONSTANT / CONSTANT_DELTA / DELTA_FOR / FOR mode, packed with fastpforlib.
 * Replaces BitpackingScanPartial / BitpackingScan (:736-826) and BitpackingFetchRow (:827-870).
 * ------------------------------------------------------------------------------------------- */
typedef struct adac_bp_layout adac_bp_layout;

/* block_offs[i]: byte offset (multiple of 16) of segment i's block inside the device buffer handed to the scan
 * calls — blocks must be followed by at least 16 readable bytes (a whole Storage::BLOCK_SIZE image is);
 * counts[i]: rows of segment i; out_offs[i]: element offset of its first row in the output, NULL = back to back. */
adac_status adac_bp_layout_create(adac_ctx *ctx, int physical_type, const uint64_t *block_offs, const uint32_t *counts,
                                  const uint64_t *out_offs, uint64_t nseg, adac_bp_layout **out);
void adac_bp_layout_destroy(adac_bp_layout *l);
uint64_t adac_bp_layout_ngroups(const adac_bp_layout *l);
uint64_t adac_bp_layout_total_values(const adac_bp_layout *l);
/* Parse every group's header (mode, width, frame of reference, payload position) out of the block images into
 * the layout's device table — the analogue of BitpackingScanState::LoadNextGroup (bitpacking.cpp:597-640) for all
 * groups at once.  adac_bp_unpack binds by itself when handed a different buffer; call this again after
 * rewriting blocks in place. */
adac_status adac_bp_bind(adac_bp_layout *l, const void *d_blocks);
/* Full scan: every row of every segment to d_out (16-byte aligned), one workgroup per 2048-row metadata group. */
adac_status adac_bp_unpack(adac_bp_layout *l, const void *d_blocks, void *d_out);
/* Rows [start, start + count) of segment `seg` into d_out[out_off ...] — BitpackingScanPartial / BitpackingScan
 * (bitpacking.cpp:736-826) for any start: a DELTA_FOR group that the range enters in the middle is prefix-summed
 * from its first row, as the reference's Skip + LoadNextGroup do. */
adac_status adac_bp_unpack_range(adac_bp_layout *l, const void *d_blocks, uint64_t seg, uint64_t start, uint64_t count,
                                 void *d_out, uint64_t out_off);

/* d_out[k] = row d_rows[k] of segment d_segs[k] */
adac_status adac_bp_fetch_rows(adac_bp_layout *l, const void *d_blocks, const uint32_t *d_segs, const uint32_t *d_rows,
                               uint64_t n, void *d_out);

/* Compress side (BitpackingCompress / BitpackingFinalizeCompress, bitpacking.cpp:514-538): per-group statistics
 * on the device, the mode decision of BitpackingState::Flush (:229-294) and the sequential placement of groups
 * into Storage::BLOCK_SIZE blocks (:453-512) on the host, then one device pass writing every group image.
 * force_mode: BitpackingMode (0 AUTO, 1 CONSTANT, 2 CONSTANT_DELTA, 3 DELTA_FOR, 4 FOR — the reference's
 * force_bitpacking_mode).  NULL rows are encoded as the value 0 (the reference leaves them indeterminate). */
typedef struct adac_bp_plan adac_bp_plan;
adac_status adac_bp_plan_create(adac_ctx *ctx, int physical_type, const void *d_vals, const uint64_t *d_validity,
                                uint64_t n, int force_mode, adac_bp_plan **out);
void adac_bp_plan_destroy(adac_bp_plan *p);
/* 0 when some group can be stored in no mode (Flush() == false: BitpackingFinalAnalyze returns INVALID_INDEX) */
int adac_bp_plan_encodable(const adac_bp_plan *p);
uint64_t adac_bp_plan_nseg(const adac_bp_plan *p);
uint64_t adac_bp_plan_groups_by_mode(const adac_bp_plan *p, int mode);
/* first row, row count and used bytes (FlushSegment's total_segment_size) of segment i */
adac_status adac_bp_plan_segment(const adac_bp_plan *p, uint64_t i, uint64_t *start, uint64_t *count,
                                 uint64_t *total_size);
/* Write every segment's block image: segment i at d_blocks + i*block_stride (block_stride >= 262136, multiple of 16).
 * The whole nseg*block_stride range is rewritten (unused bytes zero). */
adac_status adac_bp_write(adac_bp_plan *p, const void *d_vals, const uint64_t *d_validity, void *d_blocks,
                          uint64_t block_stride);

#ifdef __cplusplus
}
#endif
#endif /* ADACODEC_H */
