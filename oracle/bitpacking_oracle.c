/*
 * bitpacking_oracle.c — CPU restatement of DuckDB's on-disk BITPACKING codec as shipped in the reference
 * (the persistent counterpart of the succinct codec).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as succinct_oracle.c): the product never links or calls it.
 *
 * Restated from (paths relative to /root/reference):
 *   BitpackingState::{Reset,Update,CalculateFORStats,CalculateDeltaStats,Flush}   src/storage/compression/bitpacking.cpp:91-318
 *   BitpackingCompressState + BitpackingWriter (block layout, metadata growing down, FlushSegment)  :357-538
 *   BitpackingScanState::LoadNextGroup / BitpackingScanPartial / BitpackingFetchRow                 :583-862
 *   BitpackingPrimitives (MinimumBitWidth, GetEffectiveWidth, GetRequiredSize, PackBuffer)  src/include/duckdb/common/bitpacking.hpp
 *   TrySubtractOperator                                                          src/function/scalar/operators/subtract.cpp:82-160
 *   bit layout of a 32-value algorithm group: duckdb_fastpforlib::fastpack (third_party/fastpforlib/bitpacking.cpp):
 *   values masked to `width` bits, LSB first, contiguous over 8/16/32-bit little-endian words — i.e. one
 *   contiguous little-endian bit stream; checked against the real fastpforlib sources by oracle/_ref (see
 *   oracle/Makefile, tests/test_bitpacking_oracle.py) and the vectors committed from it.
 *
 * Pinning: the reference's tests for this codec are SQL-level only (the .test files under test/sql/storage/compression/bitpacking/);
 * there are no byte-level fixtures, so block images are "pinned by restatement" except for the fastpforlib layer.
 * Known indeterminacy of the reference, fixed here: rows that are NULL keep whatever the (uninitialised)
 * compression buffer held (bitpacking.cpp:306-309); this restatement starts from an all-zero buffer.
 *
 * Types are handled as bit patterns in uint64_t with (type_size, is_signed); every arithmetic step is
 * truncated to the type's width, so results equal the templates' T / T_S / T_U arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { BP_GROUP = 2048, BP_ALGO = 32, BP_HEADER = 8, BP_BLOCK = 262144 - 8 };
enum { MODE_AUTO = 0, MODE_CONSTANT = 1, MODE_CONSTANT_DELTA = 2, MODE_DELTA_FOR = 3, MODE_FOR = 4 }; /* storage/compression/bitpacking.hpp:15-22 */

typedef struct {
	unsigned ts; /* type size in bytes */
	int sg;      /* signed */
	unsigned bits;
	uint64_t mask;
} ty;

static ty mk_ty(unsigned ts, int sg) {
	ty t = {ts, sg, ts * 8, ts == 8 ? ~0ULL : ((1ULL << (ts * 8)) - 1)};
	return t;
}
static inline uint64_t trunc_t(ty t, uint64_t x) { return x & t.mask; }
/* value of the bit pattern as int64 (sign-extended for signed types, zero-extended otherwise) */
static inline int64_t as_s64(ty t, uint64_t x) {
	x &= t.mask;
	if (t.ts == 8) return (int64_t)x;
	uint64_t sb = 1ULL << (t.bits - 1);
	return (int64_t)((x ^ sb) - sb);
}
/* T-ordered comparison */
static inline int lt_t(ty t, uint64_t a, uint64_t b) {
	if (t.sg) return as_s64(t, a) < as_s64(t, b);
	return (a & t.mask) < (b & t.mask);
}
static inline uint64_t t_max(ty t) { return t.sg ? (t.mask >> 1) : t.mask; }
static inline uint64_t t_min(ty t) { return t.sg ? trunc_t(t, ~(t.mask >> 1)) : 0; }

/* TrySubtractOperator::Operation for T (subtract.cpp:82-160) */
static int try_sub(ty t, uint64_t l, uint64_t r, uint64_t *res) {
	if (!t.sg) {
		if ((r & t.mask) > (l & t.mask)) return 0;
		*res = trunc_t(t, l - r);
		return 1;
	}
	if (t.ts == 8) {
		int64_t o;
		if (__builtin_sub_overflow((int64_t)l, (int64_t)r, &o)) return 0;
		*res = (uint64_t)o;
		return 1;
	}
	int64_t d = as_s64(t, l) - as_s64(t, r);
	if (d < as_s64(t, t_min(t)) || d > as_s64(t, t_max(t))) return 0;
	*res = trunc_t(t, (uint64_t)d);
	return 1;
}

/* BitpackingPrimitives::GetEffectiveWidth (bitpacking.hpp:208-216) */
static unsigned eff_width(ty t, unsigned w) { return (w + t.ts > t.bits) ? t.bits : w; }

/* FindMinimumBitWidth<T>(value, value) for an UNSIGNED view (T_U) — bitpacking.hpp:131-176 */
static unsigned min_width_unsigned(ty t, uint64_t v) {
	v &= t.mask;
	if (v == 0) return 0;
	unsigned w = 0;
	while (v) {
		w++;
		v >>= 1;
	}
	return eff_width(t, w);
}
/* ... and for the signed T itself (min == max == v) */
static unsigned min_width_signed(ty t, uint64_t v) {
	if ((v & t.mask) == t_min(t)) return t.bits;
	int64_t s = as_s64(t, v);
	uint64_t mag = (uint64_t)(s < 0 ? -s : s);
	if (mag == 0) return 0;
	unsigned w = 1;
	while (mag) {
		w++;
		mag >>= 1;
	}
	return eff_width(t, w);
}

static uint64_t required_size(uint64_t count, unsigned w) { /* GetRequiredSize, bitpacking.hpp:99-102 */
	uint64_t c = (count + BP_ALGO - 1) / BP_ALGO * BP_ALGO;
	return c * w / 8;
}

static void store_t(uint8_t *p, ty t, uint64_t v) { memcpy(p, &v, t.ts); }
static uint64_t load_t(const uint8_t *p, ty t) {
	uint64_t v = 0;
	memcpy(&v, p, t.ts);
	return v;
}

/* duckdb_fastpforlib::fastpack of ONE 32-value algorithm group: masked fields, one contiguous LE bit stream */
ORC_API void bp_pack_group(const uint64_t *vals /* 32 bit patterns */, unsigned w, uint8_t *dst /* 4*w bytes */) {
	memset(dst, 0, 4 * w);
	if (w == 0) return;
	uint64_t m = w >= 64 ? ~0ULL : ((1ULL << w) - 1);
	for (unsigned i = 0; i < BP_ALGO; i++) {
		uint64_t v = vals[i] & m;
		uint64_t bit = (uint64_t)i * w;
		for (unsigned b = 0; b < w; b += 8) { /* byte-wise OR keeps it endian-explicit */
			uint64_t pos = bit + b;
			unsigned sh = (unsigned)(pos & 7);
			uint64_t chunk = (v >> b) & 0xff;
			dst[pos >> 3] |= (uint8_t)(chunk << sh);
			if (sh && (pos >> 3) + 1 < 4 * (uint64_t)w) dst[(pos >> 3) + 1] |= (uint8_t)(chunk >> (8 - sh));
		}
	}
}

ORC_API void bp_unpack_group(const uint8_t *src, unsigned w, uint64_t *vals /* 32 */) {
	for (unsigned i = 0; i < BP_ALGO; i++) {
		uint64_t v = 0;
		uint64_t bit = (uint64_t)i * w;
		for (unsigned b = 0; b < w; b++) {
			uint64_t pos = bit + b;
			v |= (uint64_t)((src[pos >> 3] >> (pos & 7)) & 1) << b;
		}
		vals[i] = v;
	}
}

/* ------------------------------------------------------------------------------------------------
 * Compress: a column -> a list of segments (block images)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
	uint8_t *block;      /* BP_BLOCK bytes */
	uint64_t start;      /* first row */
	uint64_t count;      /* rows */
	uint64_t total_size; /* bytes used after FlushSegment */
} bp_segment;

typedef struct {
	ty t;
	int force_mode;
	int null_zero; /* NULL rows take the value 0 instead of keeping the buffer's stale content */
	/* BitpackingState */
	uint64_t buf_internal[BP_GROUP + 1];
	uint64_t *buf; /* = buf_internal + 1 */
	uint64_t delta[BP_GROUP];
	uint8_t valid[BP_GROUP];
	uint64_t idx;
	uint64_t minimum, maximum, min_max_diff;
	uint64_t min_delta, max_delta, min_max_delta_diff, delta_offset; /* T_S bit patterns */
	int all_valid, all_invalid, can_do_delta, can_do_for;
	/* compress state */
	bp_segment *segs;
	uint64_t nseg, cap;
	uint8_t *base, *data_ptr, *meta_ptr;
	uint64_t rows_done;
	uint64_t groups_by_mode[6];
	int failed;
} bp_state;

static void st_reset(bp_state *s) { /* bitpacking.cpp:131-145 */
	s->minimum = t_max(s->t);
	s->min_delta = t_max(mk_ty(s->t.ts, 1));
	s->maximum = t_min(s->t);
	s->max_delta = t_min(mk_ty(s->t.ts, 1));
	s->delta_offset = 0;
	s->all_valid = 1;
	s->all_invalid = 1;
	s->can_do_delta = 0;
	s->can_do_for = 0;
	s->idx = 0;
	s->min_max_diff = 0;
	s->min_max_delta_diff = 0;
}

static void create_segment(bp_state *s, uint64_t row_start) { /* CreateEmptySegment :462-474 */
	if (s->nseg == s->cap) {
		s->cap = s->cap ? s->cap * 2 : 8;
		s->segs = (bp_segment *)realloc(s->segs, s->cap * sizeof(bp_segment));
	}
	bp_segment *g = &s->segs[s->nseg++];
	g->block = (uint8_t *)calloc(1, BP_BLOCK);
	g->start = row_start;
	g->count = 0;
	g->total_size = 0;
	s->base = g->block;
	s->data_ptr = g->block + BP_HEADER;
	s->meta_ptr = g->block + BP_BLOCK;
}

static void flush_segment(bp_state *s) { /* FlushSegment :496-512 */
	bp_segment *g = &s->segs[s->nseg - 1];
	uint64_t metadata_offset = ((uint64_t)(s->data_ptr - s->base) + 7) & ~7ULL; /* AlignValue */
	uint64_t metadata_size = (uint64_t)(s->base + BP_BLOCK - s->meta_ptr);
	memmove(s->base + metadata_offset, s->meta_ptr, metadata_size);
	uint64_t first = metadata_offset + metadata_size;
	memcpy(s->base, &first, 8);
	g->total_size = metadata_offset + metadata_size;
}

static void reserve(bp_state *s, uint64_t data_bytes) { /* ReserveSpace + FlushAndCreateSegmentIfFull */
	uint64_t need = data_bytes + 4;
	if ((uint64_t)(s->meta_ptr - s->data_ptr) < need) {
		bp_segment *g = &s->segs[s->nseg - 1];
		uint64_t row_start = g->start + g->count;
		flush_segment(s);
		create_segment(s, row_start);
	}
}

static void write_meta(bp_state *s, int mode) { /* WriteMetaData :447-451 */
	uint32_t enc = (uint32_t)(s->data_ptr - s->base) | ((uint32_t)mode << 24);
	s->meta_ptr -= 4;
	memcpy(s->meta_ptr, &enc, 4);
	s->groups_by_mode[mode]++;
}

static void write_t(bp_state *s, uint64_t v) {
	store_t(s->data_ptr, s->t, v);
	s->data_ptr += s->t.ts;
}

static void pack_buffer(bp_state *s, const uint64_t *vals, uint64_t count, unsigned w) { /* PackBuffer<T,false> */
	uint64_t tmp[BP_ALGO];
	uint64_t full = count - count % BP_ALGO;
	for (uint64_t i = 0; i < full; i += BP_ALGO) bp_pack_group(vals + i, w, s->data_ptr + i * w / 8);
	if (count % BP_ALGO) {
		memset(tmp, 0, sizeof(tmp)); /* the reference leaves the tail of tmp_buffer uninitialised */
		memcpy(tmp, vals + full, (count - full) * 8);
		bp_pack_group(tmp, w, s->data_ptr + full * w / 8);
	}
}

static int st_flush(bp_state *s) { /* BitpackingState::Flush :229-294 with BitpackingWriter */
	ty t = s->t;
	ty ts_ = mk_ty(t.ts, 1);
	uint64_t n = s->idx;
	if (n == 0) return 1;
	bp_segment *g;
	if ((s->all_invalid || s->maximum == s->minimum) && (s->force_mode == MODE_AUTO || s->force_mode == MODE_CONSTANT)) {
		reserve(s, t.ts);
		write_meta(s, MODE_CONSTANT);
		write_t(s, s->maximum);
		g = &s->segs[s->nseg - 1];
		g->count += n;
		return 1;
	}
	/* CalculateFORStats */
	s->can_do_for = try_sub(t, s->maximum, s->minimum, &s->min_max_diff);
	/* CalculateDeltaStats :150-211 */
	do {
		if (!t.sg && (s->maximum & t.mask) > t_max(ts_)) break; /* maximum > (T)NumericLimits<T_S>::Maximum() */
		if (n < 2) break;
		if (!s->all_valid) break;
		int can_do_all = 1;
		if (t.sg) {
			uint64_t bogus;
			can_do_all = try_sub(ts_, s->minimum, s->maximum, &bogus) && try_sub(ts_, s->maximum, s->minimum, &bogus);
		}
		int ok = 1;
		if (can_do_all) {
			for (uint64_t i = 0; i < n; i++) s->delta[i] = trunc_t(t, s->buf[i] - s->buf[(int64_t)i - 1]);
		} else {
			for (uint64_t i = 0; i < n && ok; i++) ok = try_sub(ts_, s->buf[i], s->buf[(int64_t)i - 1], &s->delta[i]);
			if (!ok) break;
		}
		s->can_do_delta = 1;
		for (uint64_t i = 1; i < n; i++) {
			if (lt_t(ts_, s->max_delta, s->delta[i])) s->max_delta = s->delta[i];
			if (lt_t(ts_, s->delta[i], s->min_delta)) s->min_delta = s->delta[i];
		}
		s->delta[0] = s->min_delta;
		s->can_do_delta = s->can_do_delta && try_sub(ts_, s->max_delta, s->min_delta, &s->min_max_delta_diff);
		s->can_do_delta = s->can_do_delta && try_sub(ts_, s->buf[0], s->min_delta, &s->delta_offset);
	} while (0);

	if (s->can_do_delta) {
		if (s->max_delta == s->min_delta && s->force_mode != MODE_FOR && s->force_mode != MODE_DELTA_FOR) {
			reserve(s, 2 * t.ts);
			write_meta(s, MODE_CONSTANT_DELTA);
			write_t(s, s->buf[0]);    /* frame_of_reference = compression_buffer[0] */
			write_t(s, s->max_delta); /* constant */
			g = &s->segs[s->nseg - 1];
			g->count += n;
			return 1;
		}
		unsigned dw = min_width_unsigned(t, s->min_max_delta_diff);
		unsigned rw = t.sg ? min_width_signed(t, s->min_max_diff) : min_width_unsigned(t, s->min_max_diff);
		if (dw < rw && s->force_mode != MODE_FOR) {
			for (uint64_t i = 0; i < n; i++) s->delta[i] = trunc_t(t, s->delta[i] - s->min_delta); /* SubtractFrameOfReference */
			uint64_t bp = required_size(n, dw);
			reserve(s, bp + 3 * t.ts);
			write_meta(s, MODE_DELTA_FOR);
			write_t(s, s->min_delta); /* frame_of_reference */
			write_t(s, dw);
			write_t(s, s->delta_offset);
			pack_buffer(s, s->delta, n, dw);
			s->data_ptr += bp;
			g = &s->segs[s->nseg - 1];
			g->count += n;
			return 1;
		}
	}
	if (s->can_do_for) {
		unsigned w = min_width_unsigned(t, s->min_max_diff);
		for (uint64_t i = 0; i < n; i++) s->buf[i] = trunc_t(t, s->buf[i] - s->minimum);
		uint64_t bp = required_size(n, w);
		reserve(s, bp + 2 * t.ts);
		write_meta(s, MODE_FOR);
		write_t(s, s->minimum);
		write_t(s, w);
		pack_buffer(s, s->buf, n, w);
		s->data_ptr += bp;
		g = &s->segs[s->nseg - 1];
		g->count += n;
		return 1;
	}
	return 0;
}

static void st_update(bp_state *s, uint64_t value, int is_valid) { /* Update :296-317 */
	s->valid[s->idx] = (uint8_t)is_valid;
	s->all_valid = s->all_valid && is_valid;
	s->all_invalid = s->all_invalid && !is_valid;
	if (is_valid) {
		value = trunc_t(s->t, value);
		s->buf[s->idx] = value;
		if (lt_t(s->t, value, s->minimum)) s->minimum = value;
		if (lt_t(s->t, s->maximum, value)) s->maximum = value;
	} else if (s->null_zero) {
		s->buf[s->idx] = 0;
	}
	s->idx++;
	if (s->idx == BP_GROUP) {
		if (!st_flush(s)) s->failed = 1;
		st_reset(s);
	}
}

/* Compress n values (validity: one byte per row, NULL = all valid) exactly as BitpackingCompress +
 * BitpackingFinalizeCompress drive the state.  Returns an opaque handle, NULL if the codec cannot encode the
 * data (Flush returned false: BitpackingFinalAnalyze would have reported INVALID_INDEX). */
ORC_API bp_state *bp_compress(const void *vals, const uint8_t *validity, uint64_t n, unsigned type_size, int is_signed,
                              int force_mode, int null_zero) {
	bp_state *s = (bp_state *)calloc(1, sizeof(bp_state));
	s->t = mk_ty(type_size, is_signed);
	s->force_mode = force_mode;
	s->null_zero = null_zero;
	s->buf = s->buf_internal + 1;
	st_reset(s);
	create_segment(s, 0);
	const uint8_t *src = (const uint8_t *)vals;
	for (uint64_t i = 0; i < n; i++) {
		uint64_t v = 0;
		memcpy(&v, src + i * type_size, type_size);
		st_update(s, v, validity ? validity[i] != 0 : 1);
	}
	if (!st_flush(s)) s->failed = 1; /* Finalize */
	flush_segment(s);
	if (s->failed) {
		for (uint64_t i = 0; i < s->nseg; i++) free(s->segs[i].block);
		free(s->segs);
		free(s);
		return NULL;
	}
	return s;
}

ORC_API void bp_free(bp_state *s) {
	if (!s) return;
	for (uint64_t i = 0; i < s->nseg; i++) free(s->segs[i].block);
	free(s->segs);
	free(s);
}
ORC_API uint64_t bp_num_segments(const bp_state *s) { return s->nseg; }
ORC_API const uint8_t *bp_segment_block(const bp_state *s, uint64_t i) { return s->segs[i].block; }
ORC_API uint64_t bp_segment_count(const bp_state *s, uint64_t i) { return s->segs[i].count; }
ORC_API uint64_t bp_segment_start(const bp_state *s, uint64_t i) { return s->segs[i].start; }
ORC_API uint64_t bp_segment_size(const bp_state *s, uint64_t i) { return s->segs[i].total_size; }
ORC_API uint64_t bp_groups_by_mode(const bp_state *s, int mode) { return s->groups_by_mode[mode]; }

/* ------------------------------------------------------------------------------------------------
 * Scan: decode rows [start, start+n) of one segment block (LoadNextGroup + BitpackingScanPartial, group by
 * group; the 32-value algorithm groups and the running delta are restated directly).
 * ---------------------------------------------------------------------------------------------- */
ORC_API void bp_scan(const uint8_t *block, unsigned type_size, int is_signed, uint64_t seg_count, uint64_t start,
                     uint64_t n, void *out) {
	ty t = mk_ty(type_size, is_signed);
	uint64_t first;
	memcpy(&first, block, 8);
	uint8_t *dst = (uint8_t *)out;
	uint64_t g0 = start / BP_GROUP, g1 = (start + n + BP_GROUP - 1) / BP_GROUP;
	uint64_t tmp[BP_GROUP];
	for (uint64_t g = g0; g < g1 && n; g++) {
		uint32_t enc;
		memcpy(&enc, block + first - 4 * (g + 1), 4);
		int mode = (int)(enc >> 24);
		const uint8_t *p = block + (enc & 0xffffff);
		uint64_t rows = seg_count - g * BP_GROUP < BP_GROUP ? seg_count - g * BP_GROUP : BP_GROUP;
		if (mode == MODE_CONSTANT) {
			uint64_t c = load_t(p, t);
			for (uint64_t i = 0; i < rows; i++) tmp[i] = c;
		} else if (mode == MODE_CONSTANT_DELTA) {
			uint64_t fr = load_t(p, t), c = load_t(p + t.ts, t);
			for (uint64_t i = 0; i < rows; i++) tmp[i] = trunc_t(t, i * c + fr); /* :771-774 */
		} else {
			uint64_t fr = load_t(p, t);
			unsigned w = (unsigned)(load_t(p + t.ts, t) & 0xff);
			const uint8_t *q = p + 2 * t.ts;
			uint64_t prev = 0;
			if (mode == MODE_DELTA_FOR) {
				prev = load_t(q, t);
				q += t.ts;
			}
			for (uint64_t i = 0; i < rows; i += BP_ALGO) {
				uint64_t v[BP_ALGO];
				bp_unpack_group(q + i * w / 8, w, v);
				for (uint64_t k = 0; k < BP_ALGO && i + k < rows; k++) {
					if (mode == MODE_DELTA_FOR) {
						prev = trunc_t(t, prev + v[k] + fr); /* ApplyFrameOfReference + DeltaDecode :810-813 */
						tmp[i + k] = prev;
					} else {
						tmp[i + k] = trunc_t(t, v[k] + fr);
					}
				}
			}
		}
		uint64_t lo = g * BP_GROUP > start ? g * BP_GROUP : start;
		uint64_t hi = g * BP_GROUP + rows < start + n ? g * BP_GROUP + rows : start + n;
		for (uint64_t r = lo; r < hi; r++) memcpy(dst + (r - start) * t.ts, &tmp[r - g * BP_GROUP], t.ts);
	}
}

/* mode and header of one group, for the tests */
ORC_API int bp_group_info(const uint8_t *block, unsigned type_size, uint64_t g, uint32_t *offset, uint32_t *width) {
	ty t = mk_ty(type_size, 0);
	uint64_t first;
	memcpy(&first, block, 8);
	uint32_t enc;
	memcpy(&enc, block + first - 4 * (g + 1), 4);
	int mode = (int)(enc >> 24);
	if (offset) *offset = enc & 0xffffff;
	if (width) *width = (mode == MODE_FOR || mode == MODE_DELTA_FOR) ? (uint32_t)(load_t(block + (enc & 0xffffff) + t.ts, t) & 0xff) : 0;
	return mode;
}
