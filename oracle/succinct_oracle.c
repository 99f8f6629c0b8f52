/*
 * succinct_oracle.c — CPU restatement of the reference's succinct column-segment path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU baseline.  The product path (libadacodec.so) never links or calls it.
 *
 * Pinning status: the reference holds NO tests, golden vectors or fixtures for this path
 * (SURVEY.md §4), and its SDSL dependency (simongog/sdsl-lite, unpinned git HEAD; only the headers
 * are vendored, libsdsl.a and its sources are absent) cannot be linked here without writing
 * stand-ins for the missing library, so oracle/_ref is NOT built.  The oracle is pinned by the
 * known answers recorded from the real SDSL headers / the real reference binary in SURVEY.md §8c
 * (K1..K6, the 1513 B size) and BASELINE.md §2 (end-to-end GetTotalDataSize figures), committed as
 * tests/golden/survey_known_answers.json and checked by tests/test_oracle_golden.py.
 *
 * All file:line citations are relative to /root/reference.
 *
 * What is restated:
 *   A9  sdsl::bits::{hi,read_int,write_int}          third_party/sdsl/include/sdsl/bits.hpp:392-416,456-529
 *       memory_manager::resize / size_in_bytes        third_party/sdsl/include/sdsl/memory_management.hpp:344-375,
 *                                                     int_vector.hpp:602-609,1565-1578
 *   A0  int_vector<0> + ColumnSegment succinct state  int_vector.hpp:289-327, column_segment.hpp:60-64,190-214
 *   A1  SuccinctAppendLoop / SuccinctAppend           src/storage/compression/succinct.cpp:271-322
 *   A2  ColumnSegment::BitCompressFromSuccinct        src/storage/table/column_segment.cpp:348-383
 *   A3  SuccinctScanPartial                           src/storage/compression/succinct.cpp:123-144
 *   A4  ColumnSegment::BitCompressFromUncompressed    src/storage/table/column_segment.cpp:385-456
 *   A5  ColumnSegment::UncompressSuccinct             src/storage/table/column_segment.cpp:458-506
 *   A6  SuccinctFetchRow (intended semantics)         src/storage/compression/succinct.cpp:244-260
 *   A7  ColumnSegment::{Append,Compact,Uncompact,Scan} glue   column_segment.cpp:154-188,247-346
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * A9: bit primitives
 * ---------------------------------------------------------------------------------------------- */

static inline uint64_t low_mask(unsigned len) { /* bits::lo_set[len], len in 0..64 */
	return len >= 64 ? ~0ULL : ((1ULL << len) - 1ULL);
}

/* bits::hi — index of the most significant set bit, hi(0) == 0 (bits.hpp:392-397) */
ORC_API uint32_t orc_hi(uint64_t x) {
	return x == 0 ? 0u : (uint32_t)(63 - __builtin_clzll(x));
}

/* bits::read_int — `len` bits starting `off` bits into *word, little-endian u64 stream, LSB first;
 * may straddle into word[1] (bits.hpp:501-511). */
static inline uint64_t rd_int(const uint64_t *word, unsigned off, unsigned len) {
	uint64_t lo = word[0] >> off;
	if (off + len > 64) {
		unsigned spill = (off + len) & 63u;
		return lo | ((word[1] & low_mask(spill)) << (64 - off));
	}
	return lo & low_mask(len);
}

/* bits::write_int — read-modify-write of `len` bits at `off`; bits outside the field are kept
 * (bits.hpp:456-476). */
static inline void wr_int(uint64_t *word, uint64_t x, unsigned off, unsigned len) {
	x &= low_mask(len);
	if (off + len < 64) {
		word[0] &= (~0ULL << (off + len)) | low_mask(off);
		word[0] |= x << off;
	} else {
		word[0] &= low_mask(off);
		word[0] |= x << off;
		unsigned spill = (off + len) & 63u;
		if (spill) {
			word[1] &= ~low_mask(spill);
			word[1] |= x >> (len - spill);
		}
	}
}

/* ------------------------------------------------------------------------------------------------
 * A0: int_vector<0> model
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
	uint64_t bit_size; /* m_size  (int_vector.hpp:325) */
	uint64_t *data;    /* m_data  (int_vector.hpp:326) */
	uint8_t width;     /* m_width (int_vector.hpp:327) */
	uint64_t alloc_bytes;
} orc_ivec;

static void ivec_init(orc_ivec *v) {
	/* default ctor: width 64, size 0, one padding word allocated (memory_management.hpp:353) */
	v->bit_size = 0;
	v->width = 64;
	v->alloc_bytes = 8;
	v->data = (uint64_t *)calloc(1, 8);
}

static void ivec_free(orc_ivec *v) {
	free(v->data);
	v->data = NULL;
	v->bit_size = 0;
	v->alloc_bytes = 0;
}

/* int_vector_trait<0>::set_width: out-of-range widths become 64 */
static void ivec_set_width(orc_ivec *v, unsigned w) { v->width = (w > 0 && w <= 64) ? (uint8_t)w : 64; }

static uint64_t ivec_size(const orc_ivec *v) { return v->bit_size / v->width; }

/* memory_manager::resize (memory_management.hpp:344-375).  Allocation is ((bits+64)>>6)*8 bytes; on a
 * (re)allocation the bits between bit_size and the next word boundary are cleared, and if bit_size is a
 * multiple of 64 the padding word is cleared.  When the byte size does not change NOTHING is cleared.
 * Deviation (deterministic superset): bytes gained by growth are zero here; SDSL leaves them as realloc
 * returned them (never read by the path: pack/scan loops stop at `count`). */
static void ivec_bit_resize(orc_ivec *v, uint64_t bits) {
	uint64_t old_bytes = ((v->bit_size + 63) >> 6) << 3;
	uint64_t new_bytes = ((bits + 63) >> 6) << 3;
	int do_realloc = old_bytes != new_bytes;
	v->bit_size = bits;
	if (do_realloc || v->data == NULL) {
		uint64_t alloc = ((bits + 64) >> 6) << 3;
		uint64_t *nd = (uint64_t *)calloc(1, alloc);
		if (v->data) {
			memcpy(nd, v->data, alloc < v->alloc_bytes ? alloc : v->alloc_bytes);
			free(v->data);
		}
		v->data = nd;
		v->alloc_bytes = alloc;
		uint64_t cap = ((bits + 63) >> 6) << 6;
		if (bits < cap) {
			wr_int(v->data + (bits >> 6), 0, (unsigned)(bits & 63), (unsigned)(cap - bits));
		}
		if ((bits & 63) == 0) {
			v->data[bits >> 6] = 0;
		}
	}
}

static void ivec_resize(orc_ivec *v, uint64_t n) { ivec_bit_resize(v, n * v->width); }

/* int_vector::operator[] read / write through the proxy (int_vector.hpp:1359-1363,634-680) */
static inline uint64_t ivec_get(const orc_ivec *v, uint64_t i) {
	uint64_t bit = i * v->width;
	return rd_int(v->data + (bit >> 6), (unsigned)(bit & 63), v->width);
}
static inline void ivec_set(orc_ivec *v, uint64_t i, uint64_t x) {
	uint64_t bit = i * v->width;
	wr_int(v->data + (bit >> 6), x, (unsigned)(bit & 63), v->width);
}

/* sdsl::size_in_bytes = serialize(nullstream): 8 B bit-size + 1 B width + ceil(bits/64) words
 * (io.hpp:636-640, int_vector.hpp:602-609,1565-1578) */
ORC_API uint64_t orc_size_in_bytes(uint64_t bit_size) { return 9 + (((bit_size + 63) >> 6) << 3); }

/* int_vector<0>::serialize (int_vector.hpp:1565-1578): write_header = m_size as uint64 then m_width as uint8
 * (:602-609), write_data = capacity() >> 6 words, capacity() = ((m_size + 63) >> 6) << 6 (:1546-1561).
 * Returns the bytes written (== orc_size_in_bytes). */
ORC_API uint64_t orc_serialize(const uint64_t *words, uint64_t bit_size, unsigned width, uint8_t *out) {
	uint64_t nwords = (bit_size + 63) >> 6;
	uint8_t w8 = (uint8_t)width;
	memcpy(out, &bit_size, 8);
	memcpy(out + 8, &w8, 1);
	if (nwords) memcpy(out + 9, words, nwords * 8);
	return 9 + nwords * 8;
}
/* int_vector<0>::load (int_vector.hpp:1581-1595): read_header (:593-599), bit_resize(size), read capacity() >> 6
 * words.  Returns the bytes consumed, 0 when `len` is too short. */
ORC_API uint64_t orc_load(const uint8_t *in, uint64_t len, uint64_t *bit_size, unsigned *width, uint64_t *words,
                          uint64_t cap_words) {
	uint8_t w8;
	if (len < 9) return 0;
	memcpy(bit_size, in, 8);
	memcpy(&w8, in + 8, 1);
	*width = w8;
	uint64_t nwords = (*bit_size + 63) >> 6;
	if (len < 9 + nwords * 8 || nwords > cap_words) return 0;
	if (nwords) memcpy(words, in + 9, nwords * 8);
	return 9 + nwords * 8;
}

/* ------------------------------------------------------------------------------------------------
 * width rules
 * ---------------------------------------------------------------------------------------------- */

static unsigned pad_to_byte(unsigned w) { return (w + 7u) & ~7u; } /* column_segment.cpp:356-359 */

/* BitCompressFromSuccinct: w = hi(max_factor - min_factor) + 1 in wrapping u64 arithmetic
 * (column_segment.cpp:351-359) */
ORC_API uint32_t orc_width_from_succinct(uint64_t min_factor, uint64_t max_factor, int padded) {
	unsigned w = orc_hi(max_factor - min_factor) + 1;
	return padded ? pad_to_byte(w) : w;
}

/* BitCompressFromUncompressed: the local max is reduced by the local min only when max > min (and
 * max != 0, min != UINT64_MAX), so a constant non-zero segment keeps w = hi(value)+1
 * (column_segment.cpp:404-417) */
ORC_API uint32_t orc_width_from_uncompressed(uint64_t lmin, uint64_t lmax, int padded) {
	if (lmax != 0 && lmin != UINT64_MAX && lmax > lmin) {
		lmax -= lmin;
	}
	unsigned w = orc_hi(lmax) + 1;
	return padded ? pad_to_byte(w) : w;
}

/* ------------------------------------------------------------------------------------------------
 * A0/A7: ColumnSegment succinct state
 * ---------------------------------------------------------------------------------------------- */

enum { ORC_FN_UNCOMPRESSED = 1, ORC_FN_SUCCINCT = 10 }; /* compression_type.hpp:17-27 */

typedef struct {
	/* configuration (config.hpp:189-197) */
	int padded;   /* succinct_padded_to_next_byte_enabled */
	int adaptive; /* adaptive_succinct_compression_enabled (== background_compaction_enabled) */
	/* type */
	unsigned type_size; /* 1,2,4,8 */
	int is_signed;
	/* segment */
	uint64_t segment_size; /* bytes */
	uint64_t count;        /* SegmentBase::count */
	uint64_t num_elements; /* column_segment.hpp:190 */
	uint64_t min_factor, max_factor;
	int compacted;
	int succinct_possible;
	int function;            /* ORC_FN_* */
	int force_reinit_scan;   /* force_reinitializing_scan_state */
	orc_ivec vec;            /* succinct_vec */
	uint8_t *block;          /* raw block when function == UNCOMPRESSED */
	uint64_t num_reads;      /* catalog counter, kept on the segment for convenience */
} orc_segment;

static uint64_t load_ext(const void *p, unsigned type_size, int sign_extend) {
	switch (type_size) {
	case 1: return sign_extend ? (uint64_t)(int64_t)*(const int8_t *)p : (uint64_t)*(const uint8_t *)p;
	case 2: {
		uint16_t t;
		memcpy(&t, p, 2);
		return sign_extend ? (uint64_t)(int64_t)(int16_t)t : (uint64_t)t;
	}
	case 4: {
		uint32_t t;
		memcpy(&t, p, 4);
		return sign_extend ? (uint64_t)(int64_t)(int32_t)t : (uint64_t)t;
	}
	default: {
		uint64_t t;
		memcpy(&t, p, 8);
		return t;
	}
	}
}

/* NullValue<T>() = numeric_limits<T>::min() (null_value.hpp:26-28), as u64(T) */
static uint64_t null_value_u64(unsigned type_size, int is_signed) {
	if (!is_signed) return 0;
	return (uint64_t)(-(int64_t)(1ULL << (8 * type_size - 1)));
}

/* ColumnSegment::CreateTransientSegment + ctor (column_segment.cpp:45-82,87-110) */
ORC_API orc_segment *orc_seg_create(unsigned type_size, int is_signed, uint64_t segment_size, int succinct_enabled,
                                    int adaptive, int padded) {
	if (type_size != 1 && type_size != 2 && type_size != 4 && type_size != 8) return NULL;
	orc_segment *s = (orc_segment *)calloc(1, sizeof(orc_segment));
	s->padded = padded;
	s->adaptive = adaptive;
	s->type_size = type_size;
	s->is_signed = is_signed;
	s->segment_size = segment_size;
	s->min_factor = UINT64_MAX;
	s->max_factor = 0;
	ivec_init(&s->vec);
	if (succinct_enabled && !adaptive) {
		s->succinct_possible = 1;
		s->function = ORC_FN_SUCCINCT;
		ivec_set_width(&s->vec, type_size * 8);
		ivec_resize(&s->vec, segment_size / type_size);
	} else {
		s->succinct_possible = succinct_enabled;
		s->function = ORC_FN_UNCOMPRESSED;
		s->block = (uint8_t *)calloc(1, segment_size ? segment_size : 1);
	}
	return s;
}

ORC_API void orc_seg_destroy(orc_segment *s) {
	if (!s) return;
	ivec_free(&s->vec);
	free(s->block);
	free(s);
}

/* A2 — BitCompressFromSuccinct (column_segment.cpp:348-383): in-place repack old_w -> w with the frame of
 * reference subtracted (only when min_factor != UINT64_MAX), then bit_resize(count*w). */
static void bit_compress_from_succinct(orc_segment *s) {
	uint64_t mn = s->min_factor;
	unsigned w = orc_width_from_succinct(s->min_factor, s->max_factor, s->padded);
	w &= 0xff;
	unsigned old_w = s->vec.width;
	if (old_w > w) {
		uint64_t rbit = 0, wbit = 0;
		for (uint64_t i = 0; i < s->count; i++) {
			uint64_t x = rd_int(s->vec.data + (rbit >> 6), (unsigned)(rbit & 63), old_w);
			rbit += old_w;
			if (mn != UINT64_MAX) x -= mn;
			wr_int(s->vec.data + (wbit >> 6), x, (unsigned)(wbit & 63), w);
			wbit += w;
		}
		ivec_bit_resize(&s->vec, s->count * w);
		ivec_set_width(&s->vec, w);
	}
	s->compacted = 1;
}

/* A4 — BitCompressFromUncompressed (column_segment.cpp:385-456): local min/max over the raw block (values
 * ZERO-extended here; the reference memcpy's type_size bytes into an uninitialised u64, which is UB for
 * type_size < 8 — zero extension is the defined reading), pack x - min, switch to SUCCINCT.
 * `store_min`: the reference never records the local min (defect 2); the product does. */
static void bit_compress_from_uncompressed(orc_segment *s, int store_min) {
	uint64_t mn = UINT64_MAX, mx = 0;
	for (uint64_t i = 0; i < s->count; i++) {
		uint64_t c = load_ext(s->block + i * s->type_size, s->type_size, 0);
		if (c < mn) mn = c;
		if (c > mx) mx = c;
	}
	unsigned w = orc_width_from_uncompressed(mn, mx, s->padded) & 0xff;
	unsigned old_w = s->vec.width;
	if (old_w > w) {
		uint64_t wbit = 0;
		for (uint64_t i = 0; i < s->count; i++) {
			uint64_t x = load_ext(s->block + i * s->type_size, s->type_size, 0);
			if (mn != UINT64_MAX) x -= mn;
			wr_int(s->vec.data + (wbit >> 6), x, (unsigned)(wbit & 63), w);
			wbit += w;
		}
		ivec_bit_resize(&s->vec, s->count * w);
		ivec_set_width(&s->vec, w);
		if (store_min) s->min_factor = mn;
	} else {
		/* old_w <= w: the reference leaves succinct_vec untouched (all zero slots) although the data lives in
		 * the block — nothing it later reads back is meaningful.  Mirror: copy raw values so the segment
		 * stays decodable; flagged as outside the parity domain. */
		for (uint64_t i = 0; i < s->count; i++) {
			ivec_set(&s->vec, i, load_ext(s->block + i * s->type_size, s->type_size, 0));
		}
		if (store_min) s->min_factor = UINT64_MAX;
	}
	s->function = ORC_FN_SUCCINCT;
	s->compacted = 1;
}

/* ColumnSegment::Compact (column_segment.cpp:273-322) */
ORC_API void orc_seg_compact(orc_segment *s, int store_min) {
	if (s->compacted || s->num_elements == 0 || !s->succinct_possible) return;
	if (s->function == ORC_FN_SUCCINCT) {
		bit_compress_from_succinct(s);
		return;
	}
	ivec_set_width(&s->vec, s->type_size * 8);
	ivec_resize(&s->vec, s->segment_size / s->type_size);
	bit_compress_from_uncompressed(s, store_min);
}

/* A5 — UncompressSuccinct via ColumnSegment::Uncompact (column_segment.cpp:324-346,458-506):
 * block[i] = T(v[i] + min_factor) for i < v.size(), switch to UNCOMPRESSED, v.resize(0).
 * NOTE the reference adds min_factor unconditionally here (no UINT64_MAX test). `correct` = 1 applies the
 * product's rule instead (add only when a frame of reference was subtracted). */
ORC_API void orc_seg_uncompact(orc_segment *s, int correct) {
	if (!s->compacted || s->function != ORC_FN_SUCCINCT) return;
	free(s->block);
	s->block = (uint8_t *)calloc(1, s->segment_size ? s->segment_size : 1);
	uint64_t n = ivec_size(&s->vec);
	int packed = s->vec.width < s->type_size * 8;
	for (uint64_t i = 0; i < n && (i + 1) * s->type_size <= s->segment_size; i++) {
		uint64_t c = ivec_get(&s->vec, i);
		if (correct) {
			if (packed && s->min_factor != UINT64_MAX) c += s->min_factor;
			if (packed && s->min_factor == UINT64_MAX && s->max_factor == UINT64_MAX) { /* defect 7: adac_stored_min */
				c += UINT64_MAX - ((1ull << s->vec.width) - 1ull);
			}
		} else {
			c += s->min_factor;
		}
		memcpy(s->block + i * s->type_size, &c, s->type_size);
	}
	s->function = ORC_FN_UNCOMPRESSED;
	s->compacted = 0;
	ivec_resize(&s->vec, 0);
	s->force_reinit_scan = 1;
}

/* A1 — SuccinctAppend / SuccinctAppendLoop (succinct.cpp:271-322) for SUCCINCT segments, FixedSizeAppend
 * (fixed_size_uncompressed.cpp) for UNCOMPRESSED ones; A7 — ColumnSegment::Append glue
 * (column_segment.cpp:247-271).  `validity` is a DuckDB validity mask (bit i of word i/64 set = valid) or
 * NULL for all-valid; `sel` an optional selection vector.  Returns rows consumed. */
ORC_API uint64_t orc_seg_append(orc_segment *s, const void *vals, const uint64_t *validity, const uint32_t *sel,
                                uint64_t offset, uint64_t count, int store_min) {
	int uncompacted = 0;
	if (s->compacted) {
		orc_seg_uncompact(s, store_min /* product rule when the product stores min */);
		uncompacted = 1;
	}
	uint64_t max_tuples = s->segment_size / s->type_size;
	uint64_t copy = count < max_tuples - s->count ? count : max_tuples - s->count;
	const uint8_t *src = (const uint8_t *)vals;
	if (s->function == ORC_FN_SUCCINCT) {
		uint64_t mn = UINT64_MAX, mx = 0;
		for (uint64_t i = 0; i < copy; i++) {
			uint64_t sidx = sel ? sel[offset + i] : offset + i;
			int valid = validity ? (int)((validity[sidx >> 6] >> (sidx & 63)) & 1) : 1;
			if (valid) {
				uint64_t x = load_ext(src + sidx * s->type_size, s->type_size, s->is_signed);
				ivec_set(&s->vec, s->count + i, x);
				if (x < mn) mn = x;
				if (x > mx) mx = x;
			} else {
				ivec_set(&s->vec, s->count + i, null_value_u64(s->type_size, s->is_signed));
			}
		}
		if (mn < s->min_factor) s->min_factor = mn; /* UpdateMinFactor */
		if (mx > s->max_factor) s->max_factor = mx; /* UpdateMaxFactor */
	} else {
		for (uint64_t i = 0; i < copy; i++) {
			uint64_t sidx = sel ? sel[offset + i] : offset + i;
			int valid = validity ? (int)((validity[sidx >> 6] >> (sidx & 63)) & 1) : 1;
			uint64_t x = valid ? load_ext(src + sidx * s->type_size, s->type_size, 0)
			                   : null_value_u64(s->type_size, s->is_signed);
			memcpy(s->block + (s->count + i) * s->type_size, &x, s->type_size);
		}
	}
	s->count += copy;
	s->num_elements += count; /* sic: the requested count (column_segment.cpp:263) */
	uint64_t slots = ivec_size(&s->vec);
	if (!s->compacted && !s->adaptive && (s->num_elements >= slots || uncompacted)) {
		orc_seg_compact(s, store_min);
	}
	return copy;
}

/* A3 — SuccinctScanPartial (succinct.cpp:123-144).  mode 0 = product semantics (min added only when a frame
 * of reference was actually subtracted: the parity domain of SURVEY.md §8a (iii)); mode 1 = the reference
 * bit for bit (min_factor added whenever != UINT64_MAX: defect 1 on unpacked segments); `with_copy` also
 * performs the reference's per-call deep copy of the whole int_vector (succinct.cpp:127) for the faithful
 * CPU baseline timing. */
ORC_API void orc_seg_scan_partial(const orc_segment *s, uint64_t start, uint64_t n, void *out, int mode,
                                  int with_copy) {
	const orc_ivec *src = &s->vec;
	orc_ivec tmp;
	if (with_copy) {
		tmp = s->vec;
		uint64_t bytes = ((s->vec.bit_size + 64) >> 6) << 3;
		tmp.data = (uint64_t *)calloc(1, bytes);
		memcpy(tmp.data, s->vec.data, bytes < s->vec.alloc_bytes ? bytes : s->vec.alloc_bytes);
		src = &tmp;
	}
	uint8_t *dst = (uint8_t *)out;
	unsigned ts = s->type_size;
	int add;
	if (mode == 1) {
		add = s->min_factor != UINT64_MAX;
	} else {
		add = s->min_factor != UINT64_MAX && src->width < ts * 8;
	}
	uint64_t mn = s->min_factor;
	/* product semantics, defect 7 (sentinel collision): a packed segment with min == max == UINT64_MAX holds
	 * only all-ones VALID values, stored as their unsubtracted low bits; the product decodes field + (UINT64_MAX -
	 * (2^w - 1)) (adac_stored_min), the reference returns the field */
	int all_ones = mode == 0 && src->width < ts * 8 && s->min_factor == UINT64_MAX && s->max_factor == UINT64_MAX;
	for (uint64_t i = 0; i < n; i++) {
		uint64_t e = ivec_get(src, start + i);
		if (add) e += mn;
		if (all_ones) e += UINT64_MAX - ((src->width >= 64 ? 0ull : (1ull << src->width)) - 1ull); /* adac_stored_min */
		memcpy(dst + i * ts, &e, ts);
	}
	if (with_copy) free(tmp.data);
}

/* ColumnSegment::Scan/ScanPartial glue (column_segment.cpp:154-188): count the read, lazily compact, then
 * scan through the current function. */
ORC_API void orc_seg_scan(orc_segment *s, uint64_t start, uint64_t n, void *out, int mode, int store_min) {
	s->num_reads++;
	if (!s->compacted && !s->adaptive) orc_seg_compact(s, store_min);
	s->force_reinit_scan = 0;
	if (s->function == ORC_FN_SUCCINCT) {
		orc_seg_scan_partial(s, start, n, out, mode, 0);
	} else {
		memcpy(out, s->block + start * s->type_size, n * s->type_size);
	}
}

/* A6 — SuccinctFetchRow, intended semantics: out = T(v[row] + min) (succinct.cpp:244-260 is defective:
 * it ignores row_id and writes size() values — SURVEY.md §4-3; not reproduced). */
ORC_API void orc_seg_fetch_row(const orc_segment *s, uint64_t row, void *out) {
	if (s->function == ORC_FN_SUCCINCT) {
		orc_seg_scan_partial(s, row, 1, out, 0, 0);
	} else {
		memcpy(out, s->block + row * s->type_size, s->type_size);
	}
}

/* ColumnSegment::GetDataSize (column_segment.cpp:204-214), is_data_segment == true */
ORC_API uint64_t orc_seg_data_size(const orc_segment *s) {
	if (s->function == ORC_FN_SUCCINCT) return orc_size_in_bytes(s->vec.bit_size);
	return s->segment_size;
}

/* getters for the tests */
ORC_API uint64_t orc_seg_count(const orc_segment *s) { return s->count; }
ORC_API uint64_t orc_seg_min(const orc_segment *s) { return s->min_factor; }
ORC_API uint64_t orc_seg_max(const orc_segment *s) { return s->max_factor; }
ORC_API uint32_t orc_seg_width(const orc_segment *s) { return s->vec.width; }
ORC_API uint64_t orc_seg_bit_size(const orc_segment *s) { return s->vec.bit_size; }
ORC_API int orc_seg_compacted(const orc_segment *s) { return s->compacted; }
ORC_API int orc_seg_function(const orc_segment *s) { return s->function; }
ORC_API uint64_t orc_seg_num_reads(const orc_segment *s) { return s->num_reads; }
ORC_API void orc_seg_reset_reads(orc_segment *s) { s->num_reads = 0; }
ORC_API const uint64_t *orc_seg_words(const orc_segment *s) { return s->vec.data; }
ORC_API uint64_t orc_seg_num_words(const orc_segment *s) { return (s->vec.bit_size + 63) >> 6; }
ORC_API const uint8_t *orc_seg_block(const orc_segment *s) { return s->block; }

/* ------------------------------------------------------------------------------------------------
 * Flat (segment-less) forms of A2/A4/A3 used for bulk parity checks: same arithmetic, caller-owned buffers.
 * ---------------------------------------------------------------------------------------------- */

/* min/max exactly as the two reference paths see them.
 *   rule 0 (A1, succinct.cpp:286-287): u64(T) sign-extended for signed T, NULL rows skipped;
 *   rule 1 (A4, column_segment.cpp:390-400): zero-extended, NULL slots (NullValue<T>) included. */
ORC_API void orc_analyze_flat(const void *vals, uint64_t n, unsigned type_size, int is_signed, int rule,
                              const uint64_t *validity, uint64_t vbit0, uint64_t *mn_out, uint64_t *mx_out) {
	uint64_t mn = UINT64_MAX, mx = 0;
	const uint8_t *src = (const uint8_t *)vals;
	for (uint64_t i = 0; i < n; i++) {
		int valid = validity ? (int)((validity[(vbit0 + i) >> 6] >> ((vbit0 + i) & 63)) & 1) : 1;
		uint64_t x;
		if (rule == 0) {
			if (!valid) continue;
			x = load_ext(src + i * type_size, type_size, is_signed);
		} else {
			x = valid ? load_ext(src + i * type_size, type_size, 0)
			          : (null_value_u64(type_size, is_signed) & low_mask(8 * type_size));
		}
		if (x < mn) mn = x;
		if (x > mx) mx = x;
	}
	*mn_out = mn;
	*mx_out = mx;
}

/* pack n values at width w into a zeroed word buffer: word stream identical to what A2/A4 leave behind
 * (value i at bits [i*w,(i+1)*w), (x - min) mod 2^w, tail bits zero). min == UINT64_MAX means "do not
 * subtract". NULL slots carry NullValue<T>. */
ORC_API void orc_pack_flat(const void *vals, uint64_t n, unsigned type_size, int is_signed, const uint64_t *validity,
                           uint64_t vbit0, uint64_t mn, unsigned w, uint64_t *words) {
	const uint8_t *src = (const uint8_t *)vals;
	uint64_t nwords = (n * w + 63) >> 6;
	memset(words, 0, nwords * 8);
	uint64_t wbit = 0;
	uint64_t slot_mask = low_mask(8 * type_size);
	for (uint64_t i = 0; i < n; i++) {
		int valid = validity ? (int)((validity[(vbit0 + i) >> 6] >> ((vbit0 + i) & 63)) & 1) : 1;
		/* the slot holds the value truncated to 8*type_size bits (write_int masks), re-read zero-extended */
		uint64_t x = valid ? (load_ext(src + i * type_size, type_size, 0))
		                   : (null_value_u64(type_size, is_signed) & slot_mask);
		if (mn != UINT64_MAX) x -= mn;
		uint64_t xm = x & low_mask(w);
		unsigned off = (unsigned)(wbit & 63);
		words[wbit >> 6] |= xm << off;
		if (off + w > 64) words[(wbit >> 6) + 1] |= xm >> (64 - off);
		wbit += w;
	}
}

/* decode n values starting at `start`: out[i] = T(read_int(words,(start+i)*w,w) + add) */
ORC_API void orc_unpack_flat(const uint64_t *words, uint64_t start, uint64_t n, unsigned w, uint64_t add,
                             unsigned type_size, void *out) {
	uint8_t *dst = (uint8_t *)out;
	uint64_t bit = start * w;
	for (uint64_t i = 0; i < n; i++) {
		uint64_t e = rd_int(words + (bit >> 6), (unsigned)(bit & 63), w) + add;
		memcpy(dst + i * type_size, &e, type_size);
		bit += w;
	}
}

/* ------------------------------------------------------------------------------------------------
 * CPU baseline: full scan of a list of packed segments the way ColumnData::ScanVector drives
 * SuccinctScanPartial — 2048 values per call (column_data.cpp:92-139, vector_size.hpp:17) — on `threads`
 * host threads, one contiguous segment range per thread (DuckDB's one-row-group-per-task morsels,
 * row_group_collection.cpp:119-155).
 * ---------------------------------------------------------------------------------------------- */

typedef struct {
	const uint64_t *const *words; /* per-segment packed words */
	const uint64_t *counts;
	const uint8_t *widths;
	const uint64_t *adds;
	const uint64_t *out_offs; /* element offset of each segment in out */
	uint64_t seg_begin, seg_end;
	unsigned type_size;
	int with_copy;
	void *out;
} scan_job;

static void *scan_worker(void *arg) {
	scan_job *j = (scan_job *)arg;
	for (uint64_t s = j->seg_begin; s < j->seg_end; s++) {
		uint64_t n = j->counts[s];
		unsigned w = j->widths[s];
		uint8_t *dst = (uint8_t *)j->out + j->out_offs[s] * j->type_size;
		uint64_t nbytes = ((n * w + 64) >> 6) << 3;
		for (uint64_t start = 0; start < n; start += 2048) {
			uint64_t c = n - start < 2048 ? n - start : 2048;
			const uint64_t *src = j->words[s];
			uint64_t *tmp = NULL;
			if (j->with_copy) { /* succinct.cpp:127: `auto source = segment.succinct_vec;` */
				tmp = (uint64_t *)calloc(1, nbytes);
				memcpy(tmp, src, ((n * w + 63) >> 6) << 3);
				src = tmp;
			}
			orc_unpack_flat(src, start, c, w, j->adds[s], j->type_size, dst + start * j->type_size);
			free(tmp);
		}
	}
	return NULL;
}

ORC_API void orc_scan_segments_mt(const uint64_t *const *words, const uint64_t *counts, const uint8_t *widths,
                                  const uint64_t *adds, const uint64_t *out_offs, uint64_t nseg, unsigned type_size,
                                  int with_copy, void *out, int threads) {
	if (threads < 1) threads = 1;
	if ((uint64_t)threads > nseg && nseg > 0) threads = (int)nseg;
	pthread_t *tid = (pthread_t *)calloc(threads, sizeof(pthread_t));
	scan_job *jobs = (scan_job *)calloc(threads, sizeof(scan_job));
	for (int t = 0; t < threads; t++) {
		jobs[t] = (scan_job){words, counts, widths, adds, out_offs, nseg * t / threads, nseg * (t + 1) / threads,
		                     type_size, with_copy, out};
		if (threads == 1) {
			scan_worker(&jobs[t]);
		} else {
			pthread_create(&tid[t], NULL, scan_worker, &jobs[t]);
		}
	}
	if (threads > 1) {
		for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
	}
	free(tid);
	free(jobs);
}
