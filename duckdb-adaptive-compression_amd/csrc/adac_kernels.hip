// adac_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the succinct column-segment codec.
//
// Pure integer, HBM-bound work: no MFMA.  Common shape of every streaming kernel:
//   * one 256-thread workgroup (4 wave64) per TILE = 16 KiB of decoded values of ONE segment, tiles
//     enumerated over all segments of a batch by a device tile table, so a launch has >> 256 workgroups
//     and ragged / tiny segments cost nothing special;
//   * global traffic only as 16-byte-per-lane, 16-byte-aligned accesses (1 KiB per wave instruction):
//     packed words are staged through LDS with dwordx4 loads, decoded values leave as dwordx4 stores;
//   * no inter-tile reuse, hence no XCD-aware block remap: every byte is touched once.
//
// Reference loops these kernels replace (paths relative to the reference checkout):
//   k_unpack        SuccinctScanPartial                 src/storage/compression/succinct.cpp:123-144
//                   ColumnSegment::UncompressSuccinct   src/storage/table/column_segment.cpp:458-506
//   k_analyze       SuccinctAppendLoop min/max          src/storage/compression/succinct.cpp:271-306
//                   BitCompressFromUncompressed pass 1  src/storage/table/column_segment.cpp:390-400
//   k_plan          width decision                      src/storage/table/column_segment.cpp:351-363,404-420
//   k_pack          BitCompressFromSuccinct / ...FromUncompressed pack loops   column_segment.cpp:365-376,426-443
//   k_fetch         SuccinctFetchRow (intended)         src/storage/compression/succinct.cpp:244-260
//   bit layout      sdsl::bits::read_int / write_int    third_party/sdsl/include/sdsl/bits.hpp:456-529
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "adac_internal.h"

namespace adac {

namespace {

template <int BYTES> struct uint_of;
template <> struct uint_of<1> { using type = uint8_t; };
template <> struct uint_of<2> { using type = uint16_t; };
template <> struct uint_of<4> { using type = uint32_t; };
template <> struct uint_of<8> { using type = uint64_t; };

__device__ __forceinline__ uint32_t mask32(uint32_t bits) { // bits in 0..32
	return bits >= 32 ? 0xffffffffu : ((1u << bits) - 1u);
}
__device__ __forceinline__ uint64_t mask64(uint32_t bits) { // bits in 0..64
	return bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
}

// sdsl::bits::hi (bits.hpp:392-397)
__device__ __forceinline__ uint32_t hi_bit(uint64_t x) { return x == 0 ? 0u : 63u - (uint32_t)__clzll((long long)x); }

__device__ __forceinline__ uint64_t wave_min(uint64_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		uint64_t o = __shfl_xor(v, off, 64);
		v = o < v ? o : v;
	}
	return v;
}
__device__ __forceinline__ uint64_t wave_max(uint64_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		uint64_t o = __shfl_xor(v, off, 64);
		v = o > v ? o : v;
	}
	return v;
}
__device__ __forceinline__ uint64_t wave_sum(uint64_t v) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		v += __shfl_xor(v, off, 64);
	}
	return v;
}

// A descriptor at a wave-uniform address through the scalar unit: read as eight dwords (s_load_dwordx8) and taken
// apart with scalar shifts.  Read field by field, the byte-sized members (width, flags) come through VECTOR loads —
// gfx9 has no sub-dword scalar load — each behind its own s_waitcnt.
// Used ONLY where the width decides a branch before anything else can be issued (the fused scans' work items: the
// narrow-width test made the vector load a second round trip at every workgroup start, 3 - 5 % of a scan; the grouped
// scan's eligibility test).  NOT used by the tile kernels (k_unpack, k_analyze, k_pack, k_gather ...): there the
// field-by-field form — three scalar loads and the 2-byte vector load in flight together — is the faster one; with
// this helper in resolve_tile the decode of u64 at w = 8 ran 568 instead of 660 GB/s of packed bytes and every other
// decode 3 - 10 % slower, same box, same ISA after the first twenty instructions (profiles/r03_scan_split_ab.json,
// "tile_descriptor_loads").
__device__ __forceinline__ adac_segment_desc load_desc(const adac_segment_desc *__restrict__ p) {
	static_assert(sizeof(adac_segment_desc) == 32, "record layout");
	const uint32_t *__restrict__ q = reinterpret_cast<const uint32_t *>(p);
	uint32_t w[8];
#pragma unroll
	for (int i = 0; i < 8; i++) w[i] = q[i];
	adac_segment_desc d;
	d.word_off = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
	d.val_off = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
	d.min = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
	d.count = w[6];
	d.width = (uint8_t)(w[7] & 0xffu);
	d.flags = (uint8_t)((w[7] >> 8) & 0xffu);
	d.reserved = (uint16_t)(w[7] >> 16);
	return d;
}

// A tile resolved to its segment.
struct TileCtx {
	adac_segment_desc d;
	uint32_t seg;
	uint32_t first;  // first row of the tile inside the segment
	uint32_t n;      // rows in the tile
	uint64_t elem0;  // element index of row `first` in the value buffer
};

template <int TILE>
__device__ __forceinline__ TileCtx resolve_tile(const adac_segment_desc *__restrict__ descs,
                                                const TileRef *__restrict__ tiles) {
	TileCtx t;
	const TileRef r = tiles[blockIdx.x];
	t.seg = r.seg;
	t.first = r.first;
	t.d = descs[r.seg]; // field by field on purpose: see load_desc
	const uint32_t left = t.d.count - r.first;
	t.n = left < (uint32_t)TILE ? left : (uint32_t)TILE;
	t.elem0 = t.d.val_off + r.first;
	return t;
}

__device__ __forceinline__ uint64_t effective_add(const adac_segment_desc &d);

// An expanded tile record at a wave-uniform address through the scalar unit: eight dwords, taken apart with scalar
// shifts (gfx9 has no sub-dword scalar load: read field by field the 2- and 1-byte members would come through vector
// loads).
__device__ __forceinline__ TileRec load_tile_rec(const TileRec *__restrict__ p) {
	const uint32_t *__restrict__ q = reinterpret_cast<const uint32_t *>(p);
	uint32_t w[8];
#pragma unroll
	for (int i = 0; i < 8; i++) w[i] = q[i];
	TileRec r;
	r.word_off = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
	r.elem0 = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
	r.add = (uint64_t)w[4] | ((uint64_t)w[5] << 32);
	r.first = w[6];
	r.n = (uint16_t)(w[7] & 0xffffu);
	r.width = (uint8_t)((w[7] >> 16) & 0xffu);
	r.flags = (uint8_t)(w[7] >> 24);
	return r;
}

// one thread per tile: the tile entry and its segment's current descriptor -> the tile's record
__global__ void k_expand_tiles(const adac_segment_desc *__restrict__ descs, const TileRef *__restrict__ tiles,
                               uint64_t ntiles, uint32_t tile_rows, TileRec *__restrict__ recs) {
	const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= ntiles) return;
	const TileRef r = tiles[t];
	const adac_segment_desc d = descs[r.seg];
	const uint32_t left = d.count - r.first;
	TileRec o;
	o.word_off = d.word_off;
	o.elem0 = d.val_off + r.first;
	o.add = effective_add(d);
	o.first = r.first;
	o.n = (uint16_t)(left < tile_rows ? left : tile_rows);
	o.width = d.width;
	o.flags = d.flags;
	recs[t] = o;
}

// ---------------------------------------------------------------------------------------------
// Stage the packed bits of rows [first, first+n) of a segment into LDS with 16-byte loads.
// Returns the bit offset (0..127) of row `first` inside the staged image.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t stage_packed(const uint64_t *__restrict__ seg_words, uint32_t first, uint32_t n,
                                                 uint32_t w, uint4 *lds) {
	const uint64_t bitpos = (uint64_t)first * w;
	const uint64_t chunk0 = bitpos >> 7; // 16-byte chunk holding the first bit
	const uint32_t bit0 = (uint32_t)(bitpos & 127);
	const uint32_t nchunks = (bit0 + n * w + 127u) >> 7;
	const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(seg_words) + chunk0;
	for (uint32_t c = threadIdx.x; c < nchunks; c += kWorkgroup) {
		lds[c] = src[c];
	}
	return bit0;
}

// Read the w-bit field at `bit` of the staged image (sdsl::bits::read_int, bits.hpp:501-511) as lo/hi dwords.
template <bool WIDE>
__device__ __forceinline__ void read_field(const uint32_t *lds32, uint32_t bit, uint32_t mlo, uint32_t mhi,
                                           uint32_t &lo, uint32_t &hi) {
	const uint32_t dw = bit >> 5;
	const uint32_t sh = bit & 31u;
	const uint32_t a0 = lds32[dw];
	const uint32_t a1 = lds32[dw + 1];
	lo = __builtin_amdgcn_alignbit(a1, a0, sh);
	if (WIDE) {
		const uint32_t a2 = lds32[dw + 2];
		hi = __builtin_amdgcn_alignbit(a2, a1, sh) & mhi;
	} else {
		lo &= mlo;
		hi = 0;
	}
}

// Walk the rows of a staged tile in chunks of K = 16/sizeof(U) consecutive rows, chunk boundaries aligned to
// 16 bytes of the OUTPUT element index (elem0 + row), so a sink can use one dwordx4 store per full chunk.
// sink(base, vals, full): `base` = tile-local row of vals[0] (may be < 0 or run past n on partial chunks).
// WAVE: the trip count is the same for the 64 lanes of a wave (a sink with cross-lane operations: k_gather); lanes whose
// chunk lies past the tile call the sink with base >= n and full = false.
template <typename U, bool WIDE, bool WAVE = false, typename Sink>
__device__ __forceinline__ void decode_rows(const uint32_t *lds32, uint32_t bit0, uint32_t w, uint64_t add, uint32_t n,
                                            uint32_t align, Sink &&sink) {
	constexpr int K = 16 / (int)sizeof(U);
	const uint32_t mlo = WIDE ? 0xffffffffu : mask32(w);
	const uint32_t mhi = WIDE ? mask32(w - 32u) : 0u;
	const uint32_t add_lo = (uint32_t)add;
	for (uint32_t c = threadIdx.x; (WAVE ? (c & ~63u) : c) * K < n + align; c += kWorkgroup) {
		const int32_t base = (int32_t)(c * K) - (int32_t)align;
		U vals[K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			const uint32_t row = (uint32_t)(base + j);
			const uint32_t r = row < n ? row : 0u;
			uint32_t lo, hi;
			read_field<WIDE>(lds32, bit0 + r * w, mlo, mhi, lo, hi);
			if (sizeof(U) == 8) {
				vals[j] = (U)((((uint64_t)hi << 32) | lo) + add);
			} else {
				vals[j] = (U)(lo + add_lo);
			}
		}
		const bool full = base >= 0 && (uint32_t)(base + K) <= n;
		sink(base, vals, full);
	}
}

// Fast path for a FULL tile whose first row sits on a 16-byte boundary of the output (the common case: every
// tile of a segment but its last, for 16-byte-aligned segment placement).  Static trip count, no row
// predication, bit positions by one 24-bit multiply + adds (v_mul_lo_u32 is quarter rate), all LDS reads of a
// round issued back to back.  sink(base, vals) always gets a complete chunk.
template <typename U, bool WIDE, typename Sink>
__device__ __forceinline__ void decode_full_tile(const uint32_t *lds32, uint32_t bit0, uint32_t w, uint64_t add,
                                                 Sink &&sink) {
	constexpr int K = 16 / (int)sizeof(U);
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr int ROUNDS = TILE / (kWorkgroup * K);
	const uint32_t mlo = WIDE ? 0xffffffffu : mask32(w);
	const uint32_t mhi = WIDE ? mask32(w - 32u) : 0u;
	const uint32_t add_lo = (uint32_t)add;
	uint32_t bit = bit0 + __umul24(threadIdx.x * K, w); // < 2^24: at most 16384 rows x 64 bits
	const uint32_t step = kWorkgroup * K * w;
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		U vals[K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			uint32_t lo, hi;
			read_field<WIDE>(lds32, bit + j * w, mlo, mhi, lo, hi);
			if (sizeof(U) == 8) {
				vals[j] = (U)((((uint64_t)hi << 32) | lo) + add);
			} else {
				vals[j] = (U)(lo + add_lo);
			}
		}
		sink((int32_t)(r * kWorkgroup * K + threadIdx.x * K), vals, true);
		bit += step;
	}
}

// 1-byte types, full tile, width W as a template parameter: a lane's sixteen consecutive rows are at most 128 bits of
// the image, so FIVE dword reads (instead of two per row = 32) bring them into registers, four funnel shifts move the
// first field to bit 0, and every field then sits at a compile-time position (one v_bfe each).  The generic form's
// per-row LDS reads made the 1-byte decode the slowest of the sweep (4.4 - 5.0 TB/s against 5.7 - 6.3 for the wider
// types): 16 Ki rows per 16 KiB tile.
template <int W, typename U, typename Sink>
__device__ __forceinline__ void decode_full_tile_u8(const uint32_t *lds32, uint32_t bit0, uint64_t add, Sink &&sink) {
	static_assert(sizeof(U) == 1 && W >= 1 && W <= 8, "1-byte types only");
	constexpr int K = 16;
	constexpr int TILE = kTileBytes;
	constexpr int ROUNDS = TILE / (kWorkgroup * K);
	constexpr uint32_t mask = (1u << W) - 1u;
	const uint32_t add_lo = (uint32_t)add;
	uint32_t bit = bit0 + threadIdx.x * (uint32_t)(K * W);
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const uint32_t dw = bit >> 5, sh = bit & 31u;
		const uint32_t a0 = lds32[dw], a1 = lds32[dw + 1], a2 = lds32[dw + 2], a3 = lds32[dw + 3], a4 = lds32[dw + 4];
		uint32_t nrm[4];
		nrm[0] = __builtin_amdgcn_alignbit(a1, a0, sh);
		nrm[1] = __builtin_amdgcn_alignbit(a2, a1, sh);
		nrm[2] = __builtin_amdgcn_alignbit(a3, a2, sh);
		nrm[3] = __builtin_amdgcn_alignbit(a4, a3, sh);
		U vals[K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			const int pos = j * W, d = pos >> 5, s = pos & 31;
			const uint32_t f = s + W <= 32 ? ((nrm[d] >> s) & mask) : (__builtin_amdgcn_alignbit(nrm[d + 1], nrm[d], s) & mask);
			vals[j] = (U)(f + add_lo);
		}
		sink((int32_t)(r * kWorkgroup * K + threadIdx.x * K), vals, true);
		bit += (uint32_t)(kWorkgroup * K * W);
	}
}

// One staged tile -> sink, choosing the two/three-dword window and the full-tile fast path (both wave-uniform).
template <typename U, bool WAVE = false, typename Sink>
__device__ __forceinline__ void decode_tile(const uint32_t *lds32, uint32_t bit0, uint32_t w, uint64_t add, uint32_t n,
                                            uint32_t align, Sink &&sink) {
	constexpr uint32_t TILE = kTileBytes / sizeof(U);
	const bool fast = n == TILE && align == 0;
	if constexpr (sizeof(U) == 1) {
		if (fast) {
			switch (w) {
			case 1: decode_full_tile_u8<1, U>(lds32, bit0, add, sink); return;
			case 2: decode_full_tile_u8<2, U>(lds32, bit0, add, sink); return;
			case 3: decode_full_tile_u8<3, U>(lds32, bit0, add, sink); return;
			case 4: decode_full_tile_u8<4, U>(lds32, bit0, add, sink); return;
			case 5: decode_full_tile_u8<5, U>(lds32, bit0, add, sink); return;
			case 6: decode_full_tile_u8<6, U>(lds32, bit0, add, sink); return;
			case 7: decode_full_tile_u8<7, U>(lds32, bit0, add, sink); return;
			default: decode_full_tile_u8<8, U>(lds32, bit0, add, sink); return;
			}
		}
	}
	if (sizeof(U) == 8 && w > 32) {
		if (fast) {
			decode_full_tile<U, true>(lds32, bit0, w, add, sink);
		} else {
			decode_rows<U, true, WAVE>(lds32, bit0, w, add, n, align, sink);
		}
	} else {
		if (fast) {
			decode_full_tile<U, false>(lds32, bit0, w, add, sink);
		} else {
			decode_rows<U, false, WAVE>(lds32, bit0, w, add, n, align, sink);
		}
	}
}

__device__ __forceinline__ uint64_t effective_add(const adac_segment_desc &d) {
	// product rule: the frame of reference is added back only where it was subtracted
	return ((d.flags & ADAC_SEG_PACKED) && d.min != ADAC_NO_MIN) ? d.min : 0ull;
}

// ---------------------------------------------------------------------------------------------
// k_unpack — decode.  LDS: the packed image of one tile (<= 16 KiB + 32 B).
// ---------------------------------------------------------------------------------------------
template <typename U>
struct StoreSink {
	U *dst; // element `elem0` of the output
	uint32_t n;
	__device__ __forceinline__ void operator()(int32_t base, const U *vals, bool full) const {
		constexpr int K = 16 / (int)sizeof(U);
		if (full) {
			uint4 q;
			__builtin_memcpy(&q, vals, 16);
			// non-temporal: the decoded values are written once and read by somebody else much later; streamed past
			// the caches the stores cost 11 - 17 % less for the 4-byte types (u32 w 16: 0.1105 -> 0.0916 ms per
			// 100 M rows = 6.5 TB/s) and 0.5 - 1.5 % less for the 8-byte ones (profiles/r02_nontemporal_stores.json)
			typedef uint32_t v4u __attribute__((ext_vector_type(4)));
			v4u qq = {q.x, q.y, q.z, q.w};
			__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(dst + base));
		} else {
#pragma unroll
			for (int j = 0; j < K; j++) {
				if ((uint32_t)(base + j) < n) dst[base + j] = vals[j];
			}
		}
	}
};

template <typename U, bool RANGE, bool REC = false>
__global__ __launch_bounds__(kWorkgroup) void k_unpack(const adac_segment_desc *__restrict__ descs,
                                                       const TileRef *__restrict__ tiles, RangeArgs range,
                                                       const uint64_t *__restrict__ words, U *__restrict__ out,
                                                       const TileRec *__restrict__ recs = nullptr) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	if constexpr (REC) { // the tile's expanded record: one hop before the data loads
		const TileRec r = load_tile_rec(recs + blockIdx.x);
		const uint32_t bit0 = stage_packed(words + r.word_off, r.first, r.n, r.width, lds);
		__syncthreads();
		constexpr uint32_t KK = 16 / sizeof(U);
		StoreSink<U> sink {out + r.elem0, r.n};
		decode_tile<U>(reinterpret_cast<const uint32_t *>(lds), bit0, r.width, r.add, r.n, (uint32_t)(r.elem0 & (KK - 1)), sink);
		return;
	}
	adac_segment_desc d;
	uint32_t first, n;
	uint64_t elem0;
	if (RANGE) {
		d = descs[range.seg];
		const uint32_t done = blockIdx.x * (uint32_t)TILE;
		first = range.start + done;
		const uint32_t left = range.count - done;
		n = left < (uint32_t)TILE ? left : (uint32_t)TILE;
		elem0 = range.out_off + done;
	} else {
		const TileCtx t = resolve_tile<TILE>(descs, tiles);
		d = t.d;
		first = t.first;
		n = t.n;
		elem0 = t.elem0;
	}
	const uint32_t w = d.width;
	const uint32_t bit0 = stage_packed(words + d.word_off, first, n, w, lds);
	__syncthreads();
	const uint64_t add = effective_add(d);
	constexpr uint32_t K = 16 / sizeof(U);
	const uint32_t align = (uint32_t)(elem0 & (K - 1));
	StoreSink<U> sink {out + elem0, n};
	const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds);
	decode_tile<U>(lds32, bit0, w, add, n, align, sink);
}

// k_unpack_jobs — the same decode for a short list of (segment, row range) jobs handed over BY VALUE in the kernel
// arguments (scan_vector / scan_partial for several segments at once, compression_function.hpp:84-88): a workgroup
// finds its job with a scalar scan over at most kMaxUnpackJobs first-tile numbers, everything else is k_unpack's
// range form.
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_unpack_jobs(const UnpackJobTable table,
                                                            const uint64_t *__restrict__ words, U *__restrict__ out) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr uint32_t K = 16 / sizeof(U);
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	uint32_t j = 0;
	for (uint32_t i = 1; i < table.njobs; i++) { // uniform: scalar loads from the kernarg segment
		if (table.jobs[i].tile0 <= blockIdx.x) j = i;
	}
	const UnpackJob job = table.jobs[j];
	const uint32_t done = (blockIdx.x - job.tile0) * (uint32_t)TILE;
	const uint32_t first = job.start + done;
	const uint32_t left = job.count - done;
	const uint32_t n = left < (uint32_t)TILE ? left : (uint32_t)TILE;
	const uint64_t elem0 = job.out_off + done;
	const uint32_t bit0 = stage_packed(words + job.word_off, first, n, job.width, lds);
	__syncthreads();
	StoreSink<U> sink {out + elem0, n};
	decode_tile<U>(reinterpret_cast<const uint32_t *>(lds), bit0, job.width, job.add, n, (uint32_t)(elem0 & (K - 1)), sink);
}

// ---------------------------------------------------------------------------------------------
// k_scan_agg — fused scans (SUM, COUNT(range), selection bitmap), nothing materialised.
// ---------------------------------------------------------------------------------------------
// LDS form (widths the register path does not take: w < 4, w > 32, segments whose fields are not their values):
// a workgroup owns one ScanGroup work item — several tiles of ONE segment — and stages as many whole tiles as
// fit the 16 KiB LDS image at that segment's width (8 tiles of u64 at w <= 8, 2 at w = 32, ...), so ~16 KiB of
// HBM reads are in flight per workgroup at every width.  The aggregate is carried in registers across stages
// and flushed (wave reduce + one atomic per wave) once per work item.
template <typename U, bool WIDE, typename Sink>
__device__ __forceinline__ void decode_run(const uint32_t *lds32, uint32_t bit0, uint32_t w, uint64_t add, uint32_t n,
                                           Sink &&sink) {
	constexpr int K = 16 / (int)sizeof(U);
	constexpr uint32_t PER_ROUND = kWorkgroup * K;
	if (n % PER_ROUND != 0) {
		decode_rows<U, WIDE>(lds32, bit0, w, add, n, 0u, sink);
		return;
	}
	const uint32_t mlo = WIDE ? 0xffffffffu : mask32(w);
	const uint32_t mhi = WIDE ? mask32(w - 32u) : 0u;
	const uint32_t add_lo = (uint32_t)add;
	uint32_t bit = bit0 + __umul24(threadIdx.x * K, w);
	const uint32_t step = PER_ROUND * w;
	const uint32_t rounds = n / PER_ROUND;
#pragma unroll 1
	for (uint32_t r = 0; r < rounds; r++) {
		U vals[K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			uint32_t lo, hi;
			read_field<WIDE>(lds32, bit + j * w, mlo, mhi, lo, hi);
			if (sizeof(U) == 8) {
				vals[j] = (U)((((uint64_t)hi << 32) | lo) + add);
			} else {
				vals[j] = (U)(lo + add_lo);
			}
		}
		sink((int32_t)(r * PER_ROUND + threadIdx.x * K), vals, true);
		bit += step;
	}
}

// ---------------------------------------------------------------------------------------------
// Bit-width-templated, registers-only field walk for fused scans at W <= 32.  A lane owns whole 16-byte chunks
// of the packed stream (one coalesced global_load_dwordx4 + the next dword: a 160-bit window), shifts the
// window once so that its first field starts at bit 0, and then every field sits at a COMPILE-TIME position:
// one v_bfe_u32 (or v_alignbit + and for a dword straddler) per value, no LDS, no barrier, no per-value
// address arithmetic.  A lane decodes the fields that START in its chunk, so consecutive lanes cover
// consecutive rows and nothing is decoded twice.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ uint32_t field_of(const uint32_t (&nrm)[5], int j) {
	constexpr uint32_t mask = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
	const int pos = j * W, d = pos >> 5, sh = pos & 31;
	if (sh + W <= 32) return (nrm[d] >> sh) & mask;
	return __builtin_amdgcn_alignbit(nrm[d + 1], nrm[d], sh) & mask;
}

__device__ __forceinline__ bool scan_width_is_narrow(uint32_t w) { return w == 2u || w == 3u; }

// A workgroup's 64-byte work item through the scalar unit (see load_desc; measured: the vector loads of width / flags
// cost 3 - 5 % of a fused scan once the narrow-width test made width the first field needed,
// profiles/r03_scan_split_ab.json).
__device__ __forceinline__ ScanGroup load_scan_group(const ScanGroup *__restrict__ groups, uint32_t gi) {
	static_assert(sizeof(ScanGroup) == 64 && offsetof(ScanGroup, seg) == 32 && offsetof(ScanGroup, n_first) == 56,
	              "record layout");
	ScanGroup g;
	g.d = load_desc(&groups[gi].d);
	const uint32_t *__restrict__ p = reinterpret_cast<const uint32_t *>(groups + gi);
	g.seg = p[8];
	g.first = p[9];
	g.n = p[10];
	g.seg_groups = p[11];
	g.cell_first = p[12];
	g.cell_last = p[13];
	g.n_first = (uint16_t)(p[14] & 0xffffu);
	g.n_last = (uint16_t)(p[14] >> 16);
	g.pad = 0;
	return g;
}

// A bitmap word two scan groups share: each group leaves its bits in a record of its own, k_sel_merge_edges ORs the
// records of a word and stores it.  (A global atomicOr per shared word cost 13 % of the selection scan at w = 8:
// two L2 round trips at the end of every workgroup, profiles/r02_select_writeout_ablation.json.)
struct SelEdge {
	uint64_t word; // index of the 32-bit bitmap word
	uint32_t bits;
	uint32_t valid;
};

// ---------------------------------------------------------------------------------------------
// Selection-bitmap output of the filter scan (OP 3).  A workgroup owns the rows [first, first + n) of ONE segment
// (its ScanGroup), i.e. one contiguous bit string of the result.  Lanes OR their hit bits into a zeroed LDS image
// of that string (ds_or_b32; positions are p = (val_off & 31) + row, so image word i is bitmap word
// (val_off >> 5) + (p_base >> 5) + i), and after ONE barrier at the end the workgroup writes the image out: words
// wholly inside the group with plain coalesced stores — zero words included, so the result needs no clearing
// pass — and the two edge words, which neighbouring groups share, with a global atomicOr (a tiny kernel zeroes just
// those before the scan).  Nothing is stored inside the scan loop: on gfx9 stores and loads share vmcnt, and a
// store issued between a prefetch and its use made every flush wait for the store's round trip (measured: the
// per-block flush of the first version cost as much as the whole scan, profiles/r02_select_ablation.json).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kSelMaxRows = 65536;                 // rows of the largest scan group (ensure_scan_groups)
constexpr uint32_t kSelImageWords = kSelMaxRows / 32 + 4;
struct SelOut {
	uint32_t *img;      // the workgroup's LDS image, kSelImageWords words, zero before the first sel_or
	uint32_t *bitmap32; // the output bitmap viewed as little-endian 32-bit words
	uint32_t p_base;    // position of bit 0 of the image: ((val_off & 31) + first) & ~31
	int debug;          // diagnostic (adac_set_tuning "sel_debug"): 1 = no write-out, 2 = no emit at all (results wrong)
	SelEdge *edges;     // dense value spaces: the group's two records for the words it shares (nullptr: global atomicOr)
	bool nt;            // A/B: non-temporal bitmap stores
	// dense value spaces, arrival form (round 3): the edge cells and this group's ScanGroupRef arrival fields
	unsigned long long *cells;
	uint32_t cell_first, cell_last, n_first, n_last;
};

// ---------------------------------------------------------------------------------------------
// Arrival cells: results finished INSIDE the scan kernel.  Every party adds (ORs) its part into a cell and then counts
// itself in; the one that completes the expected number takes the cell's content, stores the result with a plain
// store and leaves the cell zero for the next call.  No clearing pass before the scan and no merge kernel after it
// (at C2 they cost 3.4 us of a 73 us SUM and 8.7 us of an 85 us selection scan: launch gaps, not work).  The two
// atomics of a party are ordered by waiting for the first one's acknowledgement (vmcnt): both are performed at the
// agent's point of coherence, so whoever sees the completing count also sees every part.
// ---------------------------------------------------------------------------------------------
// A 64-bit sum (mod 2^64) does not fit one word together with the arrivals, so its two 32-bit halves travel in TWO words,
// each with its own arrival count in bits 40 and up (the halves of up to 256 parts add up below 2^40), both atomics in
// flight together: one round trip.  Whoever completes both words knows the total.  When two different parties complete
// one word each (their atomics interleaved), they meet at a third word: the first leaves its half there, the second
// takes it, stores the total and clears the word.  Segments of more than 256 groups: add, wait, count (two trips).
constexpr uint32_t kArriveShift = 40;
// (`swapped`: test hook, sel_debug 7 — this party sends the high half first and waits for it, so that two parties of a
// segment complete one word each and the meeting at the third word is exercised)
__device__ __forceinline__ void arrive_sum(unsigned long long *cell, uint64_t part, uint32_t expected,
                                           uint64_t *__restrict__ dst, bool swapped) {
	if (expected <= 1u) {
		*dst = part;
		return;
	}
	if (expected > 256u) {
		if (part) {
			__hip_atomic_fetch_add(cell, (unsigned long long)part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
		const unsigned long long before = __hip_atomic_fetch_add(cell + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (before + 1ull == expected) {
			*dst = __hip_atomic_exchange(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			__hip_atomic_store(cell + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		return;
	}
	constexpr unsigned long long one = 1ull << kArriveShift;
	const uint64_t lo = part & 0xffffffffull, hi = part >> 32;
	unsigned long long oa, ob;
	if (swapped) {
		ob = __hip_atomic_fetch_add(cell + 1, one | hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_sleep(64);
		oa = __hip_atomic_fetch_add(cell, one | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else {
		oa = __hip_atomic_fetch_add(cell, one | lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		ob = __hip_atomic_fetch_add(cell + 1, one | hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	const bool last_a = (uint32_t)(oa >> kArriveShift) + 1u == expected;
	const bool last_b = (uint32_t)(ob >> kArriveShift) + 1u == expected;
	const uint64_t ta = (oa & (one - 1ull)) + lo, tb = (ob & (one - 1ull)) + hi; // <= 2^40 each
	if (last_a) __hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (last_b) __hip_atomic_store(cell + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (last_a && last_b) {
		*dst = (tb << 32) + ta;
	} else if (last_a || last_b) {
		const unsigned long long other =
		    __hip_atomic_exchange(cell + 2, (1ull << 63) | (last_a ? ta : tb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (other) { // the completer of the other half was here first
			const uint64_t o = other & ~(1ull << 63);
			*dst = last_a ? ((o << 32) + ta) : ((tb << 32) + o);
			__hip_atomic_store(cell + 2, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	}
}
// Parts that fit 40 bits and add up below 2^40 (a segment's hit count: at most 2^32 rows) travel WITH the arrival in
// one atomic — arrivals << 40 | sum — so a party needs a single round trip and the completing one knows the total
// from the value it got back.
__device__ __forceinline__ void arrive_count(unsigned long long *cell, uint64_t part, uint32_t expected,
                                             uint64_t *__restrict__ dst) {
	if (expected <= 1u) {
		*dst = part;
		return;
	}
	const unsigned long long before =
	    __hip_atomic_fetch_add(cell, (1ull << kArriveShift) | part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if ((uint32_t)(before >> kArriveShift) + 1u == expected) {
		*dst = (before & ((1ull << kArriveShift) - 1ull)) + part;
		__hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}
// Bits of one bitmap word: the parties' bits are disjoint (different rows), so OR is ADD and they too travel with the
// arrival: arrivals << 32 | bits.
__device__ __forceinline__ void arrive_or(unsigned long long *cell, uint32_t part, uint32_t expected,
                                          uint32_t *__restrict__ dst) {
	if (expected <= 1u) {
		*dst = part;
		return;
	}
	const unsigned long long before =
	    __hip_atomic_fetch_add(cell, (1ull << 32) | part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if ((uint32_t)(before >> 32) + 1u == expected) {
		*dst = (uint32_t)before | part;
		__hip_atomic_store(cell, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
}

__device__ __forceinline__ void sel_or(const SelOut &o, uint32_t p, uint32_t bits, uint32_t nbits) {
	if (bits) {
		const uint32_t rel = p - o.p_base, sh = rel & 31u, idx = rel >> 5;
		atomicOr(&o.img[idx], bits << sh);
		if (sh + nbits > 32u) atomicOr(&o.img[idx + 1], bits >> (32u - sh));
	}
}

// after a workgroup barrier: image -> bitmap, for the group's positions [p0, p1)
__device__ __forceinline__ void sel_write_out(const SelOut &o, uint32_t *__restrict__ seg_words32, uint32_t p0,
                                              uint32_t p1) {
	const uint32_t nwords = (p1 - o.p_base + 31u) >> 5;
	for (uint32_t i = threadIdx.x; i < nwords; i += kWorkgroup) {
		const uint32_t v = o.img[i];
		const uint32_t wb = o.p_base + 32u * i;
		uint32_t *g = seg_words32 + (wb >> 5);
		if (wb >= p0 && wb + 32u <= p1) {
			if (o.nt) { // A/B (sel_debug 6): the bitmap words streamed past the caches
				__builtin_nontemporal_store(v, g);
			} else {
				*g = v;
			}
		} else if (o.cells) { // only the first and the last word of the group can be covered in part
			const bool head = i == 0u;
			arrive_or(o.cells + (head ? o.cell_first : o.cell_last), v, head ? o.n_first : o.n_last, g);
		} else if (o.edges) { // only the first and the last word of the group can be shared
			SelEdge e;
			e.word = (uint64_t)(g - o.bitmap32);
			e.bits = v;
			e.valid = 1u;
			o.edges[i == 0u ? 0 : 1] = e;
		} else if (v) {
			atomicOr(g, v);
		}
	}
	if (o.edges && threadIdx.x < 2u) { // the slots this group does not need say so (slot 1 also when nwords <= 1)
		const uint32_t last = nwords ? nwords - 1u : 0u;
		const uint32_t i = threadIdx.x == 0u ? 0u : last;
		const uint32_t wb = o.p_base + 32u * i;
		const bool whole = nwords == 0u || (wb >= p0 && wb + 32u <= p1);
		if (whole || (threadIdx.x == 1u && last == 0u)) {
			SelEdge e;
			e.word = 0;
			e.bits = 0;
			e.valid = 0u;
			o.edges[threadIdx.x] = e;
		}
	}
}

// Aggregate the fields of one chunk.  Exactness contract: SUM equals the sum of the MATERIALISED values, each
// widened to 64 bits according to T's signedness (what SQL SUM over the column computes), mod 2^64.  Partial
// sums of a chunk stay in 32-bit registers whenever MAXV fields of W bits cannot overflow them.
// Range predicate lo <= v <= hi in T's own order (signed for the INT types), through the order-preserving map
// B(v) = bits(v) ^ sbit:  B(v) - blo <= bspan as unsigned numbers.  `==`, `<`, `<=`, `>`, `>=`, BETWEEN are all
// instances (column_segment.cpp:575-844 FilterSelection's comparison kinds).
struct RangePred {
	uint64_t blo, bspan, sbit;
};
// The same predicate moved into the packed-field domain of one segment: (f - flo) <= span in 32-bit arithmetic.
struct FieldRange {
	uint32_t flo, span;
	bool any;
};

// How a segment's stored fields relate to its values in T's own order (signed for the INT types):
//   SEG_LINEAR   value = min + f without leaving T's range: sums are linear and a range predicate moves into the
//                field domain.  Every segment the append path packs is linear (its sign-extended min/max order
//                never packs a mixed-sign range, succinct.cpp:286-287).
//   SEG_RAW      the stored bits are the value (unpacked slots, or packed without a frame of reference).
//   SEG_WRAPS    packed with a min, but min + f crosses T's sign boundary: only BitCompressFromUncompressed's
//                zero-extended order (column_segment.cpp:405-420) can produce it, e.g. {INT_MAX, INT_MIN} -> w = 1.
//                Such a segment is always decoded to values first (LDS path).
enum SegKind { SEG_LINEAR, SEG_RAW, SEG_WRAPS };

template <typename U>
__device__ __forceinline__ SegKind seg_kind(const adac_segment_desc &d, uint64_t sbit) {
	if (!((d.flags & ADAC_SEG_PACKED) && d.min != ADAC_NO_MIN)) return SEG_RAW;
	if (sbit == 0) return SEG_LINEAR; // unsigned T: min + f is a stored value, it cannot wrap
	const uint64_t bmin = (uint64_t)(U)d.min ^ sbit;
	const uint64_t maxf = d.width >= 64 ? ~0ull : ((1ull << d.width) - 1ull);
	const uint64_t top = bmin + maxf;
	return (top >= bmin && top <= (uint64_t)(U)~(U)0) ? SEG_LINEAR : SEG_WRAPS;
}

// min widened to 64 bits according to T's signedness: with it, value64 = f + add64 on a linear segment
template <typename U>
__device__ __forceinline__ uint64_t widened_min(const adac_segment_desc &d, uint64_t sbit) {
	return ((uint64_t)(U)d.min ^ sbit) - sbit;
}

template <typename U>
__device__ __forceinline__ FieldRange field_range(const RangePred &p, const adac_segment_desc &d, uint32_t mask,
                                                  bool linear) {
	FieldRange r {0u, 0u, false};
	if (linear) {
		// v = min + f without wrap in T's order, so B(v) = B(min) + f
		const uint64_t bmin = (uint64_t)(U)d.min ^ p.sbit;
		const uint64_t bhi = p.blo + p.bspan;
		if (bhi < bmin) return r;
		const uint64_t flo = p.blo > bmin ? p.blo - bmin : 0ull;
		if (flo > (uint64_t)mask) return r;
		const uint64_t fhi = bhi - bmin < (uint64_t)mask ? bhi - bmin : (uint64_t)mask;
		r.flo = (uint32_t)flo;
		r.span = (uint32_t)(fhi - flo);
	} else {
		// the stored bits are the value (unpacked slots, or a segment without a frame of reference): unsigned T,
		// or a 32-bit signed T, where B(v) = f ^ 2^31 = f + 2^31 (mod 2^32) folds into the lower bound
		r.flo = (uint32_t)p.blo - (uint32_t)p.sbit;
		r.span = (uint32_t)p.bspan;
	}
	r.any = true;
	return r;
}

template <int W>
struct ChunkSum {
	static constexpr int MAXV = (128 + W - 1) / W;
	static constexpr bool kFields32 = ((uint64_t)MAXV << W) <= 0xffffffffull; // MAXV fields fit a u32 sum
	uint32_t p32 = 0;
	uint64_t p64 = 0;
	uint32_t nvalid = 0; // rows aggregated by add_masked
	__device__ __forceinline__ void add(uint32_t f) {
		if (kFields32) p32 += f; else p64 += f;
	}
	__device__ __forceinline__ void add_masked(uint32_t f, uint32_t all_ones_if_valid) { // the caller sets nvalid
		if (kFields32) p32 += f & all_ones_if_valid; else p64 += f & all_ones_if_valid;
	}
	// rows: the number of rows aggregated with add(); rows added with add_masked() are counted in nvalid.
	// SUM is linear on the segments this path takes: sum(value64) = sum(fields) + rows * add64.
	__device__ __forceinline__ uint64_t total(uint32_t rows, uint64_t add64) const {
		return (uint64_t)p32 + p64 + (uint64_t)(rows + nvalid) * add64;
	}
};

// `nbits` (<= 32) mask bits starting at element index e: bit j of the result = element e + j; bits past nbits are
// unspecified.  The second word is read only when the wanted bits reach into it, so a mask of exactly
// ceil(span / 64) words is never over-read as long as nbits is clipped to the rows that exist.
// (This form costs a branch and a wait of its own per call; it is kept where the mask is an OPTION of a kernel whose
// hot path runs without one — re-compaction, k_analyze_packed: with the branch-free form below k_repack_g<u8> went
// from 89 to 97 VGPRs and lost an occupancy step and 6 - 10 % on columns that have no mask at all.)
__device__ __forceinline__ uint32_t validity_window(const uint64_t *__restrict__ validity, uint64_t e, uint32_t nbits) {
	if (nbits == 0u) return 0u; // nothing wanted: e may lie past the mask
	const uint32_t sh = (uint32_t)(e & 63);
	uint64_t wnd = validity[e >> 6] >> sh;
	if (sh + nbits > 64u) wnd |= validity[(e >> 6) + 1] << (64 - sh);
	return (uint32_t)wnd;
}
// The same, BRANCH-FREE, for kernels that always read a mask: the second word is the one that holds the LAST wanted
// bit — always inside the mask, and the first word again when the bits do not reach into the next one — so both loads
// are unconditional and in flight together.  (Round 3: with the conditional form a lane's windows went out one round
// trip after the other: k_gather_c 0.31 -> 0.26 ms at C2, profiles/r03_gather_compaction.json.)
__device__ __forceinline__ uint32_t validity_window_pair(const uint64_t *__restrict__ validity, uint64_t e, uint32_t nbits) {
	const uint64_t end = e + nbits;
	const uint64_t last = end - (end != 0ull ? 1ull : 0ull); // the last wanted bit (the bit before e if none is wanted)
	const uint64_t first = e < last ? e : last;
	const uint32_t sh = (uint32_t)(first & 63);
	const uint64_t w0 = validity[first >> 6], w1 = validity[last >> 6];
	const uint64_t wnd = (w0 >> sh) | ((w1 << 1) << (63u - sh)); // w1 == w0 when no bit of the next word is wanted
	return nbits ? (uint32_t)wnd : 0u;
}

// hits = 2 * hits + (d <= span).  The compare lands in VCC and is consumed as the carry-in of ONE add: two vector
// instructions per field.  (Left to itself the compiler materialises every compare with v_cndmask and merges pairs
// with v_or3 / v_lshl — three to four instructions per field plus s_nop hazard fillers, profiles/r02_select_isa.txt.)
__device__ __forceinline__ void hit_shift_in(uint32_t &hits, uint32_t d, uint32_t span) {
	asm volatile("v_cmp_ge_u32 vcc, %2, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(hits) : "v"(d), "s"(span) : "vcc");
}

template <int W, typename U, int OP, bool V>
__device__ __forceinline__ void scan_run_w(const uint4 *__restrict__ seg16, uint32_t r0, uint32_t r1,
                                           const adac_segment_desc &d, const RangePred &pred, bool linear,
                                           const uint64_t *__restrict__ validity, const SelOut &sel_out,
                                           uint64_t &acc) {
	constexpr int MAXV = (128 + W - 1) / W;
	constexpr uint32_t mask = W >= 32 ? 0xffffffffu : ((1u << W) - 1u);
	constexpr bool PRED = OP == 1 || OP == 3;
	constexpr uint32_t STRIDE = (uint32_t)kWorkgroup;
	const uint32_t c0 = (uint32_t)(((uint64_t)r0 * W) >> 7);       // r0 is a multiple of 128 rows
	const uint32_t c1 = (uint32_t)(((uint64_t)r1 * W + 127) >> 7);
	const uint32_t clast = (uint32_t)(((uint64_t)d.count * W + 127) >> 7) - 1; // last chunk holding data bits
	const uint64_t add = linear ? widened_min<U>(d, pred.sbit) : 0ull;
	FieldRange fr {0u, 0u, true};
	if (PRED) {
		fr = field_range<U>(pred, d, mask, linear);
		if (!fr.any) return; // zonemap-style skip: no row of this segment can satisfy the predicate
	}
	// lane -> chunk map: the workgroup strides over the run together
	uint32_t L = c0 + threadIdx.x;
	const uint32_t lend = c1;
	if (L >= lend) return;
	const uint32_t sh0 = (uint32_t)(d.val_off & 31u); // selection: bit position of row 0 inside its bitmap word
	// software pipeline: the next chunk's loads are issued (unconditionally, index clamped into the segment)
	// before the current chunk is decoded, so a wave always has a load in flight
	const uint32_t Lc = L < clast ? L : clast;
	uint4 q = seg16[Lc];
	uint32_t e = reinterpret_cast<const uint32_t *>(seg16 + (Lc < clast ? Lc + 1 : clast))[0];
	// V: the two mask words that hold a chunk's rows travel with the chunk — requested a round ahead, unconditionally,
	// word indices (relative to the segment's first word, 32 bits) clamped to the word of the run's last row.  Looked
	// up inside the loop after the walk, the mask cost the masked scans 15 - 40 % (u64 w 8: SUM 5.07 -> 3.16 TB/s).
	const uint64_t *__restrict__ vseg = V ? validity + (d.val_off >> 6) : nullptr;
	const uint32_t vsh0 = (uint32_t)(d.val_off & 63u);
	const uint32_t vend = (vsh0 + r1 - 1u) >> 6;
	auto mask_words = [&](uint32_t Lx, uint64_t &m0, uint64_t &m1) {
		const uint32_t ix0 = (128u * Lx + (W - 1)) / W;
		const uint32_t wi = (vsh0 + (ix0 < r1 ? ix0 : r1)) >> 6;
		m0 = vseg[wi < vend ? wi : vend];
		m1 = vseg[wi + 1u < vend ? wi + 1u : vend];
	};
	// the 64 mask bits from row `at` on (bits past the run's last row are unspecified)
	auto mask_window = [&](uint64_t m0, uint64_t m1, uint32_t at) -> uint64_t {
		const uint32_t sh = (vsh0 + at) & 63u;
		return (m0 >> sh) | ((m1 << 1) << (63u - sh));
	};
	uint64_t vm0 = 0, vm1 = 0;
	if (V) mask_words(Lc, vm0, vm1);
	constexpr uint32_t adv = STRIDE;
	uint32_t wave_count = 0; // wave-uniform (scalar) COUNT accumulator; lane 0 of the wave leaves the loop last
	for (; L < lend; L += adv) {
		const uint32_t Lp = L + adv < clast ? L + adv : clast;
		const uint4 qn = seg16[Lp];
		const uint32_t en = reinterpret_cast<const uint32_t *>(seg16 + (Lp < clast ? Lp + 1 : clast))[0];
		uint64_t vn0 = 0, vn1 = 0;
		if (V) mask_words(Lp, vn0, vn1);
		const uint64_t vwnd = V ? mask_window(vm0, vm1, ((128u * L + (W - 1)) / W) < r1 ? ((128u * L + (W - 1)) / W) : r1) : 0ull;
		vm0 = vn0;
		vm1 = vn1;
		const uint32_t i0 = (128u * L + (W - 1)) / W; // first row starting in this chunk
		const uint32_t o0 = i0 * W - 128u * L;        // its bit offset, < W <= 32
		uint32_t nrm[5];
		nrm[0] = __builtin_amdgcn_alignbit(q.y, q.x, o0);
		nrm[1] = __builtin_amdgcn_alignbit(q.z, q.y, o0);
		nrm[2] = __builtin_amdgcn_alignbit(q.w, q.z, o0);
		nrm[3] = __builtin_amdgcn_alignbit(e, q.w, o0);
		nrm[4] = e >> o0;
		q = qn;
		e = en;
		if (OP == 2) { // load-only probe (diagnostic): same loop and loads, no field walk
			acc += nrm[0] + nrm[1] + nrm[2] + nrm[3] + nrm[4];
			continue;
		}
		const uint32_t starting = (128u - o0 + (W - 1)) / W; // rows starting in the chunk: MAXV-1 or MAXV
		const uint32_t lim = r1 > i0 ? r1 - i0 : 0u;
		if (OP == 1 && !V) {
			// COUNT through the scalar unit: when every active lane's chunk is interior, each field's compare
			// lands in an SGPR pair and s_bcnt1 adds its population to a wave-uniform counter — three vector
			// instructions per field (extract, subtract, compare) instead of five for the per-lane hit mask
			if (__builtin_amdgcn_ballot_w64(starting > lim) == 0ull) {
				uint32_t c = 0;
#pragma unroll
				for (int j = 0; j < MAXV - 1; j++) {
					c += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((field_of<W>(nrm, j) - fr.flo) <= fr.span));
				}
				const bool last_hit = (field_of<W>(nrm, MAXV - 1) - fr.flo) <= fr.span;
				c += (uint32_t)__popcll(
				    __builtin_amdgcn_ballot_w64(last_hit && (128 % W == 0 || starting == (uint32_t)MAXV)));
				wave_count += c;
				continue;
			}
		}
		if (PRED && W < 4) {
			// widths 2 and 3 hold 64 / 43 fields per chunk: the hit mask is built as two 32-bit halves
			const uint32_t have = starting < lim ? starting : lim; // 0 for lanes past the run
			uint32_t lo = 0, hi = 0;
#pragma unroll
			for (int j = 31; j >= 0; j--) hit_shift_in(lo, field_of<W>(nrm, j) - fr.flo, fr.span);
#pragma unroll
			for (int j = MAXV - 1; j >= 32; j--) hit_shift_in(hi, field_of<W>(nrm, j) - fr.flo, fr.span);
			const uint32_t n_lo = have < 32u ? have : 32u, n_hi = have > 32u ? have - 32u : 0u;
			lo &= n_lo >= 32u ? 0xffffffffu : ((1u << n_lo) - 1u);
			hi &= n_hi >= 32u ? 0xffffffffu : ((1u << n_hi) - 1u);
			const uint32_t at = i0 < r1 ? i0 : r1; // keeps a lane's element range inside this run
			if (V) { // NULL rows take no part
				lo &= (uint32_t)vwnd;
				hi &= (uint32_t)(vwnd >> 32);
			}
			acc += (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
			if (OP == 3 && sel_out.debug < 2) {
				sel_or(sel_out, sh0 + at, lo, n_lo);
				if (n_hi) sel_or(sel_out, sh0 + at + 32u, hi, n_hi);
			}
			continue;
		}
		if (PRED) {
			// bit j of `hits` = row i0 + j satisfies the predicate: built top-down so that each field costs
			// extract, subtract, compare and one add-with-carry (hits = 2 * hits + hit)
			const uint32_t have = starting < lim ? starting : lim; // 0 for lanes past the run
			uint32_t hits = 0;
#pragma unroll
			for (int j = MAXV - 1; j >= 0; j--) {
				hit_shift_in(hits, field_of<W>(nrm, j) - fr.flo, fr.span);
			}
			hits &= have >= 32u ? 0xffffffffu : ((1u << have) - 1u);
			const uint32_t at = i0 < r1 ? i0 : r1; // keeps a lane's element range inside this run
			// NULL rows (DuckDB validity mask over the element index space) take no part
			if (V) hits &= (uint32_t)vwnd;
			acc += (uint32_t)__popc(hits);
			if (OP == 3 && sel_out.debug < 2) sel_or(sel_out, sh0 + at, hits, have);
			continue;
		}
		ChunkSum<W> agg;
		uint32_t nv = 0;
		if (V) {
			// rows that exist AND are valid, as one mask: a field then costs its extract, ONE signed bit-field extract
			// (0 / -1 from the row's bit), an AND and an add; the rows are counted once per chunk (v_bcnt).  (Until round
			// 3: a row test, the bit, its negation, a count and the masked add per field — six to eight instructions.)
			const uint32_t have = starting < lim ? starting : lim;
			const uint32_t vb = (uint32_t)vwnd & (have >= 32u ? 0xffffffffu : ((1u << have) - 1u));
#pragma unroll
			for (int j = 0; j < MAXV; j++) agg.add_masked(field_of<W>(nrm, j), (uint32_t)__builtin_amdgcn_sbfe((int)vb, j, 1));
			agg.nvalid = (uint32_t)__popc(vb);
		} else if (starting <= lim) { // interior chunk: only the last slot may be absent
			nv = starting;
#pragma unroll
			for (int j = 0; j < MAXV - 1; j++) agg.add(field_of<W>(nrm, j));
			if (128 % W == 0 || starting == (uint32_t)MAXV) agg.add(field_of<W>(nrm, MAXV - 1));
		} else { // the run ends inside this chunk
			nv = lim;
#pragma unroll
			for (int j = 0; j < MAXV; j++) {
				if ((uint32_t)j < nv) agg.add(field_of<W>(nrm, j));
			}
		}
		acc += agg.total(nv, add);
	}
	if (OP == 1 && !V && (threadIdx.x & 63u) == 0u) acc += wave_count;
}

// The widths are dealt to TWO kernels: widths 2 and 3 (u8 / u16 columns of flags and small codes; a chunk holds 64 / 43
// fields, walked in registers by every form but the SUM under a validity mask) live in k_scan_agg<.., NARROW = true>,
// which only the scan groups of such segments run.  Inlined into the common kernel their unrolled bodies took
// k_scan_agg<u64, sum> from 42 to 74 VGPRs (occupancy 8 -> 6 waves per SIMD) and cost every width 8 .. 32 about 10 %
// (profiles/r02i vs r02k); tests/test_kernel_budget.py now holds the register budget of the common kernel.
template <typename U, int OP, bool V, bool NARROW>
__device__ __forceinline__ void scan_run_dispatch(uint32_t w, const uint4 *__restrict__ seg16, uint32_t r0,
                                                  uint32_t r1, const adac_segment_desc &d, const RangePred &pred,
                                                  bool linear, const uint64_t *__restrict__ validity,
                                                  const SelOut &sel_out, uint64_t &acc) {
	if constexpr (NARROW) {
		if (w == 2u) {
			scan_run_w<2, U, OP, V>(seg16, r0, r1, d, pred, linear, validity, sel_out, acc);
		} else if (w == 3u) {
			scan_run_w<3, U, OP, V>(seg16, r0, r1, d, pred, linear, validity, sel_out, acc);
		}
	} else {
		switch (w) {
#define ADAC_W(N) case N: scan_run_w<N, U, OP, V>(seg16, r0, r1, d, pred, linear, validity, sel_out, acc); break;
			ADAC_W(4) ADAC_W(5) ADAC_W(6) ADAC_W(7) ADAC_W(8) ADAC_W(9) ADAC_W(10) ADAC_W(11) ADAC_W(12) ADAC_W(13)
			ADAC_W(14) ADAC_W(15) ADAC_W(16) ADAC_W(17) ADAC_W(18) ADAC_W(19) ADAC_W(20) ADAC_W(21) ADAC_W(22)
			ADAC_W(23) ADAC_W(24) ADAC_W(25) ADAC_W(26) ADAC_W(27) ADAC_W(28) ADAC_W(29) ADAC_W(30) ADAC_W(31) ADAC_W(32)
#undef ADAC_W
		default: break;
		}
	}
}

// Expansion of the scans' work items: one thread per group copies its segment's CURRENT descriptor next to the
// group's row range, so the scan kernel reads one 64-byte record (a single hop) before it can issue data loads.
__global__ void k_expand_groups(const adac_segment_desc *__restrict__ descs, const ScanGroupRef *__restrict__ refs,
                                uint64_t ngroups, ScanGroup *__restrict__ groups, uint32_t *__restrict__ narrow_idx,
                                uint32_t *__restrict__ narrow_count) {
	const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (g >= ngroups) return;
	const ScanGroupRef r = refs[g];
	ScanGroup out;
	out.d = descs[r.seg];
	out.seg = r.seg;
	out.first = r.first;
	out.n = r.rows;
	out.seg_groups = r.seg_groups;
	out.cell_first = r.cell_first;
	out.cell_last = r.cell_last;
	out.n_first = r.n_first;
	out.n_last = r.n_last;
	out.pad = 0;
	groups[g] = out;
	// the groups of segments at widths 2 and 3 are also listed for the narrow scan kernel (order of arrival: the list
	// only says who runs, every result is an exact integer)
	if (scan_width_is_narrow(out.d.width)) narrow_idx[atomicAdd(narrow_count, 1u)] = (uint32_t)g;
}

// After a selection scan over a DENSE value space (segments back to back): the words two or more groups share.  The
// scan wrote every word that lies inside one group whole and left two records per group for its first / last word
// when they are partial; records of one word are near neighbours in group order (slots that are not needed lie
// between them).  The first record of a word ORs its followers and stores the word; thread 0 also zeroes the odd 32-bit
// half of the last 64-bit word.  No atomics, and no clearing pass before the scan.
__global__ void k_sel_merge_edges(const SelEdge *__restrict__ edges, uint64_t nrec, uint32_t *__restrict__ bitmap32,
                                  uint64_t tail_word) {
	const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (r == 0 && tail_word != ~0ull) bitmap32[tail_word] = 0u;
	if (r >= nrec) return;
	const SelEdge me = edges[r];
	if (!me.valid) return;
	// a word holds 32 rows and a group at least one: the records of one word lie within 64 slots of each other (the
	// runs of unused slots between the records of DIFFERENT words can be long: the searches are bounded)
	const uint64_t lo = r > 64 ? r - 64 : 0, hi = r + 64 < nrec ? r + 64 : nrec - 1;
	for (uint64_t p = r; p > lo;) { // is an earlier record of the same word there?  (skip the unused slots)
		--p;
		const SelEdge q = edges[p];
		if (!q.valid) continue;
		if (q.word == me.word) return; // not the first one
		break;
	}
	uint32_t bits = me.bits;
	for (uint64_t p = r + 1; p <= hi; p++) {
		const SelEdge q = edges[p];
		if (!q.valid) continue;
		if (q.word != me.word) break;
		bits |= q.bits;
	}
	bitmap32[me.word] = bits;
}

// OP 0: SUM, 1: COUNT(lo <= v <= hi), 2: load-only probe, 3: COUNT + selection bitmap; V: validity mask given.
// One workgroup per ScanGroup: up to scan_tiles_per_wg consecutive tiles of ONE segment.
// NARROW: the kernel of the groups at widths 2 and 3 (listed in narrow_idx by k_expand_groups); the common kernel
// leaves those groups alone.  The SUM under a validity mask has no narrow form (its widths 2 and 3 take the LDS path).
template <int OP, bool V> constexpr bool kScanHasNarrowKernel = !(OP == 0 && V) && OP != 2;

template <typename U, int OP, bool V, bool NARROW>
__global__ __launch_bounds__(kWorkgroup, NARROW ? 4 : 6) void k_scan_agg(
    const ScanGroup *__restrict__ groups, const uint32_t *__restrict__ narrow_idx, int templated,
    const uint64_t *__restrict__ words, RangePred pred, const uint64_t *__restrict__ validity,
    uint64_t *__restrict__ result, uint32_t *__restrict__ bitmap32, SelEdge *__restrict__ edges,
    unsigned long long *__restrict__ res_cells, uint32_t *__restrict__ edge_cells, uint32_t last_group,
    uint64_t tail_word) {
	// LDS: the packed image of one stage of the fallback path (the register path uses none) and, for the selection
	// scan, the bitmap image of the group.  The selection scan halves the stage so that both fit 16.4 KiB: at
	// 24.6 KiB only six workgroups fit a CU instead of eight, and these kernels are bound by the bytes a CU keeps
	// in flight (measured: 3.4 -> 4.2 TB/s at w = 8 from the occupancy alone).
	constexpr uint32_t kStageBytes = OP == 3 ? kTileBytes / 2 : kTileBytes;
	constexpr uint32_t PER_ROUND = kWorkgroup * (16 / sizeof(U)); // rows one decode round of the workgroup covers
	__shared__ uint4 lds[kStageBytes / 16 + 2];
	__shared__ uint32_t sel_img[OP == 3 ? kSelImageWords : 1];
	const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds);
	const uint32_t gi = NARROW ? narrow_idx[blockIdx.x] : blockIdx.x;
	const ScanGroup g = load_scan_group(groups, gi);
	const adac_segment_desc &d = g.d;
	if (!NARROW && kScanHasNarrowKernel<OP, V> && scan_width_is_narrow(d.width)) return; // the narrow kernel's group
	const bool arrive_swapped = (templated >> 1) == 7 && (gi & 1u); // test hook of arrive_sum
	SelOut sel_out {sel_img, bitmap32, 0u, (templated >> 1) >= 6 ? 0 : (templated >> 1),
	                edges ? edges + 2u * (uint64_t)gi : nullptr, (templated >> 1) == 6,
	                reinterpret_cast<unsigned long long *>(edge_cells), g.cell_first, g.cell_last, g.n_first, g.n_last};
	templated &= 1;
	const uint32_t sel_p0 = (uint32_t)(d.val_off & 31u) + g.first; // the group's positions [sel_p0, sel_p0 + n)
	if (OP == 3) {
		sel_out.p_base = sel_p0 & ~31u;
		const uint32_t nwords = (sel_p0 + g.n - sel_out.p_base + 31u) >> 5;
		for (uint32_t i = threadIdx.x; i < nwords; i += kWorkgroup) sel_img[i] = 0u;
		__syncthreads();
	}
	uint64_t acc = 0;
	const uint32_t w = d.width;
	const SegKind kind = seg_kind<U>(d, pred.sbit);
	// the register path works on fields: linear segments, and raw ones whose field IS the value it needs
	// (unsigned T; for the predicates also 32-bit signed T, whose order is a shift of the field's)
	const bool by_field = kind == SEG_LINEAR ||
	                      (kind == SEG_RAW && (pred.sbit == 0 || ((OP == 1 || OP == 3) && sizeof(U) == 4)));
	constexpr uint32_t kMinRegisterWidth = kScanHasNarrowKernel<OP, V> ? 2u : 4u;
	if (templated && w >= kMinRegisterWidth && w <= 32 && (uint64_t)d.count * w < (1ull << 31) && by_field) {
		// width-templated register path over the whole group, no LDS
		const uint4 *seg16 = reinterpret_cast<const uint4 *>(words + d.word_off);
		scan_run_dispatch<U, OP, V, NARROW>(w, seg16, g.first, g.first + g.n, d, pred, kind == SEG_LINEAR, validity, sel_out,
		                            acc);
	} else {
		// the group in stages of as many whole decode rounds as fit the image (a round of 64-bit fields is 4 KiB)
		uint32_t stage_rows = ((8u * kStageBytes) / w) / PER_ROUND * PER_ROUND;
		stage_rows = stage_rows < PER_ROUND ? PER_ROUND : stage_rows;
		for (uint32_t done = 0; done < g.n;) {
			const uint32_t first = g.first + done; // a multiple of PER_ROUND rows: its bits start on a 16-byte boundary
			const uint32_t left = g.n - done;
			const uint32_t n = left < stage_rows ? left : stage_rows;
			const uint32_t bit0 = stage_packed(words + d.word_off, first, n, w, lds);
			__syncthreads();
			const uint64_t elem0 = d.val_off + first;
			auto sink = [&](int32_t base, const U *vals, bool full) {
				constexpr int KK = 16 / (int)sizeof(U);
				const uint32_t rows_here = full || n - (uint32_t)base >= (uint32_t)KK ? (uint32_t)KK : n - (uint32_t)base;
				const uint32_t vbits = V ? validity_window_pair(validity, elem0 + (uint32_t)base, rows_here) : 0xffffffffu;
				if (OP == 3) { // decode_run walks with align 0: base >= 0, lanes ascending
					uint32_t hits = 0;
#pragma unroll
					for (int j = 0; j < KK; j++) {
						hits |= (((((uint64_t)vals[j] ^ pred.sbit) - pred.blo) <= pred.bspan) ? 1u : 0u) << j;
					}
					const uint32_t rest = n - (uint32_t)base;
					const uint32_t have = full || rest >= (uint32_t)KK ? (uint32_t)KK : rest;
					hits &= vbits & ((1u << have) - 1u);
					acc += (uint32_t)__popc(hits);
					if (sel_out.debug < 2) sel_or(sel_out, (uint32_t)(d.val_off & 31u) + first + (uint32_t)base, hits, have);
					return;
				}
#pragma unroll
				for (int j = 0; j < KK; j++) {
					if ((full || (uint32_t)(base + j) < n) && ((vbits >> j) & 1u)) {
						if (OP == 1) {
							acc += (((uint64_t)vals[j] ^ pred.sbit) - pred.blo) <= pred.bspan ? 1ull : 0ull;
						} else {
							acc += ((uint64_t)vals[j] ^ pred.sbit) - pred.sbit; // widened by T's signedness
						}
					}
				}
			};
			if (sizeof(U) == 8 && w > 32) {
				decode_run<U, true>(lds32, bit0, w, effective_add(d), n, sink);
			} else {
				decode_run<U, false>(lds32, bit0, w, effective_add(d), n, sink);
			}
			__syncthreads(); // the image is rewritten by the next stage
			done += n;
		}
	}
	const uint64_t tot = wave_sum(acc);
	__shared__ uint64_t wave_tot[kWorkgroup / 64];
	if (res_cells && (threadIdx.x & 63u) == 0u) wave_tot[threadIdx.x >> 6] = tot;
	if (OP == 3 || res_cells) __syncthreads(); // every wave's bits are in the image / every wave's total is in LDS
	if (OP == 3) {
		if (sel_out.debug == 0) sel_write_out(sel_out, bitmap32 + (d.val_off >> 5), sel_p0, sel_p0 + g.n);
		// the odd 32-bit half of the bitmap's last 64-bit word belongs to no group
		if (edge_cells && gi == last_group && tail_word != ~0ull && threadIdx.x == 128u) bitmap32[tail_word] = 0u;
	}
	if (res_cells) { // the group's total: one party per group at the segment's result cell (a wave that has no edge word)
		if (threadIdx.x == 64u) {
			uint64_t all = 0;
#pragma unroll
			for (int i = 0; i < kWorkgroup / 64; i++) all += wave_tot[i];
			if (OP == 1 || OP == 3) {
				arrive_count(res_cells + 4u * (uint64_t)g.seg, all, g.seg_groups, result + g.seg);
			} else {
				arrive_sum(res_cells + 4u * (uint64_t)g.seg, all, g.seg_groups, result + g.seg, arrive_swapped);
			}
		}
	} else if ((threadIdx.x & 63) == 0 && tot != 0) { // wrapping sums commute: one atomic per wave and group
		atomicAdd(reinterpret_cast<unsigned long long *>(result + g.seg), (unsigned long long)tot);
	}
}

// ---------------------------------------------------------------------------------------------
// Persistent form of the decode kernel (A/B option).  A single tile's chain (tile table -> descriptor -> packed loads ->
// barrier -> decode [-> wave reduce -> atomic]) is serial and its fixed part is as long as the decode of a
// 16 KiB tile, so read-only scans leave HBM idle.  Here a workgroup owns a CONTIGUOUS run of tiles and keeps
// the next tile's packed bytes in flight with LDS-DMA (global_load_lds_dwordx4: global -> LDS, no VGPRs)
// into the other half of a double-buffered LDS image while it decodes the current half; one barrier per
// tile; aggregates are carried in registers across the tiles of a segment and flushed once per segment.
// (A first version prefetched into registers under `if (c < nchunks)`: hipcc put s_waitcnt vmcnt(0) behind
// every conditional load — cdna_hip_programming.md §5 item 4(c) — and it ran 0.6x; LDS-DMA has no
// destination register to merge, so nothing forces an early wait.)
// ---------------------------------------------------------------------------------------------
constexpr int kMaxChunks = kTileBytes / 16 + 2;

struct TileJob {
	const uint4 *src;  // first 16-byte chunk of the tile's packed bits
	uint64_t elem0;    // element index of the tile's first row in the value buffer
	uint64_t add;      // frame of reference to add back (0 if none)
	uint32_t seg, n, w, bit0, nchunks;
};

template <int TILE>
__device__ __forceinline__ TileJob make_job(const adac_segment_desc *__restrict__ descs,
                                            const TileRef *__restrict__ tiles, uint32_t t,
                                            const uint64_t *__restrict__ words) {
	const TileRef r = tiles[t];
	const adac_segment_desc d = descs[r.seg];
	TileJob j;
	j.seg = r.seg;
	const uint32_t left = d.count - r.first;
	j.n = left < (uint32_t)TILE ? left : (uint32_t)TILE;
	j.w = d.width;
	j.elem0 = d.val_off + r.first;
	j.add = effective_add(d);
	const uint64_t bitpos = (uint64_t)r.first * j.w;
	j.bit0 = (uint32_t)(bitpos & 127);
	j.nchunks = (j.bit0 + j.n * j.w + 127u) >> 7;
	j.src = reinterpret_cast<const uint4 *>(words + d.word_off) + (bitpos >> 7);
	return j;
}

using gptr_t = const __attribute__((address_space(1))) void *;
using lptr_t = __attribute__((address_space(3))) void *;

// Fire-and-forget copy of `nchunks` 16-byte chunks global -> LDS.  One wave instruction moves 1 KiB; its LDS
// destination is the wave-uniform base + lane*16 (M0), so each wave copies 64 consecutive chunks per round.
__device__ __forceinline__ void dma_chunks(const uint4 *__restrict__ src, uint32_t nchunks, uint4 *lds_buf) {
	const uint32_t lane = threadIdx.x & 63u;
	for (uint32_t base = threadIdx.x & ~63u; base < nchunks; base += kWorkgroup) {
		const uint32_t c = base + lane;
		if (c < nchunks) {
			__builtin_amdgcn_global_load_lds((gptr_t)(src + c), (lptr_t)(lds_buf + base), 16, 0, 0);
		}
	}
}

__device__ __forceinline__ void tile_range(uint32_t ntiles, uint32_t &lo, uint32_t &hi) {
	const uint32_t per = (ntiles + gridDim.x - 1) / gridDim.x;
	lo = blockIdx.x * per;
	hi = lo + per < ntiles ? lo + per : ntiles;
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_unpack_p(const adac_segment_desc *__restrict__ descs,
                                                         const TileRef *__restrict__ tiles, uint32_t ntiles,
                                                         const uint64_t *__restrict__ words, U *__restrict__ out) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr uint32_t K = 16 / sizeof(U);
	__shared__ uint4 lds[2][kMaxChunks];
	uint32_t t, hi;
	tile_range(ntiles, t, hi);
	if (t >= hi) return;
	TileJob cur = make_job<TILE>(descs, tiles, t, words);
	dma_chunks(cur.src, cur.nchunks, lds[0]);
	uint32_t buf = 0;
	for (;;) {
		__syncthreads(); // drains this wave's DMA (vmcnt) and publishes the image; also fences the other half
		const bool more = t + 1 < hi;
		TileJob nxt = cur;
		if (more) {
			nxt = make_job<TILE>(descs, tiles, t + 1, words);
			dma_chunks(nxt.src, nxt.nchunks, lds[buf ^ 1]); // lands while this tile is decoded
		}
		const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds[buf]);
		StoreSink<U> sink {out + cur.elem0, cur.n};
		const uint32_t align = (uint32_t)(cur.elem0 & (K - 1));
		decode_tile<U>(lds32, cur.bit0, cur.w, cur.add, cur.n, align, sink);
		if (!more) break;
		cur = nxt;
		buf ^= 1;
		t++;
	}
}

// ---------------------------------------------------------------------------------------------
// Raw value access shared by k_analyze and k_pack: chunks of K rows aligned to 16 bytes of the element index.
// ---------------------------------------------------------------------------------------------
template <typename U>
__device__ __forceinline__ void load_chunk(const U *__restrict__ src /* element elem0 */, int32_t base, uint32_t n,
                                           U *vals) {
	constexpr int K = 16 / (int)sizeof(U);
	if (base >= 0 && (uint32_t)(base + K) <= n) {
		const uint4 q = *reinterpret_cast<const uint4 *>(src + base);
		__builtin_memcpy(vals, &q, 16);
	} else {
#pragma unroll
		for (int j = 0; j < K; j++) {
			vals[j] = (uint32_t)(base + j) < n ? src[base + j] : (U)0;
		}
	}
}

// Validity bits of the K rows of one chunk.  A chunk starts at an element index that is a multiple of K and K
// divides 64, so its K bits live in ONE word of the DuckDB validity mask: one 8-byte load per chunk.
// Bit j of the result = row (chunk_elem + j) is valid.
__device__ __forceinline__ uint32_t chunk_validity(const uint64_t *__restrict__ validity, uint64_t chunk_elem) {
	if (validity == nullptr) return 0xffffffffu;
	return (uint32_t)(validity[chunk_elem >> 6] >> (chunk_elem & 63));
}

// ---------------------------------------------------------------------------------------------
// k_analyze — per-segment min/max.  Per-lane running min/max, wave64 xor-shuffle reduce, 4 partials through
// LDS, one atomic min + one atomic max per tile (order-independent, so deterministic).
// ---------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_analyze(const adac_segment_desc *__restrict__ descs,
                                                        const TileRef *__restrict__ tiles,
                                                        const U *__restrict__ vals,
                                                        const uint64_t *__restrict__ validity, int sign_extend,
                                                        uint64_t null_bits, int rule,
                                                        uint64_t *__restrict__ minmax) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr int K = 16 / (int)sizeof(U);
	using S = typename std::make_signed<U>::type;
	// The 1-, 2- and 4-byte types reduce in 32 bits, on the bit PATTERN of T: sign extension to 64 bits is monotone in
	// the unsigned order of the pattern (0x7f < 0x80 and 0x7f < 0xff...80), so min / max commute with the widening of
	// rule APPEND (succinct.cpp:286-287) and the result is widened once.  (In 64 bits the sixteen rows of a 1-byte
	// chunk took k_analyze<u8> to 132 VGPRs, three waves per SIMD.)
	using X = typename std::conditional<sizeof(U) == 8, uint64_t, uint32_t>::type;
	__shared__ X pmin[kWorkgroup / 64], pmax[kWorkgroup / 64];
	const TileCtx t = resolve_tile<TILE>(descs, tiles);
	const U *src = vals + t.elem0;
	const uint32_t align = (uint32_t)(t.elem0 & (K - 1));
	const X flip = rule == ADAC_RULE_ZONEMAP ? (X)null_bits : (X)0; // the zonemap's order-preserving bias (T's sign bit)
	X mn = ~(X)0, mx = 0; // "nothing seen": mn > mx
	if (t.n == (uint32_t)TILE && align == 0 && validity == nullptr) {
		// full, aligned, all-valid tile: the four 16-byte loads of a lane are issued back to back
		constexpr int ROUNDS = TILE / (kWorkgroup * K);
		uint4 q[ROUNDS];
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			q[r] = *reinterpret_cast<const uint4 *>(src + (r * kWorkgroup + threadIdx.x) * K);
		}
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			U v[K];
			__builtin_memcpy(v, &q[r], 16);
#pragma unroll
			for (int j = 0; j < K; j++) {
				const X x = (X)v[j] ^ flip;
				mn = x < mn ? x : mn;
				mx = x > mx ? x : mx;
			}
		}
	} else
	for (uint32_t c = threadIdx.x; c * K < t.n + align; c += kWorkgroup) {
		const int32_t base = (int32_t)(c * K) - (int32_t)align;
		U v[K];
		load_chunk<U>(src, base, t.n, v);
		const uint32_t vbits = chunk_validity(validity, (uint64_t)((int64_t)t.elem0 + base));
#pragma unroll
		for (int j = 0; j < K; j++) {
			if ((uint32_t)(base + j) >= t.n) continue;
			const bool valid = (vbits >> j) & 1u;
			X x;
			if (rule == ADAC_RULE_RECOMPACT) {
				// column_segment.cpp:392-399: every slot, zero-extended; NULL slots hold NullValue<T>
				x = valid ? (X)v[j] : (X)null_bits;
			} else {
				// NumericStatistics::Update<T> (numeric_statistics.hpp:54-67): typed min/max over the valid rows, in the
				// biased form bits(v) ^ signbit; succinct.cpp:286-287: uint64_t(sdata[i]), NULL rows do not take part
				if (!valid) continue;
				x = (X)v[j] ^ flip;
			}
			mn = x < mn ? x : mn;
			mx = x > mx ? x : mx;
		}
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const X omn = __shfl_xor(mn, off, 64), omx = __shfl_xor(mx, off, 64);
		mn = omn < mn ? omn : mn;
		mx = omx > mx ? omx : mx;
	}
	if ((threadIdx.x & 63) == 0) {
		pmin[threadIdx.x >> 6] = mn;
		pmax[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
#pragma unroll
		for (int i = 1; i < kWorkgroup / 64; i++) {
			mn = pmin[i] < mn ? pmin[i] : mn;
			mx = pmax[i] > mx ? pmax[i] : mx;
		}
		if (mn <= mx) { // something was seen (otherwise the atomics would be no-ops on the initial ~0 / 0 anyway)
			uint64_t mn64 = mn, mx64 = mx;
			if (sizeof(U) < 8 && rule == ADAC_RULE_APPEND && sign_extend) {
				mn64 = (uint64_t)(int64_t)(S)(U)mn;
				mx64 = (uint64_t)(int64_t)(S)(U)mx;
			}
			atomicMin(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)t.seg), (unsigned long long)mn64);
			atomicMax(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)t.seg + 1), (unsigned long long)mx64);
		}
	}
}

__global__ void k_minmax_init(uint64_t *__restrict__ minmax, uint64_t nseg) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < nseg) {
		minmax[2 * i] = ~0ull; // ColumnSegment ctor: min_factor(UINT64_MAX), max_factor(0)  column_segment.cpp:94
		minmax[2 * i + 1] = 0;
	}
}

// ---------------------------------------------------------------------------------------------
// k_plan — widths, flags and arena offsets for every segment, on the device (one workgroup, chunked scan).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t width_for(uint64_t mn, uint64_t mx, int rule, int pad) {
	uint32_t w;
	if (rule == ADAC_RULE_APPEND) {
		w = hi_bit(mx - mn) + 1; // column_segment.cpp:351-354 (wrapping u64 arithmetic)
	} else {
		if (mx != 0 && mn != ~0ull && mx > mn) mx -= mn; // column_segment.cpp:404-407
		w = hi_bit(mx) + 1;
	}
	if (pad) w = (w + 7u) & ~7u; // column_segment.cpp:356-359
	return w;
}

constexpr int kPlanThreads = 1024;

__global__ __launch_bounds__(kPlanThreads) void k_plan(adac_segment_desc *__restrict__ descs,
                                                       const uint64_t *__restrict__ minmax, uint64_t nseg,
                                                       uint32_t type_bits, int rule, int pad) {
	// exclusive scan of the segments' arena footprints: wave64 shuffle scan, the 16 wave totals through LDS,
	// a running carry across chunks of 1024 segments — two barriers per chunk
	constexpr int kWaves = kPlanThreads / 64;
	__shared__ uint64_t wave_total[kWaves];
	__shared__ uint64_t carry_s;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry_s = 0;
	__syncthreads();
	for (uint64_t base = 0; base < nseg; base += kPlanThreads) {
		const uint64_t s = base + threadIdx.x;
		uint64_t fp = 0;
		uint32_t w = type_bits;
		uint8_t flags = 0;
		uint64_t mn = ~0ull;
		if (s < nseg) {
			mn = minmax[2 * s];
			const uint64_t mx = minmax[2 * s + 1];
			const uint32_t cand = width_for(mn, mx, rule, pad);
			if (type_bits > cand) { // `if (old_width > min_width)` column_segment.cpp:363,420
				w = cand;
				flags = ADAC_SEG_PACKED;
				// Sentinel collision (reference defect 7): min == max == UINT64_MAX means every valid row is
				// all-ones (-1 for the INT types), but UINT64_MAX is also "no min": the reference then packs
				// WITHOUT subtracting (column_segment.cpp:371-373) and scans without adding (succinct.cpp:138-140),
				// returning 2^w - 1 instead of -1.  Storing min = UINT64_MAX - (2^w - 1) keeps the packed bits
				// identical (x - min' = 2^w - 1 = x mod 2^w) and makes field + min' the original value.
				if (mn == ~0ull && mx == ~0ull) mn = ~0ull - mask64(w);
			}
			const uint64_t bits = (uint64_t)descs[s].count * w;
			fp = (((bits + 64) >> 6) + 15) & ~15ull; // SDSL allocation, rounded to 128 B
		}
		uint64_t incl = fp;
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const uint64_t up = __shfl_up(incl, off, 64);
			if (lane >= (uint32_t)off) incl += up;
		}
		if (lane == 63) wave_total[wave] = incl;
		__syncthreads();
		uint64_t before = carry_s;
		for (uint32_t i = 0; i < wave; i++) before += wave_total[i];
		if (s < nseg) {
			descs[s].word_off = before + incl - fp;
			descs[s].min = mn;
			descs[s].width = (uint8_t)w;
			descs[s].flags = flags;
			descs[s].reserved = 0;
		}
		__syncthreads();
		if (threadIdx.x == kPlanThreads - 1) carry_s = before + incl;
		__syncthreads();
	}
}

// ---------------------------------------------------------------------------------------------
// k_pack — bit-pack one tile.  Phase 1: (x - min) mod 2^w of every row into LDS (as U: w <= 8*sizeof(U)).
// Phase 2: each lane OWNS whole output uint64 words and gathers the rows that overlap its word — no atomics,
// no read-modify-write, coalesced 8-byte stores; tail bits of the last word come out zero by construction.
// Tiles start at multiples of TILE (a multiple of 64 rows), so a tile's bits start on a word boundary.
// ---------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_pack(const adac_segment_desc *__restrict__ descs,
                                                     const TileRef *__restrict__ tiles, const U *__restrict__ vals,
                                                     const uint64_t *__restrict__ validity, uint64_t null_bits,
                                                     uint64_t *__restrict__ words) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr int K = 16 / (int)sizeof(U);
	__shared__ __attribute__((aligned(16))) U delta[TILE];
	const TileCtx t = resolve_tile<TILE>(descs, tiles);
	const uint32_t w = t.d.width;
	const bool packed = (t.d.flags & ADAC_SEG_PACKED) != 0;
	const uint64_t sub = (packed && t.d.min != ADAC_NO_MIN) ? t.d.min : 0ull; // column_segment.cpp:371-373
	const U wmask = (U)mask64(w);
	const U *src = vals + t.elem0;
	const uint32_t align = (uint32_t)(t.elem0 & (K - 1));
	if (t.n == (uint32_t)TILE && align == 0 && validity == nullptr) {
		constexpr int ROUNDS = TILE / (kWorkgroup * K);
		uint4 q[ROUNDS];
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			q[r] = *reinterpret_cast<const uint4 *>(src + (r * kWorkgroup + threadIdx.x) * K);
		}
#pragma unroll
		for (int r = 0; r < ROUNDS; r++) {
			U v[K];
			__builtin_memcpy(v, &q[r], 16);
#pragma unroll
			for (int j = 0; j < K; j++) {
				v[j] = (U)(v[j] - (U)sub) & wmask;
			}
			uint4 o;
			__builtin_memcpy(&o, v, 16);
			*reinterpret_cast<uint4 *>(delta + (r * kWorkgroup + threadIdx.x) * K) = o; // ds_write_b128
		}
	} else
	for (uint32_t c = threadIdx.x; c * K < t.n + align; c += kWorkgroup) {
		const int32_t base = (int32_t)(c * K) - (int32_t)align;
		U v[K];
		load_chunk<U>(src, base, t.n, v);
		const uint32_t vbits = chunk_validity(validity, (uint64_t)((int64_t)t.elem0 + base));
#pragma unroll
		for (int j = 0; j < K; j++) {
			const uint32_t row = (uint32_t)(base + j);
			if (row >= t.n) continue;
			const bool valid = (vbits >> j) & 1u;
			const U x = valid ? v[j] : (U)null_bits;
			delta[row] = (U)(x - (U)sub) & wmask; // low w bits of (x - min): w <= 8*sizeof(U)
		}
	}
	__syncthreads();
	const uint64_t word0 = ((uint64_t)t.first * w) >> 6;
	const uint32_t nwords = (t.n * w + 63u) >> 6;
	uint64_t *__restrict__ dst = words + t.d.word_off + word0;
	for (uint32_t q = threadIdx.x; q < nwords; q += kWorkgroup) {
		const uint32_t bitlo = q << 6;
		uint32_t i = bitlo / w;
		uint32_t last = (bitlo + 63u) / w;
		last = last < t.n ? last : t.n - 1;
		uint64_t acc = 0;
		for (; i <= last; i++) {
			const uint64_t v = (uint64_t)delta[i];
			const int32_t pos = (int32_t)(i * w) - (int32_t)bitlo;
			acc |= pos >= 0 ? (v << pos) : (v >> (-pos));
		}
		__builtin_nontemporal_store(acc, &dst[q]); // written once, read much later
	}
}

// ---------------------------------------------------------------------------------------------
// Re-compaction packed -> packed (SURVEY.md §8b `adac_repack`, §8d "old_w -> new_w: n (old_w + new_w) / 8 bytes").
// The reference only ever compacts from full-width slots (BitCompressFromSuccinct, column_segment.cpp:348-383);
// these two kernels are the same two passes with a PACKED source: the tile's values come from decoding the
// source segment's staged bits instead of from a raw array, everything downstream (min/max rules, width, the
// word-owner pack) is shared with k_analyze / k_pack.  Source and destination layouts have the same counts.
// ---------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_analyze_packed(const adac_segment_desc *__restrict__ src_descs,
                                                               const TileRef *__restrict__ tiles,
                                                               const uint64_t *__restrict__ src_words,
                                                               const uint64_t *__restrict__ validity, int sign_extend,
                                                               uint64_t null_bits, int rule,
                                                               uint64_t *__restrict__ minmax) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	using S = typename std::make_signed<U>::type;
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ uint64_t pmin[kWorkgroup / 64], pmax[kWorkgroup / 64];
	const TileCtx t = resolve_tile<TILE>(src_descs, tiles);
	const uint32_t bit0 = stage_packed(src_words + t.d.word_off, t.first, t.n, t.d.width, lds);
	__syncthreads();
	uint64_t mn = ~0ull, mx = 0;
	auto sink = [&](int32_t base, const U *v, bool full) { // align 0: base >= 0
		constexpr int K = 16 / (int)sizeof(U);
		const uint32_t rows_here = full || t.n - (uint32_t)base >= (uint32_t)K ? (uint32_t)K : t.n - (uint32_t)base;
		const uint32_t vbits = validity ? validity_window(validity, t.elem0 + (uint32_t)base, rows_here) : 0xffffffffu;
#pragma unroll
		for (int j = 0; j < K; j++) {
			if (!full && (uint32_t)(base + j) >= t.n) continue;
			const bool valid = (vbits >> j) & 1u;
			uint64_t x;
			if (rule == ADAC_RULE_APPEND) { // succinct.cpp:286-287: NULL rows do not take part
				if (!valid) continue;
				x = sign_extend ? (uint64_t)(int64_t)(S)v[j] : (uint64_t)v[j];
			} else { // column_segment.cpp:392-399: every slot, zero-extended; NULL slots hold NullValue<T>
				x = valid ? (uint64_t)v[j] : null_bits;
			}
			mn = x < mn ? x : mn;
			mx = x > mx ? x : mx;
		}
	};
	decode_tile<U>(reinterpret_cast<const uint32_t *>(lds), bit0, t.d.width, effective_add(t.d), t.n, 0u, sink);
	mn = wave_min(mn);
	mx = wave_max(mx);
	if ((threadIdx.x & 63) == 0) {
		pmin[threadIdx.x >> 6] = mn;
		pmax[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
#pragma unroll
		for (int i = 1; i < kWorkgroup / 64; i++) {
			mn = pmin[i] < mn ? pmin[i] : mn;
			mx = pmax[i] > mx ? pmax[i] : mx;
		}
		atomicMin(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)t.seg), (unsigned long long)mn);
		atomicMax(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)t.seg + 1), (unsigned long long)mx);
	}
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_repack(const adac_segment_desc *__restrict__ src_descs,
                                                       const adac_segment_desc *__restrict__ dst_descs,
                                                       const TileRef *__restrict__ tiles,
                                                       const uint64_t *__restrict__ src_words,
                                                       const uint64_t *__restrict__ validity, uint64_t null_bits,
                                                       uint64_t *__restrict__ dst_words) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ __attribute__((aligned(16))) U delta[TILE];
	const TileCtx t = resolve_tile<TILE>(src_descs, tiles);
	const adac_segment_desc dd = dst_descs[t.seg];
	const uint32_t w = dd.width;
	const bool packed = (dd.flags & ADAC_SEG_PACKED) != 0;
	const U sub = (U)((packed && dd.min != ADAC_NO_MIN) ? dd.min : 0ull); // column_segment.cpp:371-373
	const U wmask = (U)mask64(w);
	const uint32_t bit0 = stage_packed(src_words + t.d.word_off, t.first, t.n, t.d.width, lds);
	__syncthreads();
	auto sink = [&](int32_t base, const U *v, bool full) {
		constexpr int K = 16 / (int)sizeof(U);
		const uint32_t rows_here = full || t.n - (uint32_t)base >= (uint32_t)K ? (uint32_t)K : t.n - (uint32_t)base;
		const uint32_t vbits = validity ? validity_window(validity, t.elem0 + (uint32_t)base, rows_here) : 0xffffffffu;
		if (full) {
			U o[K];
#pragma unroll
			for (int j = 0; j < K; j++) {
				const U x = ((vbits >> j) & 1u) ? v[j] : (U)null_bits;
				o[j] = (U)(x - sub) & wmask;
			}
			uint4 q;
			__builtin_memcpy(&q, o, 16);
			*reinterpret_cast<uint4 *>(delta + base) = q; // base is a multiple of K: ds_write_b128
		} else {
#pragma unroll
			for (int j = 0; j < K; j++) {
				if ((uint32_t)(base + j) >= t.n) continue;
				const U x = ((vbits >> j) & 1u) ? v[j] : (U)null_bits;
				delta[base + j] = (U)(x - sub) & wmask;
			}
		}
	};
	decode_tile<U>(reinterpret_cast<const uint32_t *>(lds), bit0, t.d.width, effective_add(t.d), t.n, 0u, sink);
	__syncthreads();
	const uint64_t word0 = ((uint64_t)t.first * w) >> 6;
	const uint32_t nwords = (t.n * w + 63u) >> 6;
	uint64_t *__restrict__ dst = dst_words + dd.word_off + word0;
	for (uint32_t q = threadIdx.x; q < nwords; q += kWorkgroup) { // word-owner gather, as in k_pack
		const uint32_t bitlo = q << 6;
		uint32_t i = bitlo / w;
		uint32_t last = (bitlo + 63u) / w;
		last = last < t.n ? last : t.n - 1;
		uint64_t acc = 0;
		for (; i <= last; i++) {
			const uint64_t v = (uint64_t)delta[i];
			const int32_t pos = (int32_t)(i * w) - (int32_t)bitlo;
			acc |= pos >= 0 ? (v << pos) : (v >> (-pos));
		}
		__builtin_nontemporal_store(acc, &dst[q]); // written once, read much later
	}
}

// ---------------------------------------------------------------------------------------------
// Grouped re-compaction (the default form).  One tile per workgroup moves only n (old_w + new_w) / 8 bytes — 4 KiB at
// 8 -> 8 bits — behind a full HBM round trip and two barriers, and the 16 KiB row image allows four workgroups per
// CU: ~8 KiB in flight per CU, i.e. ~1.2 TB/s at narrow widths whatever the inner loops do (measured, DESIGN.md
// §2).  Here a workgroup owns a ScanGroup of the source layout (several tiles of one segment) and walks it in stages
// of as many whole tiles as fit 16 KiB of PACKED bytes on both sides, so the bytes in flight per workgroup no longer
// shrink with the width; and the row image is gone: a lane combines the K rows of its chunk into one bit string of
// K * new_w <= 128 bits in registers and ORs it into a zeroed LDS image of the OUTPUT words (ds_or_b64, one to three
// per chunk), which the workgroup then copies out with 16-byte stores.  Adjacent lanes touch adjacent words, so
// there is no strided LDS access left to conflict.
// ---------------------------------------------------------------------------------------------
// K masked fields of `w` bits (w <= 8 * sizeof(U)), row order -> the K * w-bit string {lo, hi}
template <typename U>
__device__ __forceinline__ void concat_fields(const U *f, uint32_t w, uint64_t &lo, uint64_t &hi) {
	uint64_t a, b;
	uint32_t s; // bits of a and of b
	if (sizeof(U) == 8) {
		a = (uint64_t)f[0];
		b = (uint64_t)f[1];
		s = w;
	} else {
		uint32_t q[4];
		uint32_t qw; // bits of each q
		if (sizeof(U) == 4) {
#pragma unroll
			for (int i = 0; i < 4; i++) q[i] = (uint32_t)f[i];
			qw = w;
		} else if (sizeof(U) == 2) {
#pragma unroll
			for (int i = 0; i < 4; i++) q[i] = (uint32_t)f[2 * i] | ((uint32_t)f[2 * i + 1] << w);
			qw = 2 * w;
		} else {
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const uint32_t p0 = (uint32_t)f[4 * i] | ((uint32_t)f[4 * i + 1] << w);
				const uint32_t p1 = (uint32_t)f[4 * i + 2] | ((uint32_t)f[4 * i + 3] << w);
				q[i] = p0 | (p1 << (2 * w));
			}
			qw = 4 * w;
		}
		a = (uint64_t)q[0] | ((uint64_t)q[1] << qw); // qw <= 32
		b = (uint64_t)q[2] | ((uint64_t)q[3] << qw);
		s = 2 * qw;
	}
	lo = s >= 64 ? a : (a | (b << s));
	hi = s >= 64 ? b : (b >> (64 - s)); // s >= 1
}

// OR the string {lo, hi} into the zeroed 64-bit word image at bit position p (ds_or_b64; words that get no bit are
// skipped, so narrow strings cost one atomic)
__device__ __forceinline__ void image_or(unsigned long long *img, uint32_t p, uint64_t lo, uint64_t hi) {
	const uint32_t idx = p >> 6, sh = p & 63u;
	const uint64_t o0 = lo << sh;
	const uint64_t o1 = sh ? ((lo >> (64 - sh)) | (hi << sh)) : hi;
	const uint64_t o2 = sh ? (hi >> (64 - sh)) : 0ull;
	if (o0) atomicOr(&img[idx], (unsigned long long)o0);
	if (o1) atomicOr(&img[idx + 1], (unsigned long long)o1);
	if (o2) atomicOr(&img[idx + 2], (unsigned long long)o2);
}

// ---- the same emission in 32-bit arithmetic, for destination widths <= 32 (every type; for the 8-byte types this is
// the common case: a frame-of-reference width above 32 bits is rare).  64-bit shifts and the per-word "is it zero"
// branches made the generic form above cost ~100 instructions per chunk — the pack phase, not HBM, bounded the
// kernels that use it.  Here the K fields of a chunk (low dwords only: just the low w bits of x - min matter) are
// folded to four sub-strings of wq = (K / 4) * w <= 32 bits with v_lshl_or_b32, those to a string of <= 128 bits
// {s0..s3} with uniform funnel shifts, and the string is placed at its lane-dependent bit position with
// v_alignbit_b32 and a fixed (wave-uniform) number of ds_or_b32.
struct Str128 {
	uint32_t s0, s1, s2, s3;
};

// (hi:lo) << t for a UNIFORM t in 0..32, as {d0, d1, d2}
__device__ __forceinline__ void shl64_u(uint32_t lo, uint32_t hi, uint32_t t, uint32_t &d0, uint32_t &d1, uint32_t &d2) {
	if (t == 0u) {
		d0 = lo, d1 = hi, d2 = 0u;
	} else if (t == 32u) {
		d0 = 0u, d1 = lo, d2 = hi;
	} else {
		d0 = lo << t;
		d1 = __builtin_amdgcn_alignbit(hi, lo, 32u - t);
		d2 = hi >> (32u - t);
	}
}

// four sub-strings of wq bits each (uniform wq in 1..32) -> their 4 * wq-bit concatenation
__device__ __forceinline__ Str128 concat4(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t wq) {
	// pairs (a, b) and (c, d): 2 * wq <= 64 bits each
	uint32_t p0, p1, q0, q1, x;
	shl64_u(b, 0u, wq, p0, p1, x);
	p0 |= a;
	shl64_u(d, 0u, wq, q0, q1, x);
	q0 |= c;
	// (q1:q0) << 2 * wq, 2 * wq in 2..64
	Str128 r;
	const uint32_t t = 2u * wq;
	if (t <= 32u) {
		uint32_t d0, d1, d2;
		shl64_u(q0, q1, t, d0, d1, d2);
		r.s0 = p0 | d0, r.s1 = p1 | d1, r.s2 = d2, r.s3 = 0u;
	} else {
		uint32_t d0, d1, d2;
		shl64_u(q0, q1, t - 32u, d0, d1, d2);
		r.s0 = p0, r.s1 = p1 | d0, r.s2 = d1, r.s3 = d2;
	}
	return r;
}

// the chunk's K values (x[j] = value or NullValue, already chosen) -> (x - sub) & mask fields -> one string.
// w <= 32 and uniform.  sub32 / mask32: low dwords of the frame of reference and of the width mask.
template <typename U>
__device__ __forceinline__ Str128 chunk_string32(const U *x, uint32_t sub32, uint32_t mask32, uint32_t w) {
	constexpr int K = 16 / (int)sizeof(U);
	uint32_t f[K];
#pragma unroll
	for (int j = 0; j < K; j++) f[j] = ((uint32_t)x[j] - sub32) & mask32;
	if (K == 2) {
		uint32_t d0, d1, d2;
		shl64_u(f[1], 0u, w, d0, d1, d2);
		return Str128 {d0 | f[0], d1, 0u, 0u};
	}
	uint32_t q[4];
	if (K == 4) {
#pragma unroll
		for (int i = 0; i < 4; i++) q[i] = f[i];
	} else if (K == 8) { // w <= 16: a pair fits a dword
#pragma unroll
		for (int i = 0; i < 4; i++) q[i] = f[2 * i] | (f[2 * i + 1] << w);
	} else { // K == 16, w <= 8
#pragma unroll
		for (int i = 0; i < 4; i++) {
			q[i] = f[4 * i] | (f[4 * i + 1] << w) | (f[4 * i + 2] << (2u * w)) | (f[4 * i + 3] << (3u * w));
		}
	}
	return concat4(q[0], q[1], q[2], q[3], (uint32_t)(K / 4) * w);
}

// OR a string of `nbits` bits (uniform, <= 128) into the zeroed 32-bit word image at bit position p (per lane)
__device__ __forceinline__ void image_or32(uint32_t *img32, uint32_t p, const Str128 &s, uint32_t nbits) {
	const uint32_t idx = p >> 5, sh = p & 31u, rs = 32u - sh;
	const bool z = sh == 0u; // v_alignbit takes its shift mod 32: a shift of 32 must be patched
	atomicOr(&img32[idx], s.s0 << sh);
	if (nbits > 1u) { // the string may reach the next word (nbits + sh > 32 for some lane)
		atomicOr(&img32[idx + 1], z ? s.s1 : __builtin_amdgcn_alignbit(s.s1, s.s0, rs));
	}
	if (nbits > 33u) atomicOr(&img32[idx + 2], z ? s.s2 : __builtin_amdgcn_alignbit(s.s2, s.s1, rs));
	if (nbits > 65u) atomicOr(&img32[idx + 3], z ? s.s3 : __builtin_amdgcn_alignbit(s.s3, s.s2, rs));
	if (nbits > 97u) atomicOr(&img32[idx + 4], z ? 0u : (s.s3 >> rs));
}

// whole tiles of a stage: both packed sides within 16 KiB
template <typename U>
__device__ __forceinline__ uint32_t stage_tiles(uint32_t w_a, uint32_t w_b) {
	constexpr uint32_t TILE = kTileBytes / sizeof(U);
	const uint32_t w = w_a > w_b ? w_a : w_b;
	const uint32_t fit = (8u * kTileBytes) / (TILE * w);
	return fit < 1u ? 1u : fit;
}

// ---------------------------------------------------------------------------------------------
// Width-templated source walk of the re-compaction (4 <= old_w <= 32, new_w <= 32, no NULL mask): the fused scans'
// chunk walk — a lane owns whole 16-byte chunks of the SOURCE stream and finds every field at a compile-time
// position (one v_bfe each, no LDS staging, no per-row address arithmetic) — feeding a streaming emission: the fields
// of a chunk are consecutive rows, so their re-based values form ONE bit string of have * new_w bits; it is built in a
// 64-bit accumulator at wave-uniform offsets (scalar bookkeeping), and every completed dword is shifted to the
// lane's own bit position with one v_alignbit and ORed into the zeroed LDS image of the stage's output words.
// The generic form read every row back from an LDS image (address arithmetic + two ds_read + funnel + mask + add
// per row: 26 vector instructions per row at 8 bits, 75 of 87 us) — this one spends ~8.
// ---------------------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ void repack_run_w(const uint4 *__restrict__ seg16, uint32_t r0, uint32_t r1, uint32_t count,
                                             uint32_t delta, uint32_t w_new, uint32_t *img32) {
	constexpr int MAXV = (128 + W - 1) / W;
	const uint32_t mask_new = w_new >= 32u ? 0xffffffffu : ((1u << w_new) - 1u);
	const uint32_t c0 = (uint32_t)(((uint64_t)r0 * W) >> 7); // r0 is a multiple of 128 rows
	const uint32_t c1 = (uint32_t)(((uint64_t)r1 * W + 127) >> 7);
	const uint32_t clast = (uint32_t)(((uint64_t)count * W + 127) >> 7) - 1; // last chunk holding data bits
	uint32_t L = c0 + threadIdx.x;
	if (L >= c1) return;
	const uint32_t Lc = L < clast ? L : clast;
	uint4 q = seg16[Lc];
	uint32_t e = reinterpret_cast<const uint32_t *>(seg16 + (Lc < clast ? Lc + 1 : clast))[0];
	for (; L < c1; L += kWorkgroup) {
		const uint32_t Lp = L + kWorkgroup < clast ? L + kWorkgroup : clast;
		const uint4 qn = seg16[Lp];
		const uint32_t en = reinterpret_cast<const uint32_t *>(seg16 + (Lp < clast ? Lp + 1 : clast))[0];
		const uint32_t i0 = (128u * L + (W - 1)) / W; // first row starting in this chunk
		const uint32_t o0 = i0 * W - 128u * L;        // its bit offset, < W <= 32
		uint32_t nrm[5];
		nrm[0] = __builtin_amdgcn_alignbit(q.y, q.x, o0);
		nrm[1] = __builtin_amdgcn_alignbit(q.z, q.y, o0);
		nrm[2] = __builtin_amdgcn_alignbit(q.w, q.z, o0);
		nrm[3] = __builtin_amdgcn_alignbit(e, q.w, o0);
		nrm[4] = e >> o0;
		q = qn;
		e = en;
		const uint32_t starting = (128u - o0 + (W - 1)) / W; // rows starting in the chunk: MAXV-1 or MAXV
		const uint32_t lim = r1 > i0 ? r1 - i0 : 0u;
		const uint32_t have = starting < lim ? starting : lim;
		if (have == 0u) continue;
		// the string starts at bit p of the stage image; dwords leave through a funnel by p's offset in its dword
		const uint32_t p = (i0 - r0) * w_new;
		uint32_t *out = img32 + (p >> 5);
		const uint32_t sh = p & 31u, rs = 32u - sh;
		const bool z = sh == 0u;
		uint64_t acc = 0;
		uint32_t prev = 0;
		uint32_t t = 0, nd = 0; // wave-uniform: bits in acc, dwords emitted
#pragma unroll
		for (int j = 0; j < MAXV; j++) {
			const uint32_t g = (uint32_t)j < have ? ((field_of<W>(nrm, j) + delta) & mask_new) : 0u;
			acc |= (uint64_t)g << t;
			t += w_new;
			if (t >= 32u) { // uniform
				const uint32_t cur = (uint32_t)acc;
				atomicOr(&out[nd], z ? cur : __builtin_amdgcn_alignbit(cur, prev, rs));
				prev = cur;
				acc >>= 32;
				t -= 32u;
				nd++;
			}
		}
		const uint32_t cur = (uint32_t)acc; // the last, partial dword (zero when t == 0) and what the funnel still holds
		atomicOr(&out[nd], z ? cur : __builtin_amdgcn_alignbit(cur, prev, rs));
		if (!z) atomicOr(&out[nd + 1], cur >> rs);
	}
}

// The same source walk for the min / max pass of a re-compaction: the extrema of the rows' T-bit patterns
// x = (field + old min) mod 2^(8 sizeof T), in unsigned order (which IS the order of the sign- or zero-extended 64-bit
// values both rules compare: negative values extend to the largest numbers).  Types of up to 32 bits track x itself;
// the 8-byte types track the field (a packed 8-byte segment never wraps mod 2^64: a mixed-sign range needs 64 bits and
// stays unpacked) and add the old min once at the end.
template <int W, int TBITS>
__device__ __forceinline__ void analyze_run_w(const uint4 *__restrict__ seg16, uint32_t r0, uint32_t r1, uint32_t count,
                                              uint32_t add32, uint32_t &mn32, uint32_t &mx32) {
	constexpr int MAXV = (128 + W - 1) / W;
	constexpr uint32_t tmask = TBITS >= 32 ? 0xffffffffu : ((1u << (TBITS & 31)) - 1u);
	const uint32_t c0 = (uint32_t)(((uint64_t)r0 * W) >> 7);
	const uint32_t c1 = (uint32_t)(((uint64_t)r1 * W + 127) >> 7);
	const uint32_t clast = (uint32_t)(((uint64_t)count * W + 127) >> 7) - 1;
	uint32_t L = c0 + threadIdx.x;
	if (L >= c1) return;
	const uint32_t Lc = L < clast ? L : clast;
	uint4 q = seg16[Lc];
	uint32_t e = reinterpret_cast<const uint32_t *>(seg16 + (Lc < clast ? Lc + 1 : clast))[0];
	for (; L < c1; L += kWorkgroup) {
		const uint32_t Lp = L + kWorkgroup < clast ? L + kWorkgroup : clast;
		const uint4 qn = seg16[Lp];
		const uint32_t en = reinterpret_cast<const uint32_t *>(seg16 + (Lp < clast ? Lp + 1 : clast))[0];
		const uint32_t i0 = (128u * L + (W - 1)) / W;
		const uint32_t o0 = i0 * W - 128u * L;
		uint32_t nrm[5];
		nrm[0] = __builtin_amdgcn_alignbit(q.y, q.x, o0);
		nrm[1] = __builtin_amdgcn_alignbit(q.z, q.y, o0);
		nrm[2] = __builtin_amdgcn_alignbit(q.w, q.z, o0);
		nrm[3] = __builtin_amdgcn_alignbit(e, q.w, o0);
		nrm[4] = e >> o0;
		q = qn;
		e = en;
		const uint32_t starting = (128u - o0 + (W - 1)) / W;
		const uint32_t lim = r1 > i0 ? r1 - i0 : 0u;
		const uint32_t have = starting < lim ? starting : lim;
#pragma unroll
		for (int j = 0; j < MAXV; j++) {
			if ((uint32_t)j < have) {
				const uint32_t x = TBITS == 64 ? field_of<W>(nrm, j) : ((field_of<W>(nrm, j) + add32) & tmask);
				mn32 = x < mn32 ? x : mn32;
				mx32 = x > mx32 ? x : mx32;
			}
		}
	}
}

template <typename U>
__device__ __forceinline__ void analyze_run_dispatch(uint32_t w_old, const uint4 *__restrict__ seg16, uint32_t r0,
                                                     uint32_t r1, uint32_t count, uint32_t add32, uint32_t &mn32,
                                                     uint32_t &mx32) {
	switch (w_old) {
#define ADAC_W(N) case N: analyze_run_w<N, 8 * (int)sizeof(U)>(seg16, r0, r1, count, add32, mn32, mx32); break;
		ADAC_W(4) ADAC_W(5) ADAC_W(6) ADAC_W(7) ADAC_W(8) ADAC_W(9) ADAC_W(10) ADAC_W(11) ADAC_W(12) ADAC_W(13)
		ADAC_W(14) ADAC_W(15) ADAC_W(16) ADAC_W(17) ADAC_W(18) ADAC_W(19) ADAC_W(20) ADAC_W(21) ADAC_W(22)
		ADAC_W(23) ADAC_W(24) ADAC_W(25) ADAC_W(26) ADAC_W(27) ADAC_W(28) ADAC_W(29) ADAC_W(30) ADAC_W(31) ADAC_W(32)
#undef ADAC_W
	default: break;
	}
}

template <typename U>
__device__ __forceinline__ void repack_run_dispatch(uint32_t w_old, const uint4 *__restrict__ seg16, uint32_t r0,
                                                    uint32_t r1, uint32_t count, uint32_t delta, uint32_t w_new,
                                                    uint32_t *img32) {
	switch (w_old) {
#define ADAC_W(N) case N: repack_run_w<N>(seg16, r0, r1, count, delta, w_new, img32); break;
		ADAC_W(4) ADAC_W(5) ADAC_W(6) ADAC_W(7) ADAC_W(8) ADAC_W(9) ADAC_W(10) ADAC_W(11) ADAC_W(12) ADAC_W(13)
		ADAC_W(14) ADAC_W(15) ADAC_W(16) ADAC_W(17) ADAC_W(18) ADAC_W(19) ADAC_W(20) ADAC_W(21) ADAC_W(22)
		ADAC_W(23) ADAC_W(24) ADAC_W(25) ADAC_W(26) ADAC_W(27) ADAC_W(28) ADAC_W(29) ADAC_W(30) ADAC_W(31) ADAC_W(32)
#undef ADAC_W
	default: break;
	}
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_repack_g(const ScanGroup *__restrict__ src_groups,
                                                         const adac_segment_desc *__restrict__ dst_descs,
                                                         const uint64_t *__restrict__ src_words,
                                                         const uint64_t *__restrict__ validity, uint64_t null_bits,
                                                         int templated, uint64_t *__restrict__ dst_words) {
	constexpr int K = 16 / (int)sizeof(U);
	constexpr uint32_t PER_ROUND = kWorkgroup * K; // rows one decode round of the workgroup covers
	constexpr uint32_t kImgWords = kTileBytes / 8 + 4;
	// the generic path's staged source: 8 KiB, so that stage + image stay under 25 KiB and six workgroups fit a CU
	// (the register path uses only the image, and is bound by the bytes a CU keeps in flight)
	constexpr uint32_t kStageBytes = kTileBytes / 2;
	__shared__ uint4 lds[kStageBytes / 16 + 2];
	__shared__ __attribute__((aligned(16))) unsigned long long img[kImgWords];
	const ScanGroup g = load_scan_group(src_groups, blockIdx.x);
	const adac_segment_desc &sd = g.d;
	const adac_segment_desc dd = dst_descs[g.seg];
	const uint32_t w_old = sd.width, w = dd.width;
	const bool packed = (dd.flags & ADAC_SEG_PACKED) != 0;
	const U sub = (U)((packed && dd.min != ADAC_NO_MIN) ? dd.min : 0ull); // column_segment.cpp:371-373
	const U wmask = (U)mask64(w);
	const uint64_t add = effective_add(sd);
	for (uint32_t i = threadIdx.x; i < kImgWords; i += kWorkgroup) img[i] = 0ull;
	if (templated && !validity && w_old >= 4u && w_old <= 32u && w <= 32u && (uint64_t)sd.count * w_old < (1ull << 31)) {
		// register path (repack_run_w): only the low 32 bits of (field + old min - new min) matter at new_w <= 32,
		// and they are right for every T (the value is formed mod 2^(8 sizeof T) >= 2^new_w)
		const uint32_t delta = (uint32_t)add - (uint32_t)sub;
		const uint4 *seg16 = reinterpret_cast<const uint4 *>(src_words + sd.word_off);
		uint32_t stage_rows = ((8u * kTileBytes) / w) & ~127u; // output bits of a stage fit the 16 KiB image
		__syncthreads();
		for (uint32_t done = 0; done < g.n; done += stage_rows) {
			const uint32_t r0 = g.first + done; // a multiple of 128 rows: chunk- and word-aligned on both sides
			const uint32_t n = g.n - done < stage_rows ? g.n - done : stage_rows;
			repack_run_dispatch<U>(w_old, seg16, r0, r0 + n, sd.count, delta, w, reinterpret_cast<uint32_t *>(img));
			// LDS-only barriers: __syncthreads() also drains vmcnt, i.e. waits for every lane's last (unused) prefetch
			// and for the previous stage's stores
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
			const uint32_t nwords = (n * w + 63u) >> 6;
			unsigned long long *__restrict__ dst =
			    reinterpret_cast<unsigned long long *>(dst_words) + dd.word_off + (((uint64_t)r0 * w) >> 6);
			for (uint32_t q = 2u * threadIdx.x; q < nwords; q += 2u * kWorkgroup) {
				const uint4 o = *reinterpret_cast<const uint4 *>(&img[q]);
				*reinterpret_cast<uint4 *>(&img[q]) = make_uint4(0u, 0u, 0u, 0u);
				if (q + 1 < nwords) { // non-temporal: written once, read much later
					typedef uint32_t v4u __attribute__((ext_vector_type(4)));
					v4u qq = {o.x, o.y, o.z, o.w};
					__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(dst + q));
				} else {
					dst[q] = ((unsigned long long)o.y << 32) | o.x;
				}
			}
			asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
		}
		return;
	}
	const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds);
	// stages of whole decode rounds whose packed bits fit 8 KiB on both sides (a round of 64-bit fields is 4 KiB)
	uint32_t stage_rows = ((8u * kStageBytes) / (w_old > w ? w_old : w)) / PER_ROUND * PER_ROUND;
	stage_rows = stage_rows < PER_ROUND ? PER_ROUND : stage_rows;
	for (uint32_t done = 0; done < g.n;) {
		const uint32_t first = g.first + done; // a multiple of PER_ROUND rows: both bit streams start on a 16-byte boundary
		const uint32_t left = g.n - done;
		const uint32_t n = left < stage_rows ? left : stage_rows;
		const uint32_t bit0 = stage_packed(src_words + sd.word_off, first, n, w_old, lds);
		__syncthreads(); // the staged bits are visible; the image is zero (start, or the previous copy-out)
		const uint64_t elem0 = sd.val_off + first;
		auto sink = [&](int32_t base, const U *v, bool full) { // decode_run walks with align 0: base >= 0
			const uint32_t rest = n - (uint32_t)base;
			const uint32_t rows_here = full || rest >= (uint32_t)K ? (uint32_t)K : rest;
			const uint32_t vbits = validity ? validity_window(validity, elem0 + (uint32_t)base, rows_here) : 0xffffffffu;
			if (w <= 32u) { // 32-bit emission (see chunk_string32)
				U x[K];
#pragma unroll
				for (int j = 0; j < K; j++) { // rows past the run leave no bit: x = sub gives field 0
					x[j] = (uint32_t)j < rows_here ? (((vbits >> j) & 1u) ? v[j] : (U)null_bits) : sub;
				}
				const Str128 str = chunk_string32<U>(x, (uint32_t)sub, (uint32_t)wmask, w);
				image_or32(reinterpret_cast<uint32_t *>(img), (uint32_t)base * w, str, (uint32_t)K * w);
				return;
			}
			U f[K];
#pragma unroll
			for (int j = 0; j < K; j++) {
				const U x = ((vbits >> j) & 1u) ? v[j] : (U)null_bits;
				f[j] = (uint32_t)j < rows_here ? (U)((U)(x - sub) & wmask) : (U)0; // rows past the run leave no bit
			}
			uint64_t lo, hi;
			concat_fields<U>(f, w, lo, hi);
			image_or(img, (uint32_t)base * w, lo, hi);
		};
		if (sizeof(U) == 8 && w_old > 32) {
			decode_run<U, true>(lds32, bit0, w_old, add, n, sink);
		} else {
			decode_run<U, false>(lds32, bit0, w_old, add, n, sink);
		}
		__syncthreads(); // every lane's bits are in the image; the staged input may be overwritten
		const uint32_t nwords = (n * w + 63u) >> 6;
		unsigned long long *__restrict__ dst =
		    reinterpret_cast<unsigned long long *>(dst_words) + dd.word_off + (((uint64_t)first * w) >> 6);
		for (uint32_t q = 2u * threadIdx.x; q < nwords; q += 2u * kWorkgroup) {
			const unsigned long long v0 = img[q], v1 = img[q + 1];
			img[q] = 0ull;
			img[q + 1] = 0ull;
			if (q + 1 < nwords) {
				uint4 o;
				o.x = (uint32_t)v0;
				o.y = (uint32_t)(v0 >> 32);
				o.z = (uint32_t)v1;
				o.w = (uint32_t)(v1 >> 32);
				*reinterpret_cast<uint4 *>(dst + q) = o;
			} else {
				dst[q] = v0;
			}
		}
		done += n;
	}
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_analyze_packed_g(const ScanGroup *__restrict__ src_groups,
                                                                 const uint64_t *__restrict__ src_words,
                                                                 const uint64_t *__restrict__ validity,
                                                                 int sign_extend, uint64_t null_bits, int rule,
                                                                 int templated, uint64_t *__restrict__ minmax) {
	constexpr uint32_t TILE = kTileBytes / sizeof(U);
	using S = typename std::make_signed<U>::type;
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ uint64_t pmin[kWorkgroup / 64], pmax[kWorkgroup / 64];
	const ScanGroup g = load_scan_group(src_groups, blockIdx.x);
	const adac_segment_desc &sd = g.d;
	const uint32_t w_old = sd.width;
	const uint64_t add = effective_add(sd);
	const uint32_t fit = stage_tiles<U>(w_old, w_old);
	const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds);
	uint64_t mn = ~0ull, mx = 0;
	const bool reg_path = templated && !validity && w_old >= 4u && w_old <= 32u && (uint64_t)sd.count * w_old < (1ull << 31);
	if (reg_path) { // width-templated source walk (analyze_run_w), no LDS staging
		uint32_t mn32 = 0xffffffffu, mx32 = 0u;
		analyze_run_dispatch<U>(w_old, reinterpret_cast<const uint4 *>(src_words + sd.word_off), g.first, g.first + g.n,
		                        sd.count, (uint32_t)add, mn32, mx32);
		if (mn32 <= mx32) { // this lane saw a row
			uint64_t a, b;
			if (sizeof(U) == 8) {
				a = (uint64_t)mn32 + add;
				b = (uint64_t)mx32 + add;
			} else {
				const bool sx = rule == ADAC_RULE_APPEND && sign_extend;
				a = sx ? (uint64_t)(int64_t)(S)(U)mn32 : (uint64_t)mn32;
				b = sx ? (uint64_t)(int64_t)(S)(U)mx32 : (uint64_t)mx32;
			}
			mn = a;
			mx = b;
		}
	}
	for (uint32_t done = reg_path ? g.n : 0u; done < g.n;) {
		const uint32_t first = g.first + done;
		const uint32_t left = g.n - done;
		const uint32_t n = left < fit * TILE ? left : fit * TILE;
		const uint32_t bit0 = stage_packed(src_words + sd.word_off, first, n, w_old, lds);
		__syncthreads();
		const uint64_t elem0 = sd.val_off + first;
		auto sink = [&](int32_t base, const U *v, bool full) {
			constexpr int K = 16 / (int)sizeof(U);
			const uint32_t rest = n - (uint32_t)base;
			const uint32_t rows_here = full || rest >= (uint32_t)K ? (uint32_t)K : rest;
			const uint32_t vbits = validity ? validity_window(validity, elem0 + (uint32_t)base, rows_here) : 0xffffffffu;
#pragma unroll
			for (int j = 0; j < K; j++) {
				if ((uint32_t)j >= rows_here) continue;
				const bool valid = (vbits >> j) & 1u;
				uint64_t x;
				if (rule == ADAC_RULE_APPEND) { // succinct.cpp:286-287: NULL rows do not take part
					if (!valid) continue;
					x = sign_extend ? (uint64_t)(int64_t)(S)v[j] : (uint64_t)v[j];
				} else { // column_segment.cpp:392-399: every slot, zero-extended; NULL slots hold NullValue<T>
					x = valid ? (uint64_t)v[j] : null_bits;
				}
				mn = x < mn ? x : mn;
				mx = x > mx ? x : mx;
			}
		};
		if (sizeof(U) == 8 && w_old > 32) {
			decode_run<U, true>(lds32, bit0, w_old, add, n, sink);
		} else {
			decode_run<U, false>(lds32, bit0, w_old, add, n, sink);
		}
		__syncthreads(); // the staged input is rewritten by the next stage
		done += n;
	}
	mn = wave_min(mn);
	mx = wave_max(mx);
	if ((threadIdx.x & 63) == 0) {
		pmin[threadIdx.x >> 6] = mn;
		pmax[threadIdx.x >> 6] = mx;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
#pragma unroll
		for (int i = 1; i < kWorkgroup / 64; i++) {
			mn = pmin[i] < mn ? pmin[i] : mn;
			mx = pmax[i] > mx ? pmax[i] : mx;
		}
		atomicMin(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)g.seg), (unsigned long long)mn);
		atomicMax(reinterpret_cast<unsigned long long *>(minmax + 2 * (uint64_t)g.seg + 1), (unsigned long long)mx);
	}
}

// ---------------------------------------------------------------------------------------------
// k_fetch — point look-ups straight from HBM (two 8-byte loads per row).
// ---------------------------------------------------------------------------------------------
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_fetch(const adac_segment_desc *__restrict__ descs,
                                                      const uint64_t *__restrict__ words,
                                                      const uint32_t *__restrict__ segs,
                                                      const uint32_t *__restrict__ rows, uint64_t n,
                                                      U *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * kWorkgroup + threadIdx.x;
	if (k >= n) return;
	const adac_segment_desc d = descs[segs[k]];
	const uint32_t w = d.width;
	const uint64_t bit = (uint64_t)rows[k] * w;
	const uint64_t *p = words + d.word_off + (bit >> 6);
	const uint32_t off = (uint32_t)(bit & 63);
	uint64_t v = p[0] >> off;
	if (off + w > 64) v |= p[1] << (64 - off);
	v &= mask64(w);
	out[k] = (U)(v + effective_add(d));
}

template <typename F>
hipError_t dispatch_size(uint32_t type_size, F &&f) {
	switch (type_size) {
	case 1: return f(uint8_t {});
	case 2: return f(uint16_t {});
	case 4: return f(uint32_t {});
	case 8: return f(uint64_t {});
	default: return hipErrorInvalidValue;
	}
}

#include "adac_bitpacking.inl"
#include "adac_select_gather.inl"
#include "adac_block_image.inl"
#include "adac_encode_1p.inl"
#include "adac_group_sum.inl"

} // namespace

Tuning g_tuning;

namespace {

// Compute units of the CURRENT device (every launcher runs after hipSetDevice(ctx->device)): the persistent kernels
// size their grids from it — 256 on a whole MI355X, fewer on a partitioned one (CPX / DPX) — so that "one workgroup
// per CU, all resident" holds on whatever device the pool lives on.  adac_set_tuning("num_cus", n > 0) overrides.
uint64_t device_cus() {
	if (g_tuning.num_cus > 0) return (uint64_t)g_tuning.num_cus;
	static int cached[64]; // 0 = not asked yet; benign race: every thread writes the same value
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
	if (cached[dev] == 0) {
		int n = 0;
		if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
		cached[dev] = n;
	}
	return (uint64_t)cached[dev];
}

unsigned persistent_grid(uint64_t ntiles) {
	const uint64_t cap = device_cus() * (uint64_t)g_tuning.blocks_per_cu;
	return (unsigned)(ntiles < cap ? ntiles : cap);
}

} // namespace

hipError_t launch_minmax_init(hipStream_t s, uint64_t *d_minmax, uint64_t nseg) {
	if (nseg == 0) return hipSuccess;
	hipLaunchKernelGGL(k_minmax_init, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, s, d_minmax, nseg);
	return hipGetLastError();
}

hipError_t launch_analyze(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                          const adac_segment_desc *d_descs, const TileRef *d_tiles, uint64_t ntiles,
                          const void *d_vals, const uint64_t *d_validity, uint64_t *d_minmax) {
	if (ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_analyze<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles,
		                   static_cast<const U *>(d_vals), d_validity, sign_extend ? 1 : 0, null_bits, rule, d_minmax);
		return hipGetLastError();
	});
}

hipError_t launch_plan(hipStream_t s, uint32_t type_size, int rule, int pad_to_byte, adac_segment_desc *d_descs,
                       const uint64_t *d_minmax, uint64_t nseg) {
	if (nseg == 0) return hipSuccess;
	hipLaunchKernelGGL(k_plan, dim3(1), dim3(kPlanThreads), 0, s, d_descs, d_minmax, nseg, type_size * 8, rule,
	                   pad_to_byte);
	return hipGetLastError();
}

hipError_t launch_pack(hipStream_t s, uint32_t type_size, uint64_t null_bits, const adac_segment_desc *d_descs,
                       const TileRef *d_tiles, uint64_t ntiles, const void *d_vals, const uint64_t *d_validity,
                       uint64_t *d_words) {
	if (ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_pack<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles,
		                   static_cast<const U *>(d_vals), d_validity, null_bits, d_words);
		return hipGetLastError();
	});
}

hipError_t launch_analyze_packed(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                                 const adac_segment_desc *d_src_descs, const TileRef *d_tiles, uint64_t ntiles,
                                 const uint64_t *d_src_words, const uint64_t *d_validity, uint64_t *d_minmax) {
	if (ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_analyze_packed<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_src_descs, d_tiles,
		                   d_src_words, d_validity, sign_extend ? 1 : 0, null_bits, rule, d_minmax);
		return hipGetLastError();
	});
}

hipError_t launch_repack(hipStream_t s, uint32_t type_size, uint64_t null_bits, const adac_segment_desc *d_src_descs,
                         const adac_segment_desc *d_dst_descs, const TileRef *d_tiles, uint64_t ntiles,
                         const uint64_t *d_src_words, const uint64_t *d_validity, uint64_t *d_dst_words) {
	if (ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_repack<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_src_descs, d_dst_descs,
		                   d_tiles, d_src_words, d_validity, null_bits, d_dst_words);
		return hipGetLastError();
	});
}

hipError_t launch_encode_1p(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                            int pad_to_byte, adac_segment_desc *d_descs, uint64_t nseg, const void *d_vals,
                            const uint64_t *d_validity, uint64_t *d_minmax, void *d_scan_state, uint64_t *d_words) {
	if (nseg == 0) return hipSuccess;
	// scan_state: nseg look-back words followed by the ticket counter and the first-come cursor, all zero before the launch
	hipError_t e = hipMemsetAsync(d_scan_state, 0, encode_1p_state_words(nseg) * sizeof(unsigned long long), s);
	if (e != hipSuccess) return e;
	// persistent: one workgroup per CU (a segment fills half a CU's register file), segments handed out by ticket
	const uint64_t cus = device_cus();
	const unsigned grid = (unsigned)(nseg < cus ? nseg : cus);
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		unsigned long long *state = static_cast<unsigned long long *>(d_scan_state);
		hipLaunchKernelGGL(k_encode_1p<U>, dim3(grid), dim3(kEncThreads), 0, s, d_descs, d_descs, (uint32_t)nseg, d_minmax,
		                   static_cast<const U *>(d_vals), d_validity, sign_extend ? 1 : 0, null_bits, rule, pad_to_byte,
		                   state, reinterpret_cast<uint32_t *>(state + nseg), d_words, g_tuning.encode_stamps,
		                   (g_tuning.encode_placement & 1) | (g_tuning.encode_publish_ahead ? 2 : 0) |
		                       (g_tuning.encode_big_image ? 4 : 0));
		return hipGetLastError();
	});
}

// the per-workgroup partials of both scan kernels + two words used in turn: the segment pairs the register-walk kernel
// left (the buffer is cleared when it is allocated; k_group_final clears the word of the call after it)
uint64_t group_sum_partial_bytes() { return ((uint64_t)kGroupMaxWorkgroups * 2u * kGroupMaxBins + 2u) * sizeof(unsigned long long); }
uint32_t group_sum_max_groups() { return kGroupMaxBins - 1u; }

hipError_t launch_group_sum(hipStream_t s, uint32_t v_type_size, bool v_signed, uint32_t k_type_size,
                            const adac_segment_desc *d_vdescs, const TileRef *d_vtiles, uint64_t ntiles,
                            const ScanGroup *d_vgroups, uint64_t nvgroups, const uint64_t *d_vwords,
                            const adac_segment_desc *d_kdescs, const uint64_t *d_kwords, uint32_t ngroups, void *d_partial,
                            uint32_t call_parity, uint64_t *d_sums, uint64_t *d_counts) {
	GroupSumTypes ty;
	ty.v_tmask = v_type_size >= 8 ? ~0ull : ((1ull << (8 * v_type_size)) - 1ull);
	ty.v_sbit = v_signed ? (1ull << (8 * v_type_size - 1)) : 0ull;
	ty.k_tmask = k_type_size >= 8 ? ~0ull : ((1ull << (8 * k_type_size)) - 1ull);
	ty.v_tile_rows = tile_values(v_type_size);
	ty.wide_only = g_tuning.group_sum_wide ? 1u : 0u;
	const uint32_t nbins = ngroups + 1u;
	unsigned long long *partial = static_cast<unsigned long long *>(d_partial);
	unsigned long long *fallback = partial + (uint64_t)kGroupMaxWorkgroups * 2u * kGroupMaxBins + (call_parity & 1u);
	unsigned long long *next_fallback = partial + (uint64_t)kGroupMaxWorkgroups * 2u * kGroupMaxBins + ((call_parity + 1u) & 1u);
	// 1. the register-walk kernel over the value layout's scan groups (up to 8 bins; persistent, seven workgroups per CU);
	//    it counts the segment pairs it cannot take in *fallback
	uint32_t nwg_rw = 0;
	const bool rw = nbins <= kGroupPrivateBins && !ty.wide_only && g_tuning.group_sum_rw && nvgroups > 0;
	if (rw) {
		const uint64_t cap = 7ull * device_cus();
		nwg_rw = (uint32_t)(nvgroups < cap ? nvgroups : cap);
		nwg_rw = nwg_rw < kGroupMaxWorkgroups / 2 ? nwg_rw : kGroupMaxWorkgroups / 2;
		hipLaunchKernelGGL(k_group_sum_rw, dim3(nwg_rw), dim3(kWorkgroup), 0, s, d_vgroups, (uint32_t)nvgroups, d_vwords,
		                   d_kdescs, d_kwords, ty, ngroups, partial, fallback);
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	// 2. the staged-LDS kernel over the tiles: everything when the first kernel did not run, else what it left (its
	//    workgroups leave at once when that is nothing).  Persistent: as many workgroups as are resident at once (seven
	//    per CU: 21 KiB of LDS each), so nobody runs a second round on a third of the chip
	uint64_t cap = 7ull * device_cus();
	cap = cap < kGroupMaxWorkgroups / 2 ? cap : kGroupMaxWorkgroups / 2;
	const uint32_t nwg = (uint32_t)(ntiles < cap ? ntiles : cap);
	if (nwg) {
		hipLaunchKernelGGL(k_group_sum, dim3(nwg), dim3(kWorkgroup), 0, s, d_vdescs, d_vtiles, (uint32_t)ntiles, d_vwords,
		                   d_kdescs, d_kwords, ty, ngroups, partial + (uint64_t)nwg_rw * 2u * nbins,
		                   rw ? fallback : static_cast<const unsigned long long *>(nullptr));
		hipError_t e = hipGetLastError();
		if (e != hipSuccess) return e;
	}
	hipLaunchKernelGGL(k_group_final, dim3(nbins), dim3(kWorkgroup), 0, s, partial, nwg_rw, nwg, nbins, d_sums, d_counts,
	                   static_cast<const unsigned long long *>(fallback), next_fallback, rw ? 1 : 0);
	return hipGetLastError();
}

hipError_t read_encode_stamps(void *host, uint64_t bytes) {
	const uint64_t cap = sizeof(unsigned long long) * kEncStampMax * kEncStampSlots;
	return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_enc_stamps), bytes < cap ? bytes : cap);
}

hipError_t launch_analyze_packed_g(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                                   const ScanGroup *d_src_groups, uint64_t ngroups, const uint64_t *d_src_words,
                                   const uint64_t *d_validity, uint64_t *d_minmax) {
	if (ngroups == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_analyze_packed_g<U>, dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s, d_src_groups,
		                   d_src_words, d_validity, sign_extend ? 1 : 0, null_bits, rule, g_tuning.templated_scan, d_minmax);
		return hipGetLastError();
	});
}

hipError_t launch_repack_g(hipStream_t s, uint32_t type_size, uint64_t null_bits, const ScanGroup *d_src_groups,
                           uint64_t ngroups, const adac_segment_desc *d_dst_descs, const uint64_t *d_src_words,
                           const uint64_t *d_validity, uint64_t *d_dst_words) {
	if (ngroups == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_repack_g<U>, dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s, d_src_groups, d_dst_descs,
		                   d_src_words, d_validity, null_bits, g_tuning.templated_scan, d_dst_words);
		return hipGetLastError();
	});
}

hipError_t launch_expand_tiles(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const TileRef *d_tiles,
                               uint64_t ntiles, TileRec *d_recs) {
	if (ntiles == 0) return hipSuccess;
	hipLaunchKernelGGL(k_expand_tiles, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, d_descs, d_tiles, ntiles,
	                   (uint32_t)(kTileBytes / type_size), d_recs);
	return hipGetLastError();
}

hipError_t launch_unpack(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const TileRef *d_tiles,
                         uint64_t ntiles, const uint64_t *d_words, void *d_out, const TileRec *d_recs) {
	if (ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		if (g_tuning.persistent_unpack) {
			const unsigned grid = persistent_grid(ntiles);
			hipLaunchKernelGGL(k_unpack_p<U>, dim3(grid), dim3(kWorkgroup), 0, s, d_descs, d_tiles, (uint32_t)ntiles,
			                   d_words, static_cast<U *>(d_out));
			return hipGetLastError();
		}
		if (d_recs) {
			hipLaunchKernelGGL((k_unpack<U, false, true>), dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles,
			                   RangeArgs {}, d_words, static_cast<U *>(d_out), d_recs);
			return hipGetLastError();
		}
		hipLaunchKernelGGL((k_unpack<U, false>), dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles,
		                   RangeArgs {}, d_words, static_cast<U *>(d_out), static_cast<const TileRec *>(nullptr));
		return hipGetLastError();
	});
}

hipError_t launch_unpack_range(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, RangeArgs range,
                               const uint64_t *d_words, void *d_out) {
	if (range.count == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const uint32_t tile = kTileBytes / sizeof(U);
		const uint32_t nt = (range.count + tile - 1) / tile;
		hipLaunchKernelGGL((k_unpack<U, true>), dim3(nt), dim3(kWorkgroup), 0, s, d_descs,
		                   static_cast<const TileRef *>(nullptr), range, d_words, static_cast<U *>(d_out),
		                   static_cast<const TileRec *>(nullptr));
		return hipGetLastError();
	});
}

hipError_t launch_unpack_jobs(hipStream_t s, uint32_t type_size, const UnpackJobTable &table, const uint64_t *d_words,
                              void *d_out) {
	if (table.njobs == 0 || table.ntiles == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_unpack_jobs<U>, dim3(table.ntiles), dim3(kWorkgroup), 0, s, table, d_words,
		                   static_cast<U *>(d_out));
		return hipGetLastError();
	});
}

hipError_t launch_fetch(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const uint64_t *d_words,
                        const uint32_t *d_segs, const uint32_t *d_rows, uint64_t n, void *d_out) {
	if (n == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_fetch<U>, dim3((unsigned)((n + kWorkgroup - 1) / kWorkgroup)), dim3(kWorkgroup), 0, s,
		                   d_descs, d_words, d_segs, d_rows, n, static_cast<U *>(d_out));
		return hipGetLastError();
	});
}

hipError_t launch_expand_groups(hipStream_t s, const adac_segment_desc *d_descs, const ScanGroupRef *d_refs,
                                uint64_t ngroups, ScanGroup *d_groups, uint32_t *d_narrow_idx, uint32_t *d_narrow_count) {
	hipError_t e = hipMemsetAsync(d_narrow_count, 0, sizeof(uint32_t), s);
	if (e != hipSuccess || ngroups == 0) return e;
	hipLaunchKernelGGL(k_expand_groups, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s, d_descs, d_refs,
	                   ngroups, d_groups, d_narrow_idx, d_narrow_count);
	return hipGetLastError();
}

hipError_t launch_scan_sum(hipStream_t s, uint32_t type_size, const ScanGroupList &gl, const uint64_t *d_words,
                           const uint64_t *d_validity, uint64_t sbit, uint64_t *d_sums) {
	if (gl.ngroups == 0) return hipSuccess;
	const RangePred widen {0ull, 0ull, sbit}; // SUM only needs T's sign bit
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const dim3 grid((unsigned)gl.ngroups);
		const int tpl = g_tuning.templated_scan | (g_tuning.sel_debug == 7 ? 7 << 1 : 0);
		uint32_t *no_bitmap = nullptr;
		SelEdge *no_edges = nullptr;
		if (g_tuning.scan_probe) { // diagnostic: the scan's loop and loads without the field walk (result meaningless)
			hipLaunchKernelGGL((k_scan_agg<U, 2, false, false>), grid, dim3(kWorkgroup), 0, s, gl.d_groups, gl.d_narrow_idx,
			                   1, d_words, RangePred {}, static_cast<const uint64_t *>(nullptr), d_sums, no_bitmap, no_edges,
			                   gl.d_res_cells, no_bitmap, 0u, ~0ull);
		} else if (d_validity) {
			hipLaunchKernelGGL((k_scan_agg<U, 0, true, false>), grid, dim3(kWorkgroup), 0, s, gl.d_groups, gl.d_narrow_idx,
			                   tpl, d_words, widen, d_validity, d_sums, no_bitmap, no_edges, gl.d_res_cells, no_bitmap, 0u, ~0ull);
		} else {
			hipLaunchKernelGGL((k_scan_agg<U, 0, false, false>), grid, dim3(kWorkgroup), 0, s, gl.d_groups, gl.d_narrow_idx,
			                   tpl, d_words, widen, d_validity, d_sums, no_bitmap, no_edges, gl.d_res_cells, no_bitmap, 0u, ~0ull);
			if (gl.n_narrow) {
				hipLaunchKernelGGL((k_scan_agg<U, 0, false, true>), dim3(gl.n_narrow), dim3(kWorkgroup), 0, s, gl.d_groups,
				                   gl.d_narrow_idx, tpl, d_words, widen, d_validity, d_sums, no_bitmap, no_edges,
				                   gl.d_res_cells, no_bitmap, 0u, ~0ull);
			}
		}
		return hipGetLastError();
	});
}

uint64_t sel_edge_bytes(uint64_t ngroups) { return 2 * ngroups * sizeof(SelEdge); }

hipError_t launch_sel_merge_edges(hipStream_t s, const void *d_edges, uint64_t ngroups, uint64_t *d_bitmap,
                                  uint64_t tail_word) {
	const uint64_t nrec = 2 * ngroups;
	hipLaunchKernelGGL(k_sel_merge_edges, dim3((unsigned)((nrec + 255) / 256 + (nrec == 0))), dim3(256), 0, s,
	                   static_cast<const SelEdge *>(d_edges), nrec, reinterpret_cast<uint32_t *>(d_bitmap), tail_word);
	return hipGetLastError();
}

hipError_t launch_scan_count_range(hipStream_t s, uint32_t type_size, const ScanGroupList &gl, const uint64_t *d_words,
                                   const uint64_t *d_validity, uint64_t blo, uint64_t bspan, uint64_t sbit,
                                   uint64_t *d_counts, uint64_t *d_bitmap, void *d_edges, bool edge_cells,
                                   uint64_t tail_word) {
	if (gl.ngroups == 0) return hipSuccess;
	uint32_t *const ecells = edge_cells ? gl.d_edge_cells : nullptr;
	const uint32_t last_group = (uint32_t)(gl.ngroups - 1);
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const dim3 grid((unsigned)gl.ngroups);
		const RangePred pred {blo, bspan, sbit};
		uint32_t *bm = reinterpret_cast<uint32_t *>(d_bitmap);
		const int tpl = g_tuning.templated_scan | ((g_tuning.sel_debug == 5 ? 0 : g_tuning.sel_debug) << 1);
		// the common kernel over every group (it leaves the groups at widths 2 and 3 alone), then the narrow one over those
#define ADAC_SCAN(OPN, VAL)                                                                                            \
	do {                                                                                                               \
		hipLaunchKernelGGL((k_scan_agg<U, OPN, VAL, false>), grid, dim3(kWorkgroup), 0, s, gl.d_groups, gl.d_narrow_idx, \
		                   tpl, d_words, pred, d_validity, d_counts, bm, static_cast<SelEdge *>(d_edges),                \
		                   gl.d_res_cells, ecells, last_group, tail_word);                                               \
		if (gl.n_narrow) {                                                                                             \
			hipLaunchKernelGGL((k_scan_agg<U, OPN, VAL, true>), dim3(gl.n_narrow), dim3(kWorkgroup), 0, s, gl.d_groups,  \
			                   gl.d_narrow_idx, tpl, d_words, pred, d_validity, d_counts, bm,                            \
			                   static_cast<SelEdge *>(d_edges), gl.d_res_cells, ecells, last_group, tail_word);          \
		}                                                                                                              \
	} while (0)
		if (d_bitmap) {
			if (d_validity) ADAC_SCAN(3, true); else ADAC_SCAN(3, false);
		} else {
			if (d_validity) ADAC_SCAN(1, true); else ADAC_SCAN(1, false);
		}
#undef ADAC_SCAN
		return hipGetLastError();
	});
}

hipError_t launch_blocks_write(hipStream_t s, const BlockJob *d_jobs, uint64_t njobs, uint32_t max_units,
                               const uint64_t *d_words, void *d_blocks) {
	if (njobs == 0) return hipSuccess;
	const uint32_t chunks = (max_units + kImageChunk - 1) / kImageChunk;
	hipLaunchKernelGGL(k_blocks_write, dim3((unsigned)(njobs * chunks)), dim3(kWorkgroup), 0, s, d_jobs, chunks, d_words,
	                   static_cast<uint64_t *>(d_blocks));
	return hipGetLastError();
}

hipError_t launch_blocks_read(hipStream_t s, const BlockJob *d_jobs, uint64_t njobs, uint32_t max_units,
                              const void *d_blocks, uint64_t *d_words, uint32_t *d_bad) {
	if (njobs == 0) return hipSuccess;
	const uint32_t chunks = (max_units + kImageChunk - 1) / kImageChunk;
	hipLaunchKernelGGL(k_blocks_read, dim3((unsigned)(njobs * chunks)), dim3(kWorkgroup), 0, s, d_jobs, chunks,
	                   static_cast<const uint64_t *>(d_blocks), d_words, d_bad);
	return hipGetLastError();
}

hipError_t launch_bp_prepare(hipStream_t s, uint32_t type_size, void *d_groups, uint64_t ngroups, const void *d_blocks) {
	if (ngroups == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_bp_prepare<U>, dim3((unsigned)((ngroups + 255) / 256)), dim3(256), 0, s,
		                   static_cast<BpGroup *>(d_groups), ngroups, static_cast<const uint8_t *>(d_blocks));
		return hipGetLastError();
	});
}

hipError_t launch_bp_stats(hipStream_t s, uint32_t type_size, bool is_signed, const void *d_vals,
                           const uint64_t *d_validity, uint64_t n, void *d_stats) {
	if (n == 0) return hipSuccess;
	const uint64_t ngroups = (n + kBpGroupRows - 1) / kBpGroupRows;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const uint64_t sbit = is_signed ? (1ull << (8 * sizeof(U) - 1)) : 0ull;
		hipLaunchKernelGGL(k_bp_stats<U>, dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s, static_cast<const U *>(d_vals),
		                   d_validity, n, sbit, static_cast<BpStats *>(d_stats));
		return hipGetLastError();
	});
}

hipError_t launch_bp_write(hipStream_t s, uint32_t type_size, const void *d_recs, uint64_t ngroups, const void *d_vals,
                           const uint64_t *d_validity, uint64_t block_stride, void *d_blocks) {
	if (ngroups == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_bp_write<U>, dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s,
		                   static_cast<const BpWrite *>(d_recs), static_cast<const U *>(d_vals), d_validity, block_stride,
		                   static_cast<uint8_t *>(d_blocks));
		return hipGetLastError();
	});
}

hipError_t launch_bp_unpack(hipStream_t s, uint32_t type_size, const void *d_groups, uint64_t ngroups,
                            const void *d_blocks, void *d_out) {
	if (ngroups == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL((k_bp_unpack<U, false>), dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s,
		                   static_cast<const BpGroup *>(d_groups), static_cast<const uint8_t *>(d_blocks), BpRangeArgs {},
		                   static_cast<U *>(d_out));
		return hipGetLastError();
	});
}

hipError_t launch_bp_unpack_range(hipStream_t s, uint32_t type_size, const void *d_groups, uint32_t group0,
                                  uint32_t skip_first, uint64_t count, uint64_t out_off, const void *d_blocks,
                                  void *d_out) {
	if (count == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const uint64_t ngroups = ((uint64_t)skip_first + count + kBpGroupRows - 1) / kBpGroupRows;
		hipLaunchKernelGGL((k_bp_unpack<U, true>), dim3((unsigned)ngroups), dim3(kWorkgroup), 0, s,
		                   static_cast<const BpGroup *>(d_groups), static_cast<const uint8_t *>(d_blocks),
		                   BpRangeArgs {group0, skip_first, count, out_off}, static_cast<U *>(d_out));
		return hipGetLastError();
	});
}

hipError_t launch_bp_fetch(hipStream_t s, uint32_t type_size, const uint64_t *d_block_offs, const void *d_blocks,
                           const uint32_t *d_segs, const uint32_t *d_rows, uint64_t n, void *d_out) {
	if (n == 0) return hipSuccess;
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		hipLaunchKernelGGL(k_bp_fetch<U>, dim3((unsigned)((n + kWorkgroup - 1) / kWorkgroup)), dim3(kWorkgroup), 0, s,
		                   d_block_offs, static_cast<const uint8_t *>(d_blocks), d_segs, d_rows, n,
		                   static_cast<U *>(d_out));
		return hipGetLastError();
	});
}

hipError_t launch_gather_selected(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs,
                                  const TileRef *d_tiles, uint64_t ntiles, const uint64_t *d_words,
                                  const TileRec *d_recs, const uint64_t *d_bitmap, uint64_t bitmap_words, uint32_t *d_tile_cnt,
                                  uint64_t *d_tile_off, uint64_t *d_block_tot, void *d_out, uint64_t *d_out_ids,
                                  uint64_t *d_total) {
	if (ntiles == 0) return hipMemsetAsync(d_total, 0, sizeof(uint64_t), s);
	return dispatch_size(type_size, [&](auto tag) {
		using U = decltype(tag);
		const unsigned per = kWorkgroup / 64;
		hipLaunchKernelGGL(k_tile_popc<U>, dim3((unsigned)((ntiles + per - 1) / per)), dim3(kWorkgroup), 0, s, d_descs,
		                   d_tiles, (uint32_t)ntiles, d_bitmap, d_tile_cnt);
		const uint64_t nblocks = (ntiles + kScanBlock - 1) / kScanBlock;
		hipLaunchKernelGGL(k_scan_blocks, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_tile_cnt, ntiles, d_tile_off,
		                   d_block_tot);
		hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(kScanBlock), 0, s, d_block_tot, nblocks, d_total);
		hipLaunchKernelGGL(k_scan_fixup, dim3((unsigned)nblocks), dim3(kScanBlock), 0, s, d_tile_off, ntiles, d_block_tot);
		if (g_tuning.gather_compact && bitmap_words < (1ull << 31)) { // (32-bit dword indices into the bitmap)
			hipLaunchKernelGGL(k_gather_c<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles, d_recs, d_words,
			                   d_bitmap, (uint32_t)(2 * bitmap_words - 1), d_tile_cnt, d_tile_off, static_cast<U *>(d_out), d_out_ids,
			                   g_tuning.gather_compact & 2);
		} else {
			hipLaunchKernelGGL(k_gather<U>, dim3((unsigned)ntiles), dim3(kWorkgroup), 0, s, d_descs, d_tiles, d_words,
			                   d_bitmap, d_tile_off, static_cast<U *>(d_out), d_out_ids);
		}
		return hipGetLastError();
	});
}

} // namespace adac
