// adac_bitpacking.inl — device decode of DuckDB's on-disk BITPACKING segments (the persistent
// counterpart of the succinct codec).  Included into adac_kernels.hip (same translation unit: it reuses the
// packed-field reader, the LDS staging and the aligned-store sink of the succinct kernels — a 32-value
// fastpforlib algorithm group is the same contiguous little-endian bit stream as an sdsl::int_vector).
//
// Block layout decoded here (paths relative to the reference checkout):
//   [0,8)        offset of the byte just past the FIRST group's metadata entry    bitpacking.cpp:496-509
//   data         grows up from byte 8; per metadata group (2048 rows):            bitpacking.cpp:374-437
//                  CONSTANT        T constant
//                  CONSTANT_DELTA  T frame_of_reference, T_S delta
//                  FOR             T frame_of_reference, T width, packed (v - for)
//                  DELTA_FOR       T frame_of_reference, T width, T_S delta_offset, packed (delta - for)
//   metadata     one uint32 per group, mode << 24 | data offset; group g's entry at [first - 4(g+1), first - 4g)
// Decode semantics: BitpackingScanState::LoadNextGroup + BitpackingScanPartial (bitpacking.cpp:583-640,736-821).

struct BpGroup {
	uint64_t block_off; // byte offset of the segment's block in the blocks buffer (16-byte aligned)
	uint64_t out_off;   // element offset of the group's first row in the output
	uint32_t group;     // metadata group index inside the segment
	uint32_t rows;      // rows in this group (2048 except the segment's last)
	// filled by k_bp_prepare from the block image (adac_bp_bind): the group's parsed header, so the scan kernel
	// starts its payload loads after ONE descriptor load instead of a chain of four dependent ones
	uint64_t payload_off; // byte offset of the packed fields in the blocks buffer
	uint64_t frame, extra;
	uint32_t mode, width;
};

constexpr int kBpGroupRows = 2048; // BITPACKING_METADATA_GROUP_SIZE (bitpacking.cpp:19)
enum : uint32_t { kBpConstant = 1, kBpConstantDelta = 2, kBpDeltaFor = 3, kBpFor = 4 }; // bitpacking.hpp:15-22

// T value at an arbitrarily aligned address (group headers are only sizeof(T)-packed)
template <typename U>
__device__ __forceinline__ U load_unaligned(const uint8_t *p) {
	uint64_t v = 0;
#pragma unroll
	for (int i = 0; i < (int)sizeof(U); i++) v |= (uint64_t)p[i] << (8 * i);
	return (U)v;
}

struct BpHeader {
	uint32_t mode, width;
	uint64_t frame, extra;  // extra: CONSTANT_DELTA's delta / DELTA_FOR's delta_offset
	const uint8_t *payload; // packed fields (FOR / DELTA_FOR)
};

template <typename U>
__device__ __forceinline__ BpHeader bp_header(const uint8_t *blk, uint32_t group) {
	BpHeader h;
	const uint64_t first = *reinterpret_cast<const uint64_t *>(blk);
	const uint32_t enc = *reinterpret_cast<const uint32_t *>(blk + first - 4ull * (group + 1));
	h.mode = enc >> 24;
	const uint8_t *p = blk + (enc & 0x00ffffffu);
	h.frame = (uint64_t)load_unaligned<U>(p); // CONSTANT: the constant
	h.width = 0;
	h.extra = 0;
	h.payload = p;
	if (h.mode == kBpConstantDelta) {
		h.extra = (uint64_t)load_unaligned<U>(p + sizeof(U));
	} else if (h.mode == kBpFor || h.mode == kBpDeltaFor) {
		h.width = (uint32_t)load_unaligned<U>(p + sizeof(U)) & 0xffu; // (bitpacking_width_t) *(T *)ptr
		h.payload = p + 2 * sizeof(U);
		if (h.mode == kBpDeltaFor) {
			h.extra = (uint64_t)load_unaligned<U>(h.payload);
			h.payload += sizeof(U);
		}
	}
	return h;
}

template <typename U>
__global__ void k_bp_prepare(BpGroup *__restrict__ groups, uint64_t ngroups, const uint8_t *__restrict__ blocks) {
	const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ngroups) return;
	BpGroup g = groups[i];
	const BpHeader h = bp_header<U>(blocks + g.block_off, g.group);
	g.payload_off = (uint64_t)(h.payload - blocks);
	g.frame = h.frame;
	g.extra = h.extra;
	g.mode = h.mode;
	g.width = h.width;
	groups[i] = g;
}

// Inclusive wave64 prefix sum in registers with DPP (no LDS round trips: a ds_bpermute-based __shfl_up scan is a
// chain of six ~100-cycle dependent steps).  row_shr:1/2/4/8 scan each row of 16 lanes (lanes without a source
// get 0), row_bcast:15 adds row totals into rows 1 and 3, row_bcast:31 adds lane 31's total into rows 2 and 3.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
	return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_step(uint64_t v) {
	const uint32_t lo = dpp_or_zero<CTRL, ROW_MASK>((uint32_t)v);
	const uint32_t hi = dpp_or_zero<CTRL, ROW_MASK>((uint32_t)(v >> 32));
	return v + (((uint64_t)hi << 32) | lo);
}

template <typename U>
__device__ __forceinline__ U wave_inclusive_sum(U x) {
	if (sizeof(U) == 8) {
		uint64_t v = (uint64_t)x;
		v = dpp_step<0x111, 0xf>(v); // row_shr:1
		v = dpp_step<0x112, 0xf>(v); // row_shr:2
		v = dpp_step<0x114, 0xf>(v); // row_shr:4
		v = dpp_step<0x118, 0xf>(v); // row_shr:8
		v = dpp_step<0x142, 0xa>(v); // row_bcast:15 -> rows 1, 3
		v = dpp_step<0x143, 0xc>(v); // row_bcast:31 -> rows 2, 3
		return (U)v;
	}
	uint32_t v = (uint32_t)x; // narrower T: sums wrap mod 2^bits, which truncation preserves
	v += dpp_or_zero<0x111, 0xf>(v);
	v += dpp_or_zero<0x112, 0xf>(v);
	v += dpp_or_zero<0x114, 0xf>(v);
	v += dpp_or_zero<0x118, 0xf>(v);
	v += dpp_or_zero<0x142, 0xa>(v);
	v += dpp_or_zero<0x143, 0xc>(v);
	return (U)v;
}

template <typename U>
struct BpScan {
	static constexpr int K = 16 / (int)sizeof(U);
	// rounds of 256 chunks that cover one metadata group plus an output misalignment of up to K - 1 rows
	static constexpr int ROUNDS = (kBpGroupRows + K - 1 + kWorkgroup * K - 1) / (kWorkgroup * K);
};

template <typename U, bool WIDE>
__device__ __forceinline__ void bp_delta_scan(const uint32_t *lds32, uint32_t bit0, uint32_t w, uint64_t frame,
                                              U delta_offset, uint32_t n, uint32_t align, U *__restrict__ dst,
                                              uint32_t store_from, U (*wave_tot)[kWorkgroup / 64]) {
	constexpr int K = BpScan<U>::K;
	constexpr int ROUNDS = BpScan<U>::ROUNDS;
	constexpr int WAVES = kWorkgroup / 64;
	const uint32_t mlo = WIDE ? 0xffffffffu : mask32(w);
	const uint32_t mhi = WIDE ? mask32(w - 32u) : 0u;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	// every round's chunk is decoded and scanned before the single barrier, so the rounds overlap instead of
	// waiting on each other's carries; rows outside [0, n) contribute zero and store nothing
	U v[ROUNDS][K], run[ROUNDS], incl[ROUNDS];
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		const int32_t base = (int32_t)((r * kWorkgroup + threadIdx.x) * K) - (int32_t)align;
		U acc = 0;
#pragma unroll
		for (int j = 0; j < K; j++) {
			const uint32_t row = (uint32_t)(base + j); // rows < 0 wrap to huge values: out of range
			uint32_t lo, hi;
			read_field<WIDE>(lds32, bit0 + (row < n ? row : 0u) * w, mlo, mhi, lo, hi);
			const U d = sizeof(U) == 8 ? (U)((((uint64_t)hi << 32) | lo) + frame) : (U)(lo + (uint32_t)frame);
			acc = (U)(acc + (row < n ? d : (U)0));
			v[r][j] = acc;
		}
		run[r] = acc;
		incl[r] = wave_inclusive_sum<U>(acc); // inclusive scan of the lane totals inside the wave
		if (lane == 63) wave_tot[r][wave] = incl[r];
	}
	__syncthreads();
	U carry = delta_offset;
#pragma unroll
	for (int r = 0; r < ROUNDS; r++) {
		U before = carry;
#pragma unroll
		for (int wv = 0; wv < WAVES; wv++) {
			const U t = wave_tot[r][wv];
			if ((uint32_t)wv < wave) before = (U)(before + t);
			carry = (U)(carry + t);
		}
		before = (U)(before + (U)(incl[r] - run[r]));
		const int32_t base = (int32_t)((r * kWorkgroup + threadIdx.x) * K) - (int32_t)align;
#pragma unroll
		for (int j = 0; j < K; j++) v[r][j] = (U)(v[r][j] + before);
		// dst is the (possibly virtual) address of row 0; only rows in [store_from, n) are written
		if (base >= (int32_t)store_from && (uint32_t)(base + K) <= n) {
			uint4 q;
			__builtin_memcpy(&q, v[r], 16);
			{ // non-temporal: decoded values are written once and read by somebody else (see StoreSink)
				typedef uint32_t v4u __attribute__((ext_vector_type(4)));
				v4u qq = {q.x, q.y, q.z, q.w};
				__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(dst + base));
			}
		} else {
#pragma unroll
			for (int j = 0; j < K; j++) {
				const uint32_t row = (uint32_t)(base + j);
				if (row < n && row >= store_from) dst[base + j] = v[r][j];
			}
		}
	}
}

// Range form (BitpackingScanPartial with an arbitrary start, bitpacking.cpp:736-826): rows
// [start, start + count) of ONE segment, written to out[out_off ...].  group0 is the index of the metadata group
// holding `start` in the layout's group table, skip_first the row of `start` inside it.
struct BpRangeArgs {
	uint32_t group0;
	uint32_t skip_first;
	uint64_t count;
	uint64_t out_off;
};

template <typename U, bool RANGE>
__global__ __launch_bounds__(kWorkgroup) void k_bp_unpack(const BpGroup *__restrict__ groups,
                                                          const uint8_t *__restrict__ blocks, BpRangeArgs range,
                                                          U *__restrict__ out) {
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ U wave_tot[BpScan<U>::ROUNDS][kWorkgroup / 64];
	const BpGroup g = groups[(RANGE ? range.group0 : 0u) + blockIdx.x];
	BpHeader h;
	h.mode = g.mode;
	h.width = g.width;
	h.frame = g.frame;
	h.extra = g.extra;
	h.payload = blocks + g.payload_off;
	constexpr uint32_t K = 16 / sizeof(U);
	// rows [a, n) of the group are wanted; row r lands at out[shift + r].  (For DELTA_FOR rows below a are still
	// decoded: they are part of the prefix.)
	uint32_t a = 0, n = g.rows;
	int64_t shift = (int64_t)g.out_off;
	if (RANGE) {
		a = blockIdx.x == 0 ? range.skip_first : 0u;
		const uint64_t pos = (uint64_t)blockIdx.x * kBpGroupRows + a - range.skip_first; // of row a in the range
		if (pos >= range.count || a >= n) return;
		const uint64_t want = range.count - pos;
		if (want < (uint64_t)(n - a)) n = a + (uint32_t)want;
		shift = (int64_t)(range.out_off + pos) - (int64_t)a;
	}
	U *dst = out + shift; // address of row 0 of the group (virtual when a > 0: only rows >= a are touched)
	const uint32_t align = (uint32_t)((shift + (int64_t)a) & (int64_t)(K - 1)); // of the first stored row

	if (h.mode == kBpConstant || h.mode == kBpConstantDelta || h.width == 0) {
		// CONSTANT: fill; CONSTANT_DELTA: i*delta + for (bitpacking.cpp:759-780); width 0: every field is zero
		const U fr = (U)h.frame;
		const U step = h.mode == kBpConstantDelta ? (U)h.extra : (U)0;
		if (h.mode == kBpDeltaFor) { // zero-width deltas: v[i] = delta_offset + (i+1)*for
			for (uint32_t i = a + threadIdx.x; i < n; i += kWorkgroup) dst[i] = (U)((U)h.extra + (U)(i + 1) * fr);
		} else {
			for (uint32_t i = a + threadIdx.x; i < n; i += kWorkgroup) dst[i] = (U)((U)i * step + fr);
		}
		return;
	}

	// stage the packed payload: it starts at an arbitrary byte, so align down to 16 and carry the bit offset
	const uintptr_t addr = reinterpret_cast<uintptr_t>(h.payload);
	const uint4 *src = reinterpret_cast<const uint4 *>(addr & ~uintptr_t(15));
	const uint32_t bit0 = (uint32_t)(addr & 15) * 8u;
	const uint32_t w = h.width;
	const uint32_t nchunks = (bit0 + n * w + 127u) >> 7;
	for (uint32_t c = threadIdx.x; c < nchunks; c += kWorkgroup) lds[c] = src[c];
	__syncthreads();
	const uint32_t *lds32 = reinterpret_cast<const uint32_t *>(lds);

	if (h.mode == kBpFor) {
		StoreSink<U> sink {dst + a, n - a}; // value = field + frame_of_reference (ApplyFrameOfReference)
		if (sizeof(U) == 8 && w > 32) {
			decode_rows<U, true>(lds32, bit0 + a * w, w, h.frame, n - a, align, sink);
		} else {
			decode_rows<U, false>(lds32, bit0 + a * w, w, h.frame, n - a, align, sink);
		}
		return;
	}

	// DELTA_FOR: v[i] = delta_offset + sum_{j<=i} (field[j] + for), wrapping in T (bitpacking.cpp:810-813).
	// A lane decodes the K consecutive rows of one 16-byte output chunk into registers and scans them; lanes are
	// chained by a DPP wave64 prefix sum, waves by four totals through LDS, and every round of 256 chunks is
	// decoded and scanned before the single barrier.  The values never visit LDS and leave with 16-byte stores.
	const uint32_t scan_align = (uint32_t)(shift & (int64_t)(K - 1)); // chunks are aligned on row 0's address
	if (sizeof(U) == 8 && w > 32) {
		bp_delta_scan<U, true>(lds32, bit0, w, h.frame, (U)h.extra, n, scan_align, dst, a, wave_tot);
	} else {
		bp_delta_scan<U, false>(lds32, bit0, w, h.frame, (U)h.extra, n, scan_align, dst, a, wave_tot);
	}
}

// Point fetch (BitpackingFetchRow, bitpacking.cpp:827-870): one lane per row; a DELTA_FOR row needs the prefix of
// its group (the reference decodes it the same way through Skip).
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_bp_fetch(const uint64_t *__restrict__ block_offs,
                                                         const uint8_t *__restrict__ blocks,
                                                         const uint32_t *__restrict__ segs,
                                                         const uint32_t *__restrict__ rows, uint64_t n,
                                                         U *__restrict__ out) {
	const uint64_t k = (uint64_t)blockIdx.x * kWorkgroup + threadIdx.x;
	if (k >= n) return;
	const uint8_t *blk = blocks + block_offs[segs[k]];
	const uint32_t row = rows[k];
	const uint32_t r = row % kBpGroupRows;
	const BpHeader h = bp_header<U>(blk, row / kBpGroupRows);
	auto field = [&](uint32_t i) -> uint64_t {
		if (h.width == 0) return 0ull;
		const uint64_t bit = (uint64_t)i * h.width;
		uint64_t v = 0;
		const uint8_t *p = h.payload + (bit >> 3);
		const uint32_t sh = (uint32_t)(bit & 7);
		const uint32_t nbytes = (sh + h.width + 7) >> 3; // <= 9
		for (uint32_t b = 0; b < nbytes && b < 8; b++) v |= (uint64_t)p[b] << (8 * b);
		v >>= sh;
		if (nbytes > 8) v |= (uint64_t)p[8] << (64 - sh);
		return h.width >= 64 ? v : (v & ((1ull << h.width) - 1ull));
	};
	U res;
	if (h.mode == kBpConstant) {
		res = (U)h.frame;
	} else if (h.mode == kBpConstantDelta) {
		res = (U)((U)r * (U)h.extra + (U)h.frame);
	} else if (h.mode == kBpFor) {
		res = (U)(field(r) + h.frame);
	} else {
		U acc = (U)h.extra;
		for (uint32_t i = 0; i <= r; i++) acc = (U)(acc + (U)(field(i) + h.frame));
		res = acc;
	}
	out[k] = res;
}

// ---------------------------------------------------------------------------------------------
// Compress side.  The device produces per-group statistics and writes the group images; the mode decision and
// the sequential placement of groups into 256 KiB blocks (BitpackingState::Flush, ReserveSpace/FlushSegment:
// bitpacking.cpp:229-294,453-512) stay on the host — they are a few dozen scalar operations per 2048 rows.
// NULL rows: the reference leaves whatever its compression buffer held (indeterminate); here a NULL row
// contributes the value 0, i.e. the field (0 - frame_of_reference) mod 2^width.
// ---------------------------------------------------------------------------------------------
struct BpStats {
	uint64_t bmin, bmax;   // min / max of the valid rows in T's order, as bits ^ signbit(T); bmin > bmax if none valid
	uint64_t bdmin, bdmax; // min / max of d[i] = v[i] - v[i-1] (i >= 1) in T_S order, as bits ^ signbit(T_S)
	uint64_t v0;           // bits of the first row
	uint32_t rows, nvalid;
	uint32_t delta_overflow; // some v[i] - v[i-1] (v[-1] = 0) is not representable in T_S
	uint32_t pad;
};

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_bp_stats(const U *__restrict__ vals,
                                                         const uint64_t *__restrict__ validity, uint64_t n,
                                                         uint64_t sbit_t, BpStats *__restrict__ stats) {
	using S = typename std::make_signed<U>::type;
	constexpr int PER = kBpGroupRows / kWorkgroup;
	constexpr uint64_t sbit_s = 1ull << (8 * sizeof(U) - 1);
	__shared__ uint64_t red[6][kWorkgroup / 64];
	const uint64_t g0 = (uint64_t)blockIdx.x * kBpGroupRows;
	const uint32_t rows = (uint32_t)(n - g0 < (uint64_t)kBpGroupRows ? n - g0 : (uint64_t)kBpGroupRows);
	uint64_t bmin = ~0ull, bmax = 0, bdmin = ~0ull, bdmax = 0, ovf = 0, nvalid = 0;
	const uint32_t r0 = threadIdx.x * PER;
	U prev = 0;
	if (r0 > 0 && r0 - 1 < rows) prev = vals[g0 + r0 - 1];
#pragma unroll
	for (int k = 0; k < PER; k++) {
		const uint32_t r = r0 + k;
		if (r >= rows) break;
		const U v = vals[g0 + r];
		const uint64_t e = g0 + r;
		const bool valid = validity == nullptr || ((validity[e >> 6] >> (e & 63)) & 1ull);
		if (valid) {
			const uint64_t b = (uint64_t)v ^ sbit_t;
			bmin = b < bmin ? b : bmin;
			bmax = b > bmax ? b : bmax;
			nvalid++;
		}
		// (T_S)v[i] - (T_S)v[i-1] with the TrySubtractOperator overflow rule
		S d;
		const bool o = __builtin_sub_overflow((S)v, (S)prev, &d);
		ovf |= o ? 1ull : 0ull;
		if (r >= 1) {
			const uint64_t bd = ((uint64_t)(U)d) ^ sbit_s;
			bdmin = bd < bdmin ? bd : bdmin;
			bdmax = bd > bdmax ? bd : bdmax;
		}
		prev = v;
	}
	bmin = wave_min(bmin);
	bmax = wave_max(bmax);
	bdmin = wave_min(bdmin);
	bdmax = wave_max(bdmax);
	ovf = wave_max(ovf);
	nvalid = wave_sum(nvalid);
	if ((threadIdx.x & 63) == 0) {
		const uint32_t wv = threadIdx.x >> 6;
		red[0][wv] = bmin;
		red[1][wv] = bmax;
		red[2][wv] = bdmin;
		red[3][wv] = bdmax;
		red[4][wv] = ovf;
		red[5][wv] = nvalid;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		for (int i = 1; i < kWorkgroup / 64; i++) {
			bmin = red[0][i] < bmin ? red[0][i] : bmin;
			bmax = red[1][i] > bmax ? red[1][i] : bmax;
			bdmin = red[2][i] < bdmin ? red[2][i] : bdmin;
			bdmax = red[3][i] > bdmax ? red[3][i] : bdmax;
			ovf |= red[4][i];
			nvalid += red[5][i];
		}
		BpStats s;
		s.bmin = bmin;
		s.bmax = bmax;
		s.bdmin = bdmin;
		s.bdmax = bdmax;
		s.v0 = (uint64_t)vals[g0];
		s.rows = rows;
		s.nvalid = (uint32_t)nvalid;
		s.delta_overflow = (uint32_t)ovf;
		s.pad = 0;
		stats[blockIdx.x] = s;
	}
}

// What the host decided for one group (uploaded), consumed by k_bp_write.
struct BpWrite {
	uint64_t frame, extra; // FOR / DELTA_FOR frame of reference, CONSTANT's constant; delta / delta_offset
	uint64_t first;        // value of the block's first 8 bytes (offset past the first group's metadata entry)
	uint32_t seg;          // block index
	uint32_t data_off;     // byte offset of the group's data in its block
	uint32_t meta_off;     // byte offset of the group's metadata entry in its block
	uint32_t mode, width, rows;
	uint32_t first_of_segment;
	uint32_t pad;
};

template <typename U>
__device__ __forceinline__ void store_unaligned(uint8_t *p, U v) {
#pragma unroll
	for (int i = 0; i < (int)sizeof(U); i++) p[i] = (uint8_t)((uint64_t)v >> (8 * i));
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_bp_write(const BpWrite *__restrict__ recs, const U *__restrict__ vals,
                                                         const uint64_t *__restrict__ validity, uint64_t block_stride,
                                                         uint8_t *__restrict__ blocks) {
	__shared__ __attribute__((aligned(16))) U fld[kBpGroupRows];
	__shared__ uint32_t packed[kTileBytes / 4];
	const BpWrite r = recs[blockIdx.x];
	uint8_t *blk = blocks + (uint64_t)r.seg * block_stride;
	uint8_t *p = blk + r.data_off;
	const uint64_t g0 = (uint64_t)blockIdx.x * kBpGroupRows;
	if (threadIdx.x == 0) {
		// header (WriteData calls of the BitpackingWriter, bitpacking.cpp:374-437) + metadata entry (:447-451)
		store_unaligned<U>(p, (U)r.frame);
		if (r.mode == kBpConstantDelta) {
			store_unaligned<U>(p + sizeof(U), (U)r.extra);
		} else if (r.mode == kBpFor || r.mode == kBpDeltaFor) {
			store_unaligned<U>(p + sizeof(U), (U)r.width);
			if (r.mode == kBpDeltaFor) store_unaligned<U>(p + 2 * sizeof(U), (U)r.extra);
		}
		*reinterpret_cast<uint32_t *>(blk + r.meta_off) = r.data_off | (r.mode << 24);
		if (r.first_of_segment) *reinterpret_cast<uint64_t *>(blk) = r.first;
	}
	if ((r.mode != kBpFor && r.mode != kBpDeltaFor) || r.width == 0) return;
	uint8_t *payload = p + (r.mode == kBpDeltaFor ? 3 : 2) * sizeof(U);
	const uint32_t w = r.width;
	const uint32_t rows32 = (r.rows + 31u) & ~31u; // RoundUpToAlgorithmGroupSize: the tail group is zero-padded
	const U frame = (U)r.frame;
	for (uint32_t i = threadIdx.x; i < rows32; i += kWorkgroup) {
		U f = 0;
		if (i < r.rows) {
			const uint64_t e = g0 + i;
			if (r.mode == kBpFor) {
				const bool valid = validity == nullptr || ((validity[e >> 6] >> (e & 63)) & 1ull);
				const U x = valid ? vals[e] : (U)0;
				f = (U)(x - frame); // SubtractFrameOfReference
			} else {
				f = i == 0 ? (U)0 : (U)((U)(vals[e] - vals[e - 1]) - frame); // delta_buffer[0] = minimum_delta
			}
		}
		fld[i] = f;
	}
	__syncthreads();
	// every lane owns whole 32-bit words of the group's bit stream and gathers the fields overlapping them
	const uint32_t ndw = rows32 * w / 32;
	const uint64_t fmask = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
	for (uint32_t q = threadIdx.x; q < ndw; q += kWorkgroup) {
		const uint32_t bitlo = q * 32u;
		uint32_t i = bitlo / w;
		uint32_t last = (bitlo + 31u) / w;
		last = last < rows32 ? last : rows32 - 1;
		uint32_t acc = 0;
		for (; i <= last; i++) {
			const uint64_t v = (uint64_t)fld[i] & fmask; // fastpack masks every value to `width` bits
			const int32_t pos = (int32_t)(i * w) - (int32_t)bitlo;
			acc |= pos >= 0 ? (uint32_t)(v << pos) : (uint32_t)(v >> (-pos));
		}
		packed[q] = acc;
	}
	__syncthreads();
	// the payload starts wherever the previous group ended: sizeof(T)-aligned only
	const uintptr_t a = reinterpret_cast<uintptr_t>(payload);
	if ((a & 3) == 0) {
		uint32_t *dst = reinterpret_cast<uint32_t *>(payload);
		for (uint32_t q = threadIdx.x; q < ndw; q += kWorkgroup) dst[q] = packed[q];
	} else {
		const uint8_t *src = reinterpret_cast<const uint8_t *>(packed);
		for (uint32_t b = threadIdx.x; b < ndw * 4; b += kWorkgroup) payload[b] = src[b];
	}
}
