// adac_select_gather.inl — materialise only the rows a selection bitmap keeps (included by adac_kernels.hip).
//
// The counterpart of DuckDB's scan-with-selection (ColumnSegment::FilterSelection narrows a SelectionVector,
// column_segment.cpp:575-844, and the scan then copies the surviving rows): after adac_scan_select_between the
// engine wants the VALUES of the selected rows, densely, in row order.  Three steps, all on the device:
//   k_tile_popc   rows selected per tile (one wave per tile, popcount over the tile's bitmap window);
//   k_scan_*      exclusive prefix over the tile table (tile order = segment order = row order) -> each tile's
//                 first output slot, and the grand total;
//   k_gather      per tile: per-chunk popcounts -> workgroup prefix (DPP wave scan + 4 totals through LDS) -> the
//                 tile is decoded once and every selected row is stored at its slot.

// rows selected in each tile: one wave per tile, four tiles per workgroup
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_tile_popc(const adac_segment_desc *__restrict__ descs,
                                                          const TileRef *__restrict__ tiles, uint32_t ntiles,
                                                          const uint64_t *__restrict__ bitmap,
                                                          uint32_t *__restrict__ tile_cnt) {
	constexpr uint32_t TILE = kTileBytes / sizeof(U);
	const uint32_t t = blockIdx.x * (kWorkgroup / 64) + (threadIdx.x >> 6);
	if (t >= ntiles) return;
	const TileRef r = tiles[t];
	const adac_segment_desc d = descs[r.seg];
	const uint32_t left = d.count - r.first;
	const uint32_t n = left < TILE ? left : TILE;
	const uint64_t e0 = d.val_off + r.first;
	uint32_t c = 0;
	for (uint32_t i = (threadIdx.x & 63u) * 32u; i < n; i += 64u * 32u) {
		const uint32_t rest = n - i;
		const uint32_t m = rest >= 32u ? 0xffffffffu : ((1u << rest) - 1u);
		c += (uint32_t)__popc(validity_window_pair(bitmap, e0 + i, rest >= 32u ? 32u : rest) & m);
	}
	c = wave_inclusive_sum<uint32_t>(c);
	if ((threadIdx.x & 63u) == 63u) tile_cnt[t] = c;
}

// exclusive prefix of u32 counts into u64 offsets, three small kernels: 1024-entry blocks, their totals, the fix-up
constexpr int kScanBlock = 1024;
__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *total) {
	__shared__ uint64_t wave_tot[kScanBlock / 64];
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t incl = wave_inclusive_sum<uint64_t>(v);
	if (lane == 63) wave_tot[wave] = incl;
	__syncthreads();
	uint64_t before = 0, all = 0;
#pragma unroll
	for (uint32_t w = 0; w < kScanBlock / 64; w++) {
		const uint64_t tw = wave_tot[w];
		if (w < wave) before += tw;
		all += tw;
	}
	*total = all;
	return before + incl - v;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_blocks(const uint32_t *__restrict__ cnt, uint64_t n,
                                                            uint64_t *__restrict__ offs,
                                                            uint64_t *__restrict__ block_tot) {
	const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
	uint64_t total;
	const uint64_t ex = block_exclusive_scan(i < n ? (uint64_t)cnt[i] : 0ull, &total);
	if (i < n) offs[i] = ex;
	if (threadIdx.x == 0) block_tot[blockIdx.x] = total;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_totals(uint64_t *__restrict__ block_tot, uint64_t nblocks,
                                                            uint64_t *__restrict__ grand_total) {
	uint64_t carry = 0;
	for (uint64_t base = 0; base < nblocks; base += kScanBlock) { // one workgroup walks the block totals
		const uint64_t i = base + threadIdx.x;
		uint64_t total;
		const uint64_t ex = block_exclusive_scan(i < nblocks ? block_tot[i] : 0ull, &total);
		__syncthreads();
		if (i < nblocks) block_tot[i] = carry + ex;
		carry += total;
		__syncthreads();
	}
	if (threadIdx.x == 0) *grand_total = carry;
}
__global__ __launch_bounds__(kScanBlock) void k_scan_fixup(uint64_t *__restrict__ offs, uint64_t n,
                                                           const uint64_t *__restrict__ block_tot) {
	const uint64_t i = (uint64_t)blockIdx.x * kScanBlock + threadIdx.x;
	if (i < n) offs[i] += block_tot[blockIdx.x];
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_gather(const adac_segment_desc *__restrict__ descs,
                                                       const TileRef *__restrict__ tiles,
                                                       const uint64_t *__restrict__ words,
                                                       const uint64_t *__restrict__ bitmap,
                                                       const uint64_t *__restrict__ tile_off, U *__restrict__ out,
                                                       uint64_t *__restrict__ out_ids) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr int K = 16 / (int)sizeof(U);
	constexpr int CHUNKS = TILE / K;                 // 1024 chunks of K rows
	constexpr int PER_LANE = CHUNKS / kWorkgroup;    // 4 consecutive chunks per lane in the prefix pass
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ uint16_t chunk_off[CHUNKS];           // rows selected in the tile before each chunk (<= 16384)
	__shared__ uint32_t wave_tot[kWorkgroup / 64];
	const TileCtx t = resolve_tile<TILE>(descs, tiles);
	// pass 1: per-chunk popcounts straight from the bitmap, exclusive prefix over the tile
	uint32_t cnt[PER_LANE], run = 0;
#pragma unroll
	for (int k = 0; k < PER_LANE; k++) {
		const uint32_t row = (threadIdx.x * PER_LANE + k) * K;
		uint32_t bits = 0;
		if (row < t.n) {
			const uint32_t rest = t.n - row;
			const uint32_t m = rest >= (uint32_t)K ? ((1u << K) - 1u) : ((1u << rest) - 1u);
			bits = validity_window(bitmap, t.elem0 + row, rest >= (uint32_t)K ? (uint32_t)K : rest) & m;
		}
		cnt[k] = run;
		run += (uint32_t)__popc(bits);
	}
	const uint32_t incl = wave_inclusive_sum<uint32_t>(run);
	if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = incl;
	const uint32_t bit0 = stage_packed(words + t.d.word_off, t.first, t.n, t.d.width, lds);
	__syncthreads();
	uint32_t before = incl - run, total = 0;
#pragma unroll
	for (uint32_t w = 0; w < kWorkgroup / 64; w++) {
		const uint32_t tw = wave_tot[w];
		if (w < (threadIdx.x >> 6)) before += tw;
		total += tw;
	}
	if (total == 0) return; // nothing selected in this tile (workgroup-uniform): no decode at all
#pragma unroll
	for (int k = 0; k < PER_LANE; k++) chunk_off[threadIdx.x * PER_LANE + k] = (uint16_t)(before + cnt[k]);
	__syncthreads();
	// pass 2: decode the tile, store the selected rows at their slots
	const uint64_t slot0 = tile_off[blockIdx.x];
	auto sink = [&](int32_t base, const U *v, bool full) { // align 0: base = chunk * K
		const uint32_t rest = t.n - (uint32_t)base;
		const uint32_t rows_here = full || rest >= (uint32_t)K ? (uint32_t)K : rest;
		const uint32_t bits = validity_window(bitmap, t.elem0 + (uint32_t)base, rows_here) & ((1u << rows_here) - 1u);
		uint64_t slot = slot0 + chunk_off[(uint32_t)base / K];
#pragma unroll
		for (int j = 0; j < K; j++) {
			if ((bits >> j) & 1u) {
				// (plain stores: these scattered 8-byte stores rely on the L2 combining them into lines — non-temporal
				// they are 15 % slower, profiles/r02_nontemporal_stores.json)
				out[slot] = v[j];
				if (out_ids) out_ids[slot] = t.elem0 + (uint32_t)base + j;
				slot++;
			}
		}
	};
	decode_tile<U>(reinterpret_cast<const uint32_t *>(lds), bit0, t.d.width, effective_add(t.d), t.n, 0u, sink);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_gather_c — the same gather with WAVE-LEVEL COMPACTION (round 3).  k_gather stores every selected row with a store
// of its own (two per row with the ids): at 50 % selectivity four half-empty 8-byte store instructions per chunk where
// k_unpack issues one full dwordx4, and the L2 merges the partial lines: 335 us for the bytes k_unpack moves in 199.
// Here a wave compacts the rows it decodes in one round — 64 consecutive chunks, i.e. ONE contiguous run of rows whose
// selected rows are one contiguous run of output slots — in an LDS buffer of its own and writes them out densely, a
// 16-byte unit per lane, the buffer laid out at the output's own 16-byte phase (first / last unit: element stores).
// No workgroup barrier after the staging one and no chunk-offset table: every wave derives the sixteen run offsets of
// the tile itself (a lane per bitmap word or group of words, one DPP scan; the four waves repeat 256 - 2048 bytes of
// bitmap reads), the position inside a run is a second DPP scan per round.  LDS operations of one wave execute in
// order, so the write -> read -> next round's write sequence on the wave's buffer needs no barrier.
// ---------------------------------------------------------------------------------------------------------------------
// Selection bits starting at element e (bit j = element e + j) from the bitmap read as 32-bit words: every load is
// UNCONDITIONAL, its index clamped to the bitmap's last dword (never over-read).  The form with the second word under a
// condition (validity_window) compiled to a branch per call and the five windows of a lane — four rounds and the
// tile-wide prefix — went out one round trip after the other.
// (dword indices in 32 bits — the launcher takes this kernel for bitmaps of less than 2^32 dwords only — so a clamp is
// one v_min_u32; with 64-bit indices each was a compare and two selects and the 1-byte types, twelve loads a lane in
// the prefix pass, lost 9 %)
__device__ __forceinline__ uint32_t bitmap_bits32(const uint32_t *__restrict__ bm32, uint32_t last_dword, uint64_t e) {
	const uint32_t dw = (uint32_t)(e >> 5);
	const uint32_t lo = bm32[dw < last_dword ? dw : last_dword];
	const uint32_t hi = bm32[dw + 1u < last_dword ? dw + 1u : last_dword];
	return __builtin_amdgcn_alignbit(hi, lo, (uint32_t)e & 31u);
}
// popcount of the first `have` (0 .. 64 * WORDS) of the 64 * WORDS bits starting at element e: 2 * WORDS + 1 loads
template <int WORDS>
__device__ __forceinline__ uint32_t bitmap_popc(const uint32_t *__restrict__ bm32, uint32_t last_dword, uint64_t e,
                                                uint32_t have) {
	const uint32_t dw = (uint32_t)(e >> 5), sh = (uint32_t)e & 31u;
	uint32_t d[2 * WORDS + 1];
#pragma unroll
	for (int i = 0; i <= 2 * WORDS; i++) d[i] = bm32[dw + (uint32_t)i < last_dword ? dw + (uint32_t)i : last_dword];
	uint32_t c = 0;
#pragma unroll
	for (int i = 0; i < 2 * WORDS; i++) {
		const uint32_t w = __builtin_amdgcn_alignbit(d[i + 1], d[i], sh);
		const uint32_t n = have > 32u * (uint32_t)i ? have - 32u * (uint32_t)i : 0u;
		c += (uint32_t)__popc(w & (n >= 32u ? 0xffffffffu : ((1u << n) - 1u)));
	}
	return c;
}

template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_gather_c(const adac_segment_desc *__restrict__ descs,
                                                         const TileRef *__restrict__ tiles,
                                                         const TileRec *__restrict__ recs,
                                                         const uint64_t *__restrict__ words,
                                                         const uint64_t *__restrict__ bitmap, uint32_t last_dword,
                                                         const uint32_t *__restrict__ tile_cnt,
                                                         const uint64_t *__restrict__ tile_off, U *__restrict__ out,
                                                         uint64_t *__restrict__ out_ids, int nt) {
	constexpr int TILE = kTileBytes / (int)sizeof(U);
	constexpr int K = 16 / (int)sizeof(U);
	constexpr int RUN = 64 * K;                    // rows one wave decodes per round: 1 KiB of values
	constexpr int WORDS = TILE / 64;               // bitmap words of a tile: 32 / 64 / 128 / 256
	constexpr int WPL = WORDS > 64 ? WORDS / 64 : 1; // ... per lane in the offsets pass
	constexpr int LANES_PER_RUN = (RUN / 64) / WPL;  // lanes whose words make up one run: 2 / 4 / 4 / 4
	__shared__ uint4 lds[kTileBytes / 16 + 2];
	__shared__ uint4 wbuf_all[kWorkgroup / 64][65];                 // per wave: 64 units + one for the phase
	__shared__ uint32_t wid_all[kWorkgroup / 64][RUN / 2 + 1];      // per wave: row numbers inside the tile (u16 pairs)
	const uint64_t slot0 = tile_off[blockIdx.x]; // (asked for first: a round trip nothing else waits behind)
	// A tile without a selected row leaves before it reads a packed byte or a bitmap word: its count (k_tile_popc's)
	// is asked for together with the tile entry and is there a hop before the descriptor, which the staging loads
	// need anyway — the test costs the other tiles nothing.  A clustered selection (a range of a sorted column) pays
	// for the tiles it touches only.
	const uint32_t selected_here = tile_cnt[blockIdx.x];
	// the tile's geometry: from its expanded record (one hop) or through tile entry -> descriptor (two)
	struct {
		uint64_t word_off, elem0, add;
		uint32_t first, n, width;
	} t;
	if (recs) { // uniform
		const TileRec r = load_tile_rec(recs + blockIdx.x);
		t.word_off = r.word_off, t.elem0 = r.elem0, t.add = r.add, t.first = r.first, t.n = r.n, t.width = r.width;
	} else {
		const TileCtx c = resolve_tile<TILE>(descs, tiles);
		t.word_off = c.d.word_off, t.elem0 = c.elem0, t.add = effective_add(c.d), t.first = c.first, t.n = c.n, t.width = c.d.width;
	}
	if (selected_here == 0u) return;
	const uint32_t bit0 = stage_packed(words + t.word_off, t.first, t.n, t.width, lds);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t *__restrict__ bm32 = reinterpret_cast<const uint32_t *>(bitmap);
	// selected rows before each run of the tile: inclusive scan over the lanes' words
	uint32_t mine = 0;
	{
		const uint32_t row = lane * (uint32_t)WPL * 64u; // the lane's words are consecutive: one window of WPL words
		const uint32_t rest = row < t.n ? t.n - row : 0u;
		mine = bitmap_popc<WPL>(bm32, last_dword, t.elem0 + row, rest < 64u * WPL ? rest : 64u * WPL);
	}
	const uint32_t run_incl = wave_inclusive_sum<uint32_t>(mine);
	const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)run_incl, 63);
	// this lane's chunks (chunk r * 256 + thread of round r, the map of both decode forms): their selection bits and
	// the rows selected before them inside the wave's run — bitmap loads and DPP chains of all four rounds in flight
	// together and behind the staging loads, nothing but LDS work is left between the barrier and the stores
	static_assert(TILE / (kWorkgroup * K) == 4, "four decode rounds per tile");
	// (named scalars captured by value, not arrays: indexed by the round inside the sink an array went to scratch memory)
	uint32_t pk0, pk1, pk2, pk3;     // bits | rows before << 16
	uint32_t tot0, tot1, tot2, tot3; // selected rows of the run (uniform)
#define ADAC_ROUND_BITS(PK, R)                                                                                         \
	{                                                                                                                  \
		const uint32_t base = ((uint32_t)(R) * kWorkgroup + threadIdx.x) * (uint32_t)K;                                \
		const uint32_t rows_here = base < t.n ? (t.n - base < (uint32_t)K ? t.n - base : (uint32_t)K) : 0u;            \
		PK = bitmap_bits32(bm32, last_dword, t.elem0 + base) & ((1u << rows_here) - 1u);                              \
	}
#define ADAC_ROUND_SCAN(PK, TOT)                                                                                       \
	{                                                                                                                  \
		const uint32_t cnt = (uint32_t)__popc(PK);                                                                     \
		const uint32_t incl = wave_inclusive_sum<uint32_t>(cnt);                                                       \
		TOT = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);                                                      \
		PK |= (incl - cnt) << 16;                                                                                      \
	}
	ADAC_ROUND_BITS(pk0, 0) ADAC_ROUND_BITS(pk1, 1) ADAC_ROUND_BITS(pk2, 2) ADAC_ROUND_BITS(pk3, 3)
	ADAC_ROUND_SCAN(pk0, tot0) ADAC_ROUND_SCAN(pk1, tot1) ADAC_ROUND_SCAN(pk2, tot2) ADAC_ROUND_SCAN(pk3, tot3)
#undef ADAC_ROUND_BITS
#undef ADAC_ROUND_SCAN
	__syncthreads();
	if (total == 0) return; // nothing selected in this tile (uniform): no decode at all
	U *const wb = reinterpret_cast<U *>(wbuf_all[wave]);
	uint16_t *const wi = reinterpret_cast<uint16_t *>(wid_all[wave]);
	const uint64_t elem0 = t.elem0;
	uint4 *const wunits = wbuf_all[wave];
	const uint32_t *const widw = wid_all[wave];
	auto sink = [=](int32_t base, const U *v, bool full) __attribute__((always_inline)) { // align 0: base = chunk * K; all 64 lanes of the wave are here
		const uint32_t ubase = (uint32_t)base;
		const uint32_t round = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ubase / (uint32_t)(K * kWorkgroup))); // uniform
		// (the four candidates pass through an empty asm first: selected straight from the closure the compiler turned
		// the chain into ONE load at a computed address and kept the whole closure in scratch memory)
		uint32_t a0 = pk0, a1 = pk1, a2 = pk2, a3 = pk3, b0 = tot0, b1 = tot1, b2 = tot2, b3 = tot3;
		asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3));
		uint32_t mine_pk = a0, tot = b0;
		if (round == 1u) mine_pk = a1, tot = b1;
		if (round == 2u) mine_pk = a2, tot = b2;
		if (round == 3u) mine_pk = a3, tot = b3;
		if (tot == 0u) return; // uniform: nothing selected in this run
		const uint32_t bits = mine_pk & 0xffffu;
		const uint32_t run = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ubase / (uint32_t)K)) >> 6; // uniform
		const uint32_t before = run ? (uint32_t)__builtin_amdgcn_readlane((int)run_incl, (int)(run * LANES_PER_RUN - 1u)) : 0u;
		const uint64_t slot = slot0 + before;
		U *const g_out = out + slot;
		if constexpr (sizeof(U) >= 4) {
			// 4- and 8-byte types: one element per lane and store (a wave writes 256 / 512 contiguous bytes), no phase
			// and no edge cases — the unit form below executed 490 vector instructions per wave against k_unpack's 100
			// and kept the vector unit 48 % busy (SQ counters, profiles/r03_gather_compaction.json)
			uint32_t p = mine_pk >> 16;
#pragma unroll
			for (int j = 0; j < K; j++) {
				if ((bits >> j) & 1u) {
					wb[p] = v[j];
					if (out_ids) wi[p] = (uint16_t)(ubase + (uint32_t)j);
					p++;
				}
			}
			__builtin_amdgcn_wave_barrier();
			for (uint32_t i = lane; i < tot; i += 64u) {
				if (nt) {
					__builtin_nontemporal_store(wb[i], g_out + i);
				} else {
					g_out[i] = wb[i];
				}
			}
			if (out_ids) {
				uint64_t *const g_ids = out_ids + slot;
				for (uint32_t i = lane; i < tot; i += 64u) {
					const uint64_t id = elem0 + wi[i];
					if (nt) {
						__builtin_nontemporal_store(id, g_ids + i);
					} else {
						g_ids[i] = id;
					}
				}
			}
			__builtin_amdgcn_wave_barrier();
			return;
		}
		const uint32_t pe = (uint32_t)((reinterpret_cast<uintptr_t>(g_out) & 15u) / sizeof(U)); // uniform: phase, in elements
		const uint32_t pid = (uint32_t)((reinterpret_cast<uintptr_t>(out_ids + slot) >> 3) & 1u);
		// compaction: the wave's selected rows, in row order, at the output's 16-byte phase
		uint32_t p = mine_pk >> 16;
#pragma unroll
		for (int j = 0; j < K; j++) {
			if ((bits >> j) & 1u) {
				wb[pe + p] = v[j];
				if (out_ids) wi[pid + p] = (uint16_t)(ubase + (uint32_t)j);
				p++;
			}
		}
		__builtin_amdgcn_wave_barrier();
		// values: unit u = wb[u K .. u K + K) = elements u K - pe ... of the run
		const uint32_t nunits = (pe + tot + (uint32_t)K - 1u) / (uint32_t)K; // <= 65
		for (uint32_t u = lane; u < nunits; u += 64u) {
			const uint4 q = wunits[u];
			const int32_t e0 = (int32_t)(u * (uint32_t)K) - (int32_t)pe;
			U *const g = g_out + e0;
			if (e0 >= 0 && (uint32_t)e0 + (uint32_t)K <= tot) {
				typedef uint32_t v4u __attribute__((ext_vector_type(4)));
				const v4u qq = {q.x, q.y, q.z, q.w};
				if (nt) {
					__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(g));
				} else {
					*reinterpret_cast<v4u *>(g) = qq;
				}
			} else {
				U e[K];
				__builtin_memcpy(e, &q, 16);
#pragma unroll
				for (int j = 0; j < K; j++) {
					if ((uint32_t)(e0 + j) < tot) g[j] = e[j];
				}
			}
		}
		if (out_ids) { // ids: unit u = the ids of elements 2 u - pid, 2 u + 1 - pid
			uint64_t *const g_ids = out_ids + slot;
			const uint32_t nid = (pid + tot + 1u) >> 1;
			for (uint32_t u = lane; u < nid; u += 64u) {
				const uint32_t two = widw[u];
				const int32_t e0 = (int32_t)(2u * u) - (int32_t)pid;
				const uint64_t id0 = elem0 + (two & 0xffffu), id1 = elem0 + (two >> 16);
				const bool v0 = e0 >= 0, v1 = (uint32_t)(e0 + 1) < tot;
				if (v0 && v1) {
					typedef uint32_t v4u __attribute__((ext_vector_type(4)));
					const v4u qq = {(uint32_t)id0, (uint32_t)(id0 >> 32), (uint32_t)id1, (uint32_t)(id1 >> 32)};
					if (nt) {
						__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(g_ids + e0));
					} else {
						*reinterpret_cast<v4u *>(g_ids + e0) = qq;
					}
				} else if (v0) {
					g_ids[e0] = id0;
				} else if (v1) {
					g_ids[e0 + 1] = id1;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
	};
	decode_tile<U, true>(reinterpret_cast<const uint32_t *>(lds), bit0, t.width, t.add, t.n, 0u, sink);
}
