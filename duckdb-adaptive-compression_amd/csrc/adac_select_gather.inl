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
		c += (uint32_t)__popc(validity_window(bitmap, e0 + i, rest >= 32u ? 32u : rest) & m);
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
