// adac_group_sum.inl — Q1-shaped scan: SUM(value), COUNT(*) GROUP BY key over TWO packed columns of the same table.
// Included into adac_kernels.hip inside namespace adac::{anonymous}.
//
// The reference's config 3 runs TPC-H Q1 over lineitem (benchmark log TPCH_runtime.txt:2-6; SURVEY.md §8d C3 "Q1 =
// group-by sum"): the engine decodes both columns vector by vector (SuccinctScanPartial, succinct.cpp:123-144) and
// feeds a hash aggregate.  Here nothing is materialised: a workgroup walks tiles of the VALUE column's layout, stages
// the packed bits of the same rows of both columns in LDS, and every thread adds its rows into LDS bins:
//   * up to kGroupPrivateBins bins (Q1: 4 - 6 groups + the overflow bin): SIXTEEN bin sets per wave, bins[b][set] with
//     set = lane & 15 — at most four lanes collide on an LDS add whatever the keys are.  (One set per THREAD never
//     collides but costs 24 KiB per workgroup: three workgroups per CU, and the kernel — a chain of LDS and global
//     round trips per stage — ran 0.107 - 0.22 ms per 60 M-row column; with 6 KiB of bins seven workgroups fit and
//     it runs 0.085 - 0.14 ms: profiles/r02_q1_packed.json);
//   * up to kGroupMaxBins bins: one bin set per workgroup, LDS atomics (collisions serialise when few keys dominate).
// A workgroup is persistent (grid-stride over the tiles) and carries its bins across tiles; at the end it writes ONE
// partial {sum, count} per bin, and k_group_final adds the partials — no global atomics on a handful of addresses.
// Semantics: key = the key column's value as an unsigned number of its own width; rows whose key >= ngroups land in
// bin `ngroups`.  SUM = the values widened to 64 bits by the value type's signedness, mod 2^64 (adac_scan_sum's rule).

constexpr uint32_t kGroupStageBytes = 3584;   // packed bytes of one column per stage
constexpr uint32_t kGroupChunksPerThread = (kGroupStageBytes / 16 + 1 + kWorkgroup - 1) / kWorkgroup;
constexpr uint32_t kGroupCopies = 16;         // bin sets per wave (lane & 15 picks one): 4 lanes share a set (8 sets: slower)
constexpr uint32_t kGroupPrivateBins = 8;     // bins held per thread
constexpr uint32_t kGroupMaxBins = 257;       // 256 groups + overflow
constexpr uint32_t kGroupMaxWorkgroups = 2048; // capacity of the partial buffer; the launch uses 7 per CU (21 KiB of LDS each)

// the w-bit field at `bit` of a staged image, any w in 1..64: mlo / mhi = the low / high dword of the width mask.
// Branch-free on purpose (three dwords are always read): a branch on the width ends the basic block, the compiler
// then waits for every field read before it issues the next one, and the row loop becomes a chain of LDS round trips
// (measured: 0.22 ms for 60 M rows at every width, 64 % of the wave cycles waiting)
__device__ __forceinline__ uint64_t staged_field(const uint32_t *lds32, uint32_t bit, uint32_t mlo, uint32_t mhi) {
	const uint32_t dw = bit >> 5, sh = bit & 31u;
	const uint32_t a0 = lds32[dw], a1 = lds32[dw + 1], a2 = lds32[dw + 2];
	const uint32_t lo = __builtin_amdgcn_alignbit(a1, a0, sh) & mlo;
	const uint32_t hi = __builtin_amdgcn_alignbit(a2, a1, sh) & mhi;
	return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t staged_field32(const uint32_t *lds32, uint32_t bit, uint32_t mlo) {
	const uint32_t dw = bit >> 5, sh = bit & 31u;
	return __builtin_amdgcn_alignbit(lds32[dw + 1], lds32[dw], sh) & mlo;
}

// a segment descriptor through two 16-byte scalar loads (the struct copy reads width / flags with a 2-byte VECTOR
// load, which then waits behind every data load in flight)
__device__ __forceinline__ adac_segment_desc load_desc_scalar(const adac_segment_desc *__restrict__ descs, uint32_t seg) {
	const uint4 *p = reinterpret_cast<const uint4 *>(descs + seg);
	const uint4 a = p[0], b = p[1];
	adac_segment_desc d;
	d.word_off = ((uint64_t)a.y << 32) | a.x;
	d.val_off = ((uint64_t)a.w << 32) | a.z;
	d.min = ((uint64_t)b.y << 32) | b.x;
	d.count = b.z;
	d.width = (uint8_t)(b.w & 0xffu);
	d.flags = (uint8_t)((b.w >> 8) & 0xffu);
	d.reserved = (uint16_t)(b.w >> 16);
	return d;
}

struct GroupSumTypes {
	uint64_t v_tmask, v_sbit; // value type: all-ones mask of its width, its sign bit (0 for unsigned types)
	uint64_t k_tmask;         // key type: all-ones mask of its width
	uint32_t v_tile_rows;     // rows per tile of the value layout
	uint32_t wide_only;       // diagnostic: never take the 32-bit fast path
};

// one stage of work: rows [first, first + m) of a segment, both columns (all wave-uniform)
struct GroupStage {
	const uint4 *vsrc, *ksrc; // the 16-byte chunks holding the first bit of the rows, value and key column
	uint64_t vadd, kadd;
	uint32_t vbit0, kbit0, vchunks, kchunks, wv, wk, m;
};

__global__ __launch_bounds__(kWorkgroup) void k_group_sum(const adac_segment_desc *__restrict__ vdescs,
                                                          const TileRef *__restrict__ vtiles, uint32_t ntiles,
                                                          const uint64_t *__restrict__ vwords,
                                                          const adac_segment_desc *__restrict__ kdescs,
                                                          const uint64_t *__restrict__ kwords, GroupSumTypes ty,
                                                          uint32_t ngroups, unsigned long long *__restrict__ partial) {
	const bool g_narrow_ok = ty.wide_only == 0u; // A/B knob "group_sum_wide"
	// two stage buffers per column: the next stage's chunks are loaded (into registers) before the current stage is
	// aggregated and written to the other buffer after it, so a global round trip is always in flight
	__shared__ uint4 vstage[2][kGroupStageBytes / 16 + 2];
	__shared__ uint4 kstage[2][kGroupStageBytes / 16 + 2];
	// bins: up to kGroupPrivateBins bins in kGroupCopies sets per wave, or one set of up to kGroupMaxBins bins
	constexpr uint32_t kSets = (kWorkgroup / 64) * kGroupCopies;
	constexpr uint32_t kBinSlots = kGroupPrivateBins * kSets > kGroupMaxBins ? kGroupPrivateBins * kSets : kGroupMaxBins;
	__shared__ unsigned long long bsum[kBinSlots];
	__shared__ uint32_t bcnt[kBinSlots];
	const uint32_t my_set = (threadIdx.x >> 6) * kGroupCopies + (threadIdx.x & (kGroupCopies - 1u));
	const uint32_t nbins = ngroups + 1u;
	const bool priv = nbins <= kGroupPrivateBins; // uniform
	const uint32_t tid = threadIdx.x;
	for (uint32_t i = tid; i < kBinSlots; i += kWorkgroup) {
		bsum[i] = 0ull;
		bcnt[i] = 0u;
	}
	// The workgroup's tiles are blockIdx.x, + gridDim.x, ...  A tile's metadata is two dependent hops (tile -> segment
	// descriptors): the tile reference is fetched TWO tiles ahead and the descriptors ONE tile ahead, so that neither
	// round trip is waited for when a tile starts (one stage per tile at narrow widths: the chain was 5 of the 7 us a
	// tile cost).  Inside a tile as many rows per stage as fit kGroupStageBytes at the wider of the two widths.
	struct TileMeta {
		TileRef r;
		adac_segment_desc vd, kd;
		bool valid;
	};
	const uint32_t G = gridDim.x;
	auto fetch_ref = [&](uint32_t tile) { return vtiles[tile < ntiles ? tile : 0u]; };
	auto resolve = [&](TileRef r, uint32_t tile) {
		TileMeta m;
		m.r = r;
		m.vd = load_desc_scalar(vdescs, r.seg);
		m.kd = load_desc_scalar(kdescs, r.seg);
		m.valid = tile < ntiles;
		return m;
	};
	uint32_t t = blockIdx.x, done = 0;
	TileMeta mcur = resolve(fetch_ref(t), t);
	TileMeta mnxt = resolve(fetch_ref(t + G), t + G);
	TileRef rnn = fetch_ref(t + 2u * G);
	auto next_stage = [&](GroupStage &g) -> bool {
		if (mcur.valid) {
			const uint32_t left = mcur.vd.count - mcur.r.first;
			const uint32_t n = left < ty.v_tile_rows ? left : ty.v_tile_rows;
			if (done >= n) { // uniform: on to the next tile
				mcur = mnxt;
				t += G;
				mnxt = resolve(rnn, t + G);
				rnn = fetch_ref(t + 2u * G);
				done = 0;
			}
		}
		if (!mcur.valid) return false;
		const uint32_t left = mcur.vd.count - mcur.r.first;
		const uint32_t n = left < ty.v_tile_rows ? left : ty.v_tile_rows;
		g.wv = mcur.vd.width;
		g.wk = mcur.kd.width;
		const uint32_t wmax = g.wv > g.wk ? g.wv : g.wk;
		uint32_t per_stage = ((kGroupStageBytes * 8u - 256u) / wmax) & ~(uint32_t)(kWorkgroup - 1);
		per_stage = per_stage < (uint32_t)kWorkgroup ? (uint32_t)kWorkgroup : per_stage;
		g.m = n - done < per_stage ? n - done : per_stage;
		const uint64_t vpos = (uint64_t)(mcur.r.first + done) * g.wv, kpos = (uint64_t)(mcur.r.first + done) * g.wk;
		g.vsrc = reinterpret_cast<const uint4 *>(vwords + mcur.vd.word_off) + (vpos >> 7);
		g.ksrc = reinterpret_cast<const uint4 *>(kwords + mcur.kd.word_off) + (kpos >> 7);
		g.vbit0 = (uint32_t)(vpos & 127);
		g.kbit0 = (uint32_t)(kpos & 127);
		g.vchunks = (g.vbit0 + g.m * g.wv + 127u) >> 7; // <= kGroupStageBytes / 16 + 1 <= two per thread, >= 1
		g.kchunks = (g.kbit0 + g.m * g.wk + 127u) >> 7;
		g.vadd = effective_add(mcur.vd);
		g.kadd = effective_add(mcur.kd);
		done += g.m;
		return true;
	};
	GroupStage cur, nxt;
	bool have = next_stage(cur); // uniform
	uint4 vq[kGroupChunksPerThread], kq[kGroupChunksPerThread];
	if (have) {
#pragma unroll
		for (uint32_t h = 0; h < kGroupChunksPerThread; h++) { // chunks <= kGroupStageBytes / 16 + 1: inside the buffer
			const uint32_t c = tid + h * kWorkgroup;
			if (c < cur.vchunks) vstage[0][c] = cur.vsrc[c];
			if (c < cur.kchunks) kstage[0][c] = cur.ksrc[c];
		}
	}
	__syncthreads();
	uint32_t buf = 0;
	while (have) {
		const bool more = next_stage(nxt);
		if (more) { // in flight while this stage is aggregated.  UNCONDITIONAL loads (index clamped into the stage): a
			// load under a per-lane condition gets a wait of its own and the round trips run one after the other
#pragma unroll
			for (uint32_t h = 0; h < kGroupChunksPerThread; h++) {
				const uint32_t c = tid + h * kWorkgroup;
				vq[h] = nxt.vsrc[c < nxt.vchunks ? c : nxt.vchunks - 1u];
				kq[h] = nxt.ksrc[c < nxt.kchunks ? c : nxt.kchunks - 1u];
			}
		}
		const uint32_t *v32 = reinterpret_cast<const uint32_t *>(vstage[buf]);
		const uint32_t *k32 = reinterpret_cast<const uint32_t *>(kstage[buf]);
		// four rows per thread and round: the eight field reads are issued together, then the eight LDS adds (one row
		// at a time the loop was a chain of LDS round trips: 0.22 ms for 60 M rows whatever the widths)
		const uint32_t vmlo = cur.wv >= 32u ? 0xffffffffu : mask32(cur.wv), vmhi = cur.wv > 32u ? mask32(cur.wv - 32u) : 0u;
		const uint32_t kmlo = cur.wk >= 32u ? 0xffffffffu : mask32(cur.wk), kmhi = cur.wk > 32u ? mask32(cur.wk - 32u) : 0u;
		const bool narrow = g_narrow_ok && cur.wv <= 32u && cur.wk <= 32u && ty.v_tmask <= 0xffffffffull && ty.k_tmask <= 0xffffffffull;
		for (uint32_t row0 = tid; row0 < cur.m; row0 += 4u * kWorkgroup) {
			if (narrow) { // uniform: both fields and both types fit 32 bits — half the arithmetic
				uint32_t v[4], key[4];
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint32_t row = row0 + (uint32_t)u * kWorkgroup;
					const uint32_t rr = row < cur.m ? row : row0; // clamped: the read stays inside the stage
					v[u] = staged_field32(v32, cur.vbit0 + rr * cur.wv, vmlo);
					key[u] = staged_field32(k32, cur.kbit0 + rr * cur.wk, kmlo);
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					const uint32_t row = row0 + (uint32_t)u * kWorkgroup;
					uint32_t x = (v[u] + (uint32_t)cur.vadd) & (uint32_t)ty.v_tmask;
					x = (x ^ (uint32_t)ty.v_sbit) - (uint32_t)ty.v_sbit; // sign-extends to 32 bits ...
					const uint64_t x64 = ty.v_sbit ? (uint64_t)(int64_t)(int32_t)x : (uint64_t)x; // ... and on to 64
					const uint32_t k = (key[u] + (uint32_t)cur.kadd) & (uint32_t)ty.k_tmask;
					const uint32_t bin = k < ngroups ? k : ngroups;
					const uint32_t slot = priv ? bin * kSets + my_set : bin;
					if (row < cur.m) {
						atomicAdd(&bsum[slot], (unsigned long long)x64);
						atomicAdd(&bcnt[slot], 1u);
					}
				}
				continue;
			}
			uint64_t v[4], key[4];
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t row = row0 + (uint32_t)u * kWorkgroup;
				const uint32_t rr = row < cur.m ? row : row0; // clamped: the read stays inside the stage
				v[u] = staged_field(v32, cur.vbit0 + rr * cur.wv, vmlo, vmhi);
				key[u] = staged_field(k32, cur.kbit0 + rr * cur.wk, kmlo, kmhi);
			}
#pragma unroll
			for (int u = 0; u < 4; u++) {
				const uint32_t row = row0 + (uint32_t)u * kWorkgroup;
				uint64_t x = (v[u] + cur.vadd) & ty.v_tmask;
				x = (x ^ ty.v_sbit) - ty.v_sbit; // widen by T's signedness
				const uint64_t k = (key[u] + cur.kadd) & ty.k_tmask;
				const uint32_t bin = k < (uint64_t)ngroups ? (uint32_t)k : ngroups;
				const uint32_t slot = priv ? bin * kSets + my_set : bin;
				if (row < cur.m) {
					atomicAdd(&bsum[slot], (unsigned long long)x); // ds_add_u64, no return: nothing waits for it
					atomicAdd(&bcnt[slot], 1u);
				}
			}
		}
		if (more) {
#pragma unroll
			for (uint32_t h = 0; h < kGroupChunksPerThread; h++) {
				const uint32_t c = tid + h * kWorkgroup;
				if (c < nxt.vchunks) vstage[buf ^ 1u][c] = vq[h];
				if (c < nxt.kchunks) kstage[buf ^ 1u][c] = kq[h];
			}
		}
		__syncthreads();
		cur = nxt;
		have = more;
		buf ^= 1u;
	}
	// one partial per bin and workgroup
	unsigned long long *__restrict__ mine = partial + (uint64_t)blockIdx.x * 2u * nbins;
	if (priv) { // a bin's sets are added by the first wave
		if (tid < 64u) {
			for (uint32_t b = 0; b < nbins; b++) { // uniform
				static_assert(kSets <= 64u, "one lane per bin set");
				const uint64_t sv = tid < kSets ? (uint64_t)bsum[b * kSets + tid] : 0ull;
				const uint64_t cv = tid < kSets ? (uint64_t)bcnt[b * kSets + tid] : 0ull;
				const uint64_t ssum = wave_sum(sv), csum = wave_sum(cv);
				if (tid == 0u) {
					mine[2u * b] = ssum;
					mine[2u * b + 1u] = csum;
				}
			}
		}
	} else {
		for (uint32_t b = tid; b < nbins; b += kWorkgroup) {
			mine[2u * b] = bsum[b];
			mine[2u * b + 1u] = (unsigned long long)bcnt[b];
		}
	}
}

// one wave per bin: the partials of all workgroups -> sums[bin], counts[bin]
__global__ __launch_bounds__(64) void k_group_final(const unsigned long long *__restrict__ partial, uint32_t nwg,
                                                    uint32_t nbins, uint64_t *__restrict__ sums,
                                                    uint64_t *__restrict__ counts) {
	const uint32_t b = blockIdx.x;
	uint64_t s = 0, c = 0;
	for (uint32_t g = threadIdx.x; g < nwg; g += 64u) {
		s += partial[((uint64_t)g * nbins + b) * 2u];
		c += partial[((uint64_t)g * nbins + b) * 2u + 1u];
	}
	s = wave_sum(s);
	c = wave_sum(c);
	if (threadIdx.x == 0) {
		sums[b] = s;
		counts[b] = c;
	}
}
