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
constexpr uint32_t kGroupMaxWorkgroups = 4096; // capacity of the partial buffer: 8 per CU of the register-walk kernel + 7 per CU of this one

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

// ---- shared with k_group_sum_rw (below): who takes which segment pair
constexpr uint32_t kGroupRwCopies = 32;                      // bin sets per wave (two lanes share a set)

struct GroupRwPlan {
	bool ok;
	uint64_t vadd;      // added to every value field: the widened frame of reference (0 for raw segments)
	uint32_t kadd_byte; // added to every key field (fits a byte together with the field)
	bool keys_overflow; // every key of the segment is >= ngroups: the key bytes are a constant 255
};

// Can the register-walk kernel take this segment pair?  Uniform; depends on the two descriptors, the types and ngroups
// only, so both kernels agree on who takes what.
__device__ __forceinline__ GroupRwPlan group_rw_eligible(const adac_segment_desc &vd, const adac_segment_desc &kd,
                                                          const GroupSumTypes &ty, uint32_t ngroups) {
	GroupRwPlan p {false, 0ull, 0u, false};
	if (ngroups + 1u > kGroupPrivateBins || ty.wide_only) return p;
	const uint32_t wv = vd.width, wk = kd.width;
	if (wv < 4u || wv > 32u || wk > 8u || (uint64_t)vd.count * wv >= (1ull << 31)) return p;
	// value column: value64 = field + vadd without leaving T's range (seg_kind's SEG_LINEAR), or the field itself
	if ((vd.flags & ADAC_SEG_PACKED) && vd.min != ADAC_NO_MIN) {
		const uint64_t tmin = vd.min & ty.v_tmask;
		if (ty.v_sbit) {
			const uint64_t bmin = tmin ^ ty.v_sbit, top = bmin + ((1ull << wv) - 1ull);
			if (top < bmin || top > ty.v_tmask) return p; // wraps T's sign boundary
			p.vadd = (tmin ^ ty.v_sbit) - ty.v_sbit;
		} else {
			p.vadd = tmin;
		}
	} else if (ty.v_sbit) {
		return p; // raw slots of a signed type: the field is not the widened value
	}
	// key column: key = (field + kadd) & tmask as an unsigned number; all keys of the segment in a byte, or all >= ngroups
	const uint64_t kadd = effective_add(kd) & ty.k_tmask, kmaxf = (1ull << wk) - 1ull;
	if (kadd + kmaxf > ty.k_tmask) return p; // would wrap in the key type
	if (kadd + kmaxf <= 255ull) {
		p.kadd_byte = (uint32_t)kadd;
	} else if (kadd >= (uint64_t)ngroups) {
		p.keys_overflow = true;
	} else {
		return p;
	}
	p.ok = true;
	return p;
}

__global__ __launch_bounds__(kWorkgroup) void k_group_sum(const adac_segment_desc *__restrict__ vdescs,
                                                          const TileRef *__restrict__ vtiles, uint32_t ntiles,
                                                          const uint64_t *__restrict__ vwords,
                                                          const adac_segment_desc *__restrict__ kdescs,
                                                          const uint64_t *__restrict__ kwords, GroupSumTypes ty,
                                                          uint32_t ngroups, unsigned long long *__restrict__ partial,
                                                          const unsigned long long *__restrict__ rw_fallback) {
	// Runs after k_group_sum_rw (when that kernel was launched: rw_fallback != nullptr) and takes the segment pairs it
	// left: none, almost always — then every workgroup leaves at once with zero partials.
	if (rw_fallback != nullptr && *rw_fallback == 0ull) return; // uniform (k_group_final then leaves these partials out)
	const bool skip_rw = rw_fallback != nullptr;
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
		bool taken; // by k_group_sum_rw
	};
	const uint32_t G = gridDim.x;
	auto fetch_ref = [&](uint32_t tile) { return vtiles[tile < ntiles ? tile : 0u]; };
	auto resolve = [&](TileRef r, uint32_t tile) {
		TileMeta m;
		m.r = r;
		m.vd = load_desc_scalar(vdescs, r.seg);
		m.kd = load_desc_scalar(kdescs, r.seg);
		m.valid = tile < ntiles;
		m.taken = skip_rw && m.valid && group_rw_eligible(m.vd, m.kd, ty, ngroups).ok;
		return m;
	};
	uint32_t t = blockIdx.x, done = 0;
	TileMeta mcur = resolve(fetch_ref(t), t);
	TileMeta mnxt = resolve(fetch_ref(t + G), t + G);
	TileRef rnn = fetch_ref(t + 2u * G);
	auto advance = [&]() {
		mcur = mnxt;
		t += G;
		mnxt = resolve(rnn, t + G);
		rnn = fetch_ref(t + 2u * G);
		done = 0;
	};
	auto next_stage = [&](GroupStage &g) -> bool {
		if (mcur.valid) {
			const uint32_t left = mcur.vd.count - mcur.r.first;
			const uint32_t n = left < ty.v_tile_rows ? left : ty.v_tile_rows;
			if (done >= n) advance(); // uniform: on to the next tile
		}
		while (mcur.valid && mcur.taken) advance(); // uniform
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

// ------------------------------------------------------------------------------------------------------------------
// k_group_sum_rw — the same aggregate with the VALUE column on the fused scans' width-templated REGISTER WALK.
//
// k_group_sum above reads every field of both columns out of a staged LDS image: four dword reads and two LDS adds per
// row, each stage a chain of global -> LDS -> register round trips; 62 % of its wave cycles wait and it reaches 10 - 20 %
// of the HBM roofline (profiles/r02_q1_packed.json).  Here
//   * a lane owns whole 16-byte chunks of the VALUE stream (scan_run_w's form: one coalesced global_load_dwordx4 + the
//     next dword, the next chunk prefetched before this one is decoded, every field at a compile-time position);
//   * the KEYS of the rows a round of chunks covers (at most 4096 rows) are decoded once, by the lanes together, into
//     a BYTE per row in LDS: a lane takes a block of 32 rows = exactly WK dwords of the key stream (key widths 1..8),
//     finds every key at a compile-time position, adds the segment's frame of reference to four keys at a time and
//     stores 32 bytes with two ds_write_b128; a value lane then fetches the bytes of its 4 .. 32 rows with two to nine
//     dword reads and one v_alignbyte each, after which every key sits at a compile-time byte;
//   * bins as above (LDS adds without return), with 32 bin sets per wave instead of 16 (at most two lanes per set).
// The key block of the NEXT round is requested before the current round is walked and decoded into a second buffer
// after it: one LDS-only barrier per round, no load latency exposed.  LDS: 2 x 4.2 KiB of key bytes + 12 KiB of bins:
// seven workgroups per CU.
// Taken for a scan group when: nbins <= 8, value segment linear (or raw unsigned) at 4 <= w <= 32, key segment at
// w <= 8 whose keys all fit a byte (or all land in the overflow bin).  Every other group is counted in *fallback and
// left to k_group_sum, which runs after this kernel and skips the groups taken here (group_rw_eligible is shared).
// ------------------------------------------------------------------------------------------------------------------
// Key staging (the form for value widths whose chunk holds many rows): the keys of an 8-row block `q` are the wk BYTES
// [q * wk, (q + 1) * wk) of the key stream.  A lane loads the three dwords that hold them (the last one may be the
// segment's padding word: every segment owns at least one 64-bit word past its data) ...
__device__ __forceinline__ void group_rw_key_load(const uint32_t *__restrict__ kw32, uint32_t q, uint32_t wk,
                                                  uint32_t last_dword, uint32_t (&d)[3], uint32_t &byte_sh) {
	const uint32_t byte0 = q * wk;
	uint32_t dw = byte0 >> 2;
	dw = dw < last_dword ? dw : last_dword; // blocks past the segment's end are never read: any data will do
	byte_sh = byte0 & 3u;
	d[0] = kw32[dw];
	d[1] = kw32[dw + 1];                     // <= last_dword + 1: inside the padding word
	d[2] = kw32[dw + 2 <= last_dword + 1u ? dw + 2 : dw + 1];
}

// ... and takes them apart at a compile-time key width: 8 keys -> 8 bytes, the frame of reference added to four at a time
template <int WK>
__device__ __forceinline__ uint2 group_rw_key_decode(const uint32_t (&d)[3], uint32_t byte_sh, uint32_t kadd4) {
	const uint32_t w0 = __builtin_amdgcn_alignbyte(d[1], d[0], byte_sh), w1 = __builtin_amdgcn_alignbyte(d[2], d[1], byte_sh);
	constexpr uint32_t mask = (1u << WK) - 1u;
	uint32_t out[2];
#pragma unroll
	for (int g = 0; g < 2; g++) {
		uint32_t acc = 0;
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const int pos = (4 * g + b) * WK, dw = pos >> 5, sh = pos & 31; // pos + WK <= 64
			const uint32_t lo = dw ? w1 : w0;
			const uint32_t f = sh + WK <= 32 ? ((lo >> sh) & mask) : (__builtin_amdgcn_alignbit(w1, w0, sh) & mask);
			acc |= f << (8 * b);
		}
		out[g] = acc + kadd4; // four keys at once: no byte carries (field + kadd <= 255)
	}
	return make_uint2(out[0], out[1]);
}

__device__ __forceinline__ void group_rw_key_store(const uint32_t (&d)[3], uint32_t byte_sh, uint32_t wk, uint32_t kadd4,
                                                   bool overflow, uint2 *dst) {
	if (overflow) {
		*dst = make_uint2(~0u, ~0u);
		return;
	}
	switch (wk) { // uniform
	case 1: *dst = group_rw_key_decode<1>(d, byte_sh, kadd4); break;
	case 2: *dst = group_rw_key_decode<2>(d, byte_sh, kadd4); break;
	case 3: *dst = group_rw_key_decode<3>(d, byte_sh, kadd4); break;
	case 4: *dst = group_rw_key_decode<4>(d, byte_sh, kadd4); break;
	case 5: *dst = group_rw_key_decode<5>(d, byte_sh, kadd4); break;
	case 6: *dst = group_rw_key_decode<6>(d, byte_sh, kadd4); break;
	case 7: *dst = group_rw_key_decode<7>(d, byte_sh, kadd4); break;
	default: *dst = group_rw_key_decode<8>(d, byte_sh, kadd4); break;
	}
}

// One QUARTER of a scan group, walked by ONE WAVE on its own: the 64 lanes of a wave own consecutive chunks, i.e. one
// contiguous run of rows per round, so nothing is shared between the waves of a workgroup and no barrier is left in the
// loop (with a barrier per round the four waves advanced at the pace of the slowest: profiles/r03_group_sum_rw.json).
// Keys, two forms (uniform per segment pair):
//   DIRECT   the keys of a chunk's rows fit one dword (MAXV * wk <= 32: every value width >= 11 at wk = 3): a lane
//            loads the two dwords of the key stream that hold them (prefetched a round ahead like its value chunk),
//            shifts once, and finds key j with one v_bfe at the wave-uniform position j * wk.  No LDS for the keys.
//   staged   otherwise: the wave stages the key BYTES of the round's rows in a buffer of its own (8 rows per lane,
//            compile-time key width), and a lane fetches the bytes of its rows with dword reads + v_alignbyte.  LDS
//            operations of one wave are executed in order, so no barrier is needed here either.
// `wbins`: the wave's 8 x 32 bin words for the CURRENT quarter: rows << 44 | sum of their FIELDS (a word takes the rows
// of two lanes: a few hundred rows of at most 32 bits, far below 2^44 / 2^20), so a row costs ONE ds_add_u64 whose
// operand is {field, 0x1000}: no widening arithmetic, no second LDS add.  group_rw_fold turns the words into the
// wave's totals (sum of fields + rows x frame of reference) when the quarter is done.
constexpr uint32_t kGroupRwCountShift = 44;
// value widths of at most 8 bits: the same in a 32-bit word (rows << 20 | sum of fields: a word takes at most
// 2 x 16384 / 31 rows of a quarter, their fields sum to less than 2^20)
constexpr uint32_t kGroupRwNarrowShift = 20;
template <int W> constexpr bool kGroupRwNarrow = W <= 8;
constexpr uint32_t kGroupRwWaveRows = 1008;                    // key rows a wave stages per round: two 8-row blocks per lane
constexpr uint32_t kGroupRwWaveKeyBytes = kGroupRwWaveRows + 16 + 64; // + the block the round starts in + read slack
template <int W, bool DIRECT>
__device__ __forceinline__ void group_rw_walk(uint32_t r0, uint32_t r1, uint32_t count, const GroupRwPlan &plan,
                                              uint32_t wk, const uint4 *__restrict__ seg16,
                                              const uint32_t *__restrict__ kw32, uint32_t k_last_dword, uint32_t ngroups,
                                              uint8_t *keys, unsigned long long *wbins) {
	constexpr int MAXV = (128 + W - 1) / W;
	constexpr int KD = (MAXV + 3 + 3) / 4;                         // dwords holding MAXV bytes from any byte offset
	constexpr uint32_t LANES = (kGroupRwWaveRows * W / 128) < 64u ? (kGroupRwWaveRows * W / 128) : 64u; // chunks per round
	constexpr uint32_t PASSES = ((LANES * 128u / W + 8u + 7u) / 8u + 63u) / 64u; // staged: 8-row blocks per round / 64 lanes
	// r0 is a multiple of 128 rows: its bits start a chunk
	const uint32_t c0 = (uint32_t)(((uint64_t)r0 * W) >> 7);
	const uint32_t c1 = (uint32_t)(((uint64_t)r1 * W + 127) >> 7);
	const uint32_t clast = (uint32_t)(((uint64_t)count * W + 127) >> 7) - 1;
	const uint32_t lane = threadIdx.x & 63u;
	const bool walker = lane < LANES;
	const uint32_t kadd4 = plan.keys_overflow ? 0u : plan.kadd_byte * 0x01010101u;
	unsigned char *const my_bins = reinterpret_cast<unsigned char *>(wbins + (lane & (kGroupRwCopies - 1u)));
	unsigned char *const my_bins32 = reinterpret_cast<unsigned char *>(reinterpret_cast<uint32_t *>(wbins) + (lane & (kGroupRwCopies - 1u)));
	uint32_t L = c0 + lane;
	const uint32_t Lc = L < clast ? L : clast;
	uint4 q = seg16[Lc];
	uint32_t e = reinterpret_cast<const uint32_t *>(seg16 + (Lc < clast ? Lc + 1 : clast))[0];
	// DIRECT: the two dwords of the key stream holding the keys of the rows that start in chunk Lx
	auto direct_keys = [&](uint32_t Lx) {
		const uint32_t i0 = (128u * Lx + (W - 1)) / W;
		uint32_t dw = (i0 * wk) >> 5;
		dw = dw < k_last_dword ? dw : k_last_dword; // (chunks past the run: any data will do)
		return make_uint2(kw32[dw], kw32[dw + 1]);   // dw + 1 <= last data dword + 1: inside the padding word
	};
	// staged: the key rows of the round that starts at chunk `rc` = [first row starting in chunk rc, first row starting
	// in chunk rc + LANES), from the 8-row block the round starts in
	auto round_keys = [&](uint32_t rc, uint32_t &kb0, uint32_t &nblocks) {
		const uint32_t rows_lo = (128u * rc + (W - 1)) / W;
		kb0 = rows_lo & ~7u;
		uint32_t rows_hi = (128u * (rc + LANES) + (W - 1)) / W;
		rows_hi = rows_hi < r1 ? rows_hi : r1;
		nblocks = rows_hi > kb0 ? (rows_hi - kb0 + 7u) >> 3 : 0u;
	};
	uint2 kq = make_uint2(0u, 0u);
	uint32_t kb0 = 0, nblocks = 0;
	uint32_t kd[PASSES][3], ksh[PASSES];
	if (DIRECT) {
		kq = direct_keys(Lc);
	} else { // prologue: the first round's key bytes
		round_keys(c0, kb0, nblocks);
#pragma unroll
		for (uint32_t p = 0; p < PASSES; p++) group_rw_key_load(kw32, (kb0 >> 3) + lane + 64u * p, wk, k_last_dword, kd[p], ksh[p]);
#pragma unroll
		for (uint32_t p = 0; p < PASSES; p++) {
			if (lane + 64u * p < nblocks) {
				group_rw_key_store(kd[p], ksh[p], wk, kadd4, plan.keys_overflow, reinterpret_cast<uint2 *>(keys) + lane + 64u * p);
			}
		}
	}
	for (uint32_t round0 = c0; round0 < c1; round0 += LANES, L += LANES) { // uniform trip count
		// requested before this round is walked: the next chunk of the value stream and its keys
		const uint32_t Lp = L + LANES < clast ? L + LANES : clast;
		const uint4 qn = seg16[Lp];
		const uint32_t en = reinterpret_cast<const uint32_t *>(seg16 + (Lp < clast ? Lp + 1 : clast))[0];
		uint2 kqn = make_uint2(0u, 0u);
		uint32_t kb0n = 0, nblocksn = 0;
		if (DIRECT) {
			kqn = direct_keys(Lp);
		} else {
			round_keys(round0 + LANES, kb0n, nblocksn);
#pragma unroll
			for (uint32_t p = 0; p < PASSES; p++) group_rw_key_load(kw32, (kb0n >> 3) + lane + 64u * p, wk, k_last_dword, kd[p], ksh[p]);
		}
		if (walker && L < c1) {
			const uint32_t i0 = (128u * L + (W - 1)) / W; // first row starting in this chunk
			const uint32_t o0 = i0 * W - 128u * L;
			uint32_t nrm[5];
			nrm[0] = __builtin_amdgcn_alignbit(q.y, q.x, o0);
			nrm[1] = __builtin_amdgcn_alignbit(q.z, q.y, o0);
			nrm[2] = __builtin_amdgcn_alignbit(q.w, q.z, o0);
			nrm[3] = __builtin_amdgcn_alignbit(e, q.w, o0);
			nrm[4] = e >> o0;
			// MAXV - 1 rows start in every chunk, the MAXV-th one if its first bit still lies inside
			const bool last_starts = 128 % W == 0 || o0 + (uint32_t)(MAXV - 1) * W < 128u;
			const uint32_t starting = (uint32_t)(MAXV - 1) + (last_starts ? 1u : 0u);
			const uint32_t lim = r1 > i0 ? r1 - i0 : 0u;
			uint32_t kwin = 0;       // DIRECT: the keys of rows i0 .. from bit 0
			uint32_t kn[KD];         // staged: their bytes
			if (DIRECT) {
				kwin = __builtin_amdgcn_alignbit(kq.y, kq.x, (i0 * wk) & 31u);
			} else {
				// the key bytes of rows [i0, i0 + MAXV): dword reads from the byte offset rounded down, one v_alignbyte each
				const uint32_t kofs = i0 - kb0;
				const uint32_t *k32 = reinterpret_cast<const uint32_t *>(keys) + (kofs >> 2);
				uint32_t raw[KD + 1];
#pragma unroll
				for (int i = 0; i <= KD; i++) raw[i] = k32[i];
#pragma unroll
				for (int i = 0; i < KD; i++) kn[i] = __builtin_amdgcn_alignbyte(raw[i + 1], raw[i], kofs & 3u);
			}
			auto add_row = [&](int j) {
				uint32_t key;
				if (DIRECT) {
					key = plan.keys_overflow ? 255u : __builtin_amdgcn_ubfe(kwin, (uint32_t)j * wk, wk) + plan.kadd_byte;
				} else {
					key = (kn[j >> 2] >> (8 * (j & 3))) & 0xffu;
				}
				const uint32_t bin = key < ngroups ? key : ngroups;
				if (kGroupRwNarrow<W>) { // fields of at most 8 bits: rows << 20 | sum fits 32 bits, a ds_add_u32 is half the LDS work
					uint32_t *slot = reinterpret_cast<uint32_t *>(my_bins32 + bin * (kGroupRwCopies * 4u));
					atomicAdd(slot, (1u << kGroupRwNarrowShift) | field_of<W>(nrm, j));
				} else {
					unsigned long long *slot = reinterpret_cast<unsigned long long *>(my_bins + bin * (kGroupRwCopies * 8u));
					atomicAdd(slot, (1ull << kGroupRwCountShift) | (unsigned long long)field_of<W>(nrm, j)); // ds_add_u64, no return
				}
			};
			if (starting <= lim) { // every row that starts in the chunk belongs to the quarter: no per-row test
#pragma unroll
				for (int j = 0; j < MAXV - 1; j++) add_row(j);
				if (last_starts) add_row(MAXV - 1);
			} else { // the quarter ends inside this chunk
#pragma unroll
				for (int j = 0; j < MAXV; j++) {
					if ((uint32_t)j < lim) add_row(j);
				}
			}
		}
		q = qn;
		e = en;
		kq = kqn;
		if (!DIRECT) { // the next round's key bytes (issued after this round's reads of the buffer)
#pragma unroll
			for (uint32_t p = 0; p < PASSES; p++) {
				if (lane + 64u * p < nblocksn) {
					group_rw_key_store(kd[p], ksh[p], wk, kadd4, plan.keys_overflow,
					                   reinterpret_cast<uint2 *>(keys) + lane + 64u * p);
				}
			}
			kb0 = kb0n;
		}
	}
}

// The quarter is done: the wave's bin words -> its totals, and zero again.  total sum += sum of fields + rows x vadd
// (mod 2^64).  A lane reads four words (bins (lane >> 5) + 2 k); 32 lanes add into the same total.
__device__ __forceinline__ void group_rw_fold(unsigned long long *wbins, unsigned long long *wtot, uint64_t vadd,
                                              bool narrow) {
	const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
	for (uint32_t k = 0; k < kGroupPrivateBins * kGroupRwCopies / 64u; k++) {
		const uint32_t idx = lane + 64u * k;
		uint64_t cnt, fields;
		if (narrow) { // uniform: the quarter used the 32-bit words (the first half of the array)
			uint32_t *w32 = reinterpret_cast<uint32_t *>(wbins);
			const uint32_t v = w32[idx];
			if (v) w32[idx] = 0u;
			cnt = v >> kGroupRwNarrowShift;
			fields = v & ((1u << kGroupRwNarrowShift) - 1u);
		} else {
			const unsigned long long v = wbins[idx];
			if (v) wbins[idx] = 0ull;
			cnt = v >> kGroupRwCountShift;
			fields = v & ((1ull << kGroupRwCountShift) - 1ull);
		}
		if (cnt) {
			atomicAdd(&wtot[2u * (idx / kGroupRwCopies)], (unsigned long long)(fields + cnt * vadd));
			atomicAdd(&wtot[2u * (idx / kGroupRwCopies) + 1u], (unsigned long long)cnt);
		}
	}
}

__global__ __launch_bounds__(kWorkgroup, 7) void k_group_sum_rw(const ScanGroup *__restrict__ vgroups, uint32_t ngroups_work,
                                                                const uint64_t *__restrict__ vwords,
                                                                const adac_segment_desc *__restrict__ kdescs,
                                                                const uint64_t *__restrict__ kwords, GroupSumTypes ty,
                                                                uint32_t ngroups, unsigned long long *__restrict__ partial,
                                                                unsigned long long *__restrict__ fallback) {
	constexpr uint32_t kWaves = kWorkgroup / 64;
	__shared__ __attribute__((aligned(16))) uint8_t keys[kWaves][kGroupRwWaveKeyBytes];
	__shared__ unsigned long long bins[kWaves][kGroupPrivateBins * kGroupRwCopies];
	__shared__ unsigned long long tot[kWaves][2 * kGroupPrivateBins]; // per wave: {sum, rows} per bin
	const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
	const uint32_t nbins = ngroups + 1u;
	for (uint32_t i = lane; i < kGroupPrivateBins * kGroupRwCopies; i += 64u) bins[wave][i] = 0ull;
	if (lane < 2u * kGroupPrivateBins) tot[wave][lane] = 0ull;
	uint32_t skipped = 0;
	for (uint32_t gi = blockIdx.x; gi < ngroups_work; gi += gridDim.x) {
		const ScanGroup g = load_scan_group(vgroups, gi);
		const adac_segment_desc kd = load_desc(kdescs + g.seg);
		const GroupRwPlan plan = group_rw_eligible(g.d, kd, ty, ngroups);
		if (!plan.ok) { // uniform
			skipped++;
			continue;
		}
		// the group in four contiguous quarters of whole 128-row units (a quarter's bits start a 16-byte chunk), one per wave
		const uint32_t per = (((g.n + kWaves - 1u) / kWaves) + 127u) & ~127u;
		const uint32_t q0 = wave * per;
		if (q0 >= g.n) continue; // uniform per wave
		const uint32_t r0 = g.first + q0, r1 = g.first + (q0 + per < g.n ? q0 + per : g.n);
		const uint4 *seg16 = reinterpret_cast<const uint4 *>(vwords + g.d.word_off);
		const uint32_t *kw32 = reinterpret_cast<const uint32_t *>(kwords + kd.word_off);
		const uint32_t wk = kd.width;
		const uint32_t k_last = (uint32_t)(((uint64_t)kd.count * wk + 31) >> 5) - 1u;
		const uint32_t maxv = (128u + g.d.width - 1u) / g.d.width;
		if (maxv * wk <= 32u) { // uniform: the keys of a chunk's rows fit one dword
			switch (g.d.width) {
#define ADAC_W(N) case N: group_rw_walk<N, true>(r0, r1, g.d.count, plan, wk, seg16, kw32, k_last, ngroups, keys[wave], bins[wave]); break;
				ADAC_W(4) ADAC_W(5) ADAC_W(6) ADAC_W(7) ADAC_W(8) ADAC_W(9) ADAC_W(10) ADAC_W(11) ADAC_W(12) ADAC_W(13)
				ADAC_W(14) ADAC_W(15) ADAC_W(16) ADAC_W(17) ADAC_W(18) ADAC_W(19) ADAC_W(20) ADAC_W(21) ADAC_W(22)
				ADAC_W(23) ADAC_W(24) ADAC_W(25) ADAC_W(26) ADAC_W(27) ADAC_W(28) ADAC_W(29) ADAC_W(30) ADAC_W(31) ADAC_W(32)
#undef ADAC_W
			default: break;
			}
		} else {
			switch (g.d.width) {
#define ADAC_W(N) case N: group_rw_walk<N, false>(r0, r1, g.d.count, plan, wk, seg16, kw32, k_last, ngroups, keys[wave], bins[wave]); break;
				ADAC_W(4) ADAC_W(5) ADAC_W(6) ADAC_W(7) ADAC_W(8) ADAC_W(9) ADAC_W(10) ADAC_W(11) ADAC_W(12) ADAC_W(13)
				ADAC_W(14) ADAC_W(15) ADAC_W(16) ADAC_W(17) ADAC_W(18) ADAC_W(19) ADAC_W(20) ADAC_W(21) ADAC_W(22)
				ADAC_W(23) ADAC_W(24) ADAC_W(25) ADAC_W(26) ADAC_W(27) ADAC_W(28) ADAC_W(29) ADAC_W(30) ADAC_W(31) ADAC_W(32)
#undef ADAC_W
			default: break;
			}
		}
		group_rw_fold(bins[wave], tot[wave], plan.vadd, g.d.width <= 8u);
	}
	if (skipped && tid == 0u) atomicAdd(fallback, (unsigned long long)skipped);
	__syncthreads();
	// one partial per bin and workgroup: the four waves' totals
	unsigned long long *__restrict__ mine = partial + (uint64_t)blockIdx.x * 2u * nbins;
	if (tid < 2u * nbins) {
		unsigned long long v = 0;
#pragma unroll
		for (uint32_t w = 0; w < kWaves; w++) v += tot[w][tid];
		mine[tid] = v;
	}
}

// one workgroup per bin: the partials of all workgroups -> sums[bin], counts[bin].  The partial rows of k_group_sum are
// read only when it had something to do (no register-walk kernel, or *rw_fallback != 0).
__global__ __launch_bounds__(kWorkgroup) void k_group_final(const unsigned long long *__restrict__ partial,
                                                            uint32_t nwg_rw, uint32_t nwg_staged, uint32_t nbins,
                                                            uint64_t *__restrict__ sums, uint64_t *__restrict__ counts,
                                                            const unsigned long long *__restrict__ rw_fallback,
                                                            unsigned long long *__restrict__ next_fallback, int rw_ran) {
	__shared__ uint64_t ps[kWorkgroup / 64], pc[kWorkgroup / 64];
	const uint32_t b = blockIdx.x;
	// the hand-over word alternates between two slots from call to call: this call's is still being read by the other
	// workgroups of this kernel, so the one the NEXT call will use is cleared here
	if (b == 0u && threadIdx.x == 0u) *next_fallback = 0ull;
	const bool staged_ran = !rw_ran || *rw_fallback != 0ull; // uniform
	const uint32_t nwg = nwg_rw + (staged_ran ? nwg_staged : 0u);
	uint64_t s = 0, c = 0;
	for (uint32_t g = threadIdx.x; g < nwg; g += kWorkgroup) {
		s += partial[((uint64_t)g * nbins + b) * 2u];
		c += partial[((uint64_t)g * nbins + b) * 2u + 1u];
	}
	s = wave_sum(s);
	c = wave_sum(c);
	if ((threadIdx.x & 63u) == 0u) {
		ps[threadIdx.x >> 6] = s;
		pc[threadIdx.x >> 6] = c;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		uint64_t ts = 0, tc = 0;
#pragma unroll
		for (int w = 0; w < kWorkgroup / 64; w++) {
			ts += ps[w];
			tc += pc[w];
		}
		sums[b] = ts;
		counts[b] = tc;
	}
}
