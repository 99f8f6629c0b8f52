// adac_encode_1p.inl — single-pass encode: analyze -> width -> arena placement -> pack with the raw column read ONCE.
// Included into adac_kernels.hip inside namespace adac::{anonymous}.
//
// The two reference passes (min/max, then the pack loop: column_segment.cpp:390-400 + :426-443, or the append's running
// min/max + BitCompressFromSuccinct, succinct.cpp:286-299 + column_segment.cpp:365-376) read a segment twice; as two
// kernels that is 2.0 GB of HBM traffic for the 1.2 GB C2 needs.  Here ONE workgroup of 1024 threads owns a whole
// segment (<= 256 KiB of raw rows = sixteen 16-byte chunks per thread) and keeps it in REGISTERS across the phases:
//   1. all sixteen loads of a thread are issued back to back.  A segment fills half of a CU's register file, so there
//      is ONE workgroup per CU — and a CU streams at ~27 GB/s whatever the other CUs do (phase stamps,
//      profiles/r02_encode_experiments_2.json): while a workgroup reduces, waits in the look-back and packs, its CU's
//      memory pipe idles, which is why one segment per launch-time workgroup reads at 2.8 TB/s.  So the kernel is
//      PERSISTENT (one workgroup per CU, segments from a ticket counter) and keeps the pipe busy through those
//      phases: the first kEncPrefetch rounds of the NEXT segment travel global -> LDS with LDS-DMA
//      (global_load_lds_dwordx4: no destination registers, so nothing is spilled and no wait is forced — the variant
//      that handed registers over ran 0.40 ms) while the current one is processed, and move LDS ->
//      registers at the top of the next iteration while its remaining rounds load;
//   2. min / max under the rule (wave shuffles -> 16 partials through LDS), minmax[] written for adac_layout_get_minmax;
//   3. width, flags, stored min: exactly k_plan's arithmetic; the segment's arena footprint is published and its
//      arena offset obtained by a DECOUPLED LOOK-BACK over the predecessors' published footprints (one 64-bit word per
//      segment: 2 flag bits + value), so the result is the same exclusive prefix k_plan's single-workgroup scan gives.
//      Workgroups take their segments from a ticket counter and process their own in increasing order, so the lowest
//      unfinished segment is always some workgroup's CURRENT one and depends on nobody: every wait ends;
//   4. pack from the registers: a thread's K rows -> one bit string (concat_fields) -> ds_or_b64 into a zeroed LDS
//      image of the output words of a STAGE (as many 1024-chunk rounds as fit 48 KiB), copied out with 16-byte stores.
// Stages cover rows [r0 * 1024 K, r1 * 1024 K): multiples of 64 rows, so a stage's bits start on a word (in fact
// 256-byte) boundary whatever the segment's placement in the value buffer; the chunks that straddle a stage boundary
// because of that placement (thread 0's only) and the segment's ragged tail go row by row.
// Results are bit-identical to k_analyze + k_plan + k_pack (tests: run_encode_decode on every type, width, rule,
// NULL mask, padded mode, with the knob "single_pass_encode" 0 and 1).

constexpr int kEncThreads = 1024;
constexpr int kEncRounds = 16;                         // 16-byte chunks per thread
constexpr uint32_t kEncImageWords = 48 * 1024 / 8;     // LDS image of one stage's output words
constexpr int kEncPrefetch = 6;                        // rounds of the next segment prefetched into LDS (96 KiB)
constexpr unsigned long long kScanFlagAggregate = 1ull << 62, kScanFlagPrefix = 2ull << 62, kScanValueMask = (1ull << 62) - 1;

// exclusive prefix of the footprints of segments [0, seg): wave 0 of the workgroup, all 64 lanes.
// (*probe: a diagnostic of round 3 counted here the round trips that found everything they needed and the ones that
// met a predecessor which had not published yet — 1.0 - 1.4 against 41 - 43 per segment at the image-flow widths,
// profiles/r03_encode_lookback.json; the counters cost the 4-byte instantiation its last free registers and are gone)
__device__ __forceinline__ uint64_t lookback_exclusive(unsigned long long *state, uint32_t seg, uint32_t *probe) {
	const uint32_t lane = threadIdx.x & 63u;
	uint64_t sum = 0;
	int64_t hi = (int64_t)seg - 1; // nearest predecessor not yet accounted for
	while (hi >= 0) {
		const int64_t idx = hi - (int64_t)lane;
		unsigned long long v = kScanFlagPrefix; // lanes before segment 0 read as "prefix 0"
		if (idx >= 0) v = __hip_atomic_load(&state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint64_t unset = __builtin_amdgcn_ballot_w64((v >> 62) == 0ull);
		if (unset) {
			__builtin_amdgcn_s_sleep(2); // a predecessor has not published its footprint yet
			continue;
		}
		const uint64_t prefixed = __builtin_amdgcn_ballot_w64((v >> 62) == 2ull);
		// lanes are ordered nearest predecessor first: take everything up to and including the first prefix
		const uint32_t stop = prefixed ? (uint32_t)__ffsll((unsigned long long)prefixed) - 1u : 63u;
		uint64_t part = lane <= stop ? (uint64_t)(v & kScanValueMask) : 0ull;
		part = wave_sum(part);
		sum += part;
		if (prefixed) break;
		hi -= 64;
	}
	(void)probe;
	return sum;
}

// a value every lane holds alike, moved to scalar registers (the builtin returns int: mind the sign)
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
	const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
	const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
	return ((uint64_t)hi << 32) | lo;
}

// N consecutive dwords to a dword-aligned address as ONE store instruction (the segment's first row may sit in the
// middle of an aligned pair, so the address is not N-dword aligned in general; the device runs in unaligned access mode)
template <int N>
__device__ __forceinline__ void store_dwords(uint32_t *p, const uint32_t *s) {
	struct __attribute__((packed, aligned(4))) Pack {
		uint32_t d[N];
	};
	if (N == 2) { // non-temporal: the packed words are written once (1.5 - 2.5 % on the whole encode)
		typedef uint32_t v2u __attribute__((ext_vector_type(2)));
		v2u qq = {s[0], s[1]};
		__builtin_nontemporal_store(qq, reinterpret_cast<v2u *>(p));
		return;
	}
	Pack v;
#pragma unroll
	for (int i = 0; i < N; i++) v.d[i] = s[i];
	*reinterpret_cast<Pack *>(p) = v;
}

// diagnostic time stamps (tuning knob "encode_stamps"): thread 0 records the 100 MHz wall clock at the phase boundaries
// of every segment it processes; slot 7 holds the hardware id (XCC / SE / CU): adac_debug_encode_stamps reads them back
constexpr uint32_t kEncStampSlots = 8, kEncStampMax = 16384;
__device__ unsigned long long g_enc_stamps[kEncStampMax * kEncStampSlots];
#define ADAC_STAMP(k)                                                                                                  \
	do {                                                                                                               \
		if (stamps && tid == 0u && seg < kEncStampMax) g_enc_stamps[seg * kEncStampSlots + (k)] = wall_clock64();      \
	} while (0)

// what a workgroup needs to know of a segment before its rows arrive (all wave-uniform)
template <typename U>
struct EncSegment {
	uint32_t seg, n, align, nchunks, last_chunk;
	uint64_t val_off;
	const uint4 *base16; // the 16-byte chunk holding the segment's first row
};

template <typename U>
__device__ __forceinline__ EncSegment<U> enc_segment(const adac_segment_desc *__restrict__ descs, const U *__restrict__ vals,
                                                     uint32_t seg, uint32_t nseg) {
	constexpr int K = 16 / (int)sizeof(U);
	EncSegment<U> g;
	g.seg = seg;
	// a ticket past the end describes an empty segment that reads (and ignores) the first descriptor
	const adac_segment_desc d = descs[seg < nseg ? seg : 0u];
	g.n = seg < nseg ? d.count : 0u;
	g.val_off = d.val_off;
	g.align = (uint32_t)(d.val_off & (K - 1));
	// 16-byte aligned; chunk c = elements [c K, c K + K).  An empty segment reads (and ignores) a descriptor: its val_off
	// may be the end of the value buffer
	g.base16 = g.n ? reinterpret_cast<const uint4 *>(vals + (d.val_off - g.align)) : reinterpret_cast<const uint4 *>(descs);
	g.nchunks = (g.n + g.align + K - 1) / K; // <= 16 * 1024 (checked by the host)
	g.last_chunk = g.nchunks ? g.nchunks - 1u : 0u;
	return g;
}

// 16 bytes to global memory, non-temporal (streamed past the caches: written once, read much later)
__device__ __forceinline__ void nt_store16(void *p, uint4 o) {
	typedef uint32_t v4u __attribute__((ext_vector_type(4)));
	v4u qq = {o.x, o.y, o.z, o.w};
	__builtin_nontemporal_store(qq, reinterpret_cast<v4u *>(p));
}

// a barrier that orders the workgroup's LDS traffic only.  __syncthreads() also drains every vector memory operation
// of the wave (s_waitcnt vmcnt(0)): the stores of the stage just written out, and the LDS-DMA prefetch in flight
__device__ __forceinline__ void lds_barrier() {
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// descs_in is the same table as descs, read-only: the fields the kernel reads (count, val_off) are never written, and
// a const __restrict__ view lets them come through the scalar cache — a vector load would queue behind the wave's
// other vector memory operations (vmcnt returns in order)
template <typename U>
__global__ __launch_bounds__(kEncThreads) void k_encode_1p(adac_segment_desc *__restrict__ descs,
                                                           const adac_segment_desc *__restrict__ descs_in, uint32_t nseg,
                                                           uint64_t *__restrict__ minmax, const U *__restrict__ vals,
                                                           const uint64_t *__restrict__ validity, int sign_extend,
                                                           uint64_t null_bits, int rule, int pad,
                                                           unsigned long long *__restrict__ scan_state,
                                                           uint32_t *__restrict__ ticket, uint64_t *__restrict__ words,
                                                           int stamps, int placement) {
	constexpr int K = 16 / (int)sizeof(U);
	constexpr uint32_t ROUND_ROWS = kEncThreads * K; // rows one round of chunks covers: a multiple of 64
	constexpr uint32_t type_bits = 8 * sizeof(U);
	using S = typename std::make_signed<U>::type;
	// one LDS pool: [stage: the next segment's first rounds | img: the stage image of the pack]; the PARKING area of
	// the whole-dword path (2 dwords x 16 rounds x 1024 threads = 128 KiB) lies over both
	constexpr uint32_t kStageBytes = kEncPrefetch * kEncThreads * 16u, kImgBytes = (kEncImageWords + 4) * 8u;
	static_assert(kStageBytes + kImgBytes >= 2u * 4u * kEncRounds * kEncThreads, "parking area does not fit the pool");
	__shared__ __attribute__((aligned(16))) unsigned char pool[kStageBytes + kImgBytes];
	uint4 *const stage = reinterpret_cast<uint4 *>(pool);
	unsigned long long *const img = reinterpret_cast<unsigned long long *>(pool + kStageBytes);
	uint32_t *const park = reinterpret_cast<uint32_t *>(pool);
	// the whole pool as ONE image (round 3, "big image"): a segment whose packed output fits it is packed into LDS in one
	// go, the next segment is loaded and analysed, and only then is the arena offset asked for (publish-ahead, see `pend`)
	constexpr uint32_t kPoolWords = (kStageBytes + kImgBytes) / 8u - 4u;
	unsigned long long *const pool64 = reinterpret_cast<unsigned long long *>(pool);
	struct ImagePending { // the segment whose image waits in the pool (written by thread 0, read by everybody: uniform)
		uint64_t mn, mx, stored_min, footprint;
		uint32_t seg, w, flags, nwords;
	};
	__shared__ ImagePending s_ipend;
	bool ipend_active = false; // (uniform)
	bool img_dirty = false;   // the parking area was used since the image was last all zero (uniform)
	bool stage_dirty = true;  // the stage region holds prefetched rows, or was never cleared (uniform)
	bool staged = false;      // the next segment's first rounds were prefetched into the stage this iteration (uniform)
	bool next_loaded = false; // the next segment's loads were issued by the parked flow (uniform)
	__shared__ uint64_t pmin[kEncThreads / 64], pmax[kEncThreads / 64];
	__shared__ uint64_t s_word_off;
	__shared__ uint32_t s_ticket;
	const uint32_t tid0 = threadIdx.x;
	uint32_t tid = tid0;

	if (tid == 0) s_ticket = atomicAdd(ticket, 1u);
	for (uint32_t i = tid; i < kEncImageWords + 4; i += kEncThreads) img[i] = 0ull;
	__syncthreads();
	const uint32_t t0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket);
	if (t0 >= nseg) return;
	EncSegment<U> cur = enc_segment<U>(descs_in, vals, t0, nseg);

	// ---- 1. the whole segment into registers.  UNCONDITIONAL loads (index clamped to the segment's last chunk): a
	// load under a condition gets its own s_waitcnt from the compiler and the sixteen round trips would run one after
	// the other.  A segment's first / last chunk is loaded whole: the elements before its first row belong to the
	// previous segment, the ones after its last row lie inside the same aligned 16 bytes of the buffer; both are
	// masked out below by their row numbers.
	uint4 q[kEncRounds];
#pragma unroll
	for (int r = 0; r < kEncRounds; r++) {
		const uint32_t c = (uint32_t)r * kEncThreads + tid;
		q[r] = cur.base16[c < cur.last_chunk ? c : cur.last_chunk];
	}
	// thread 0 keeps the ticket of the segment after the current one; it is always taken BEFORE the wave's next batch
	// of loads is issued (a wave's vector memory operations return in order: behind the loads it would arrive last)
	uint32_t next_ticket = 0;
	if (tid == 0) next_ticket = atomicAdd(ticket, 1u);
	// what phases 2 - 3 produce for a segment (all wave-uniform)
	struct Analysis {
		uint64_t mn, mx, stored_min, footprint;
		uint32_t w;
		uint8_t flags;
	};
	// PUBLISH-AHEAD (ordered placement, parked flow): a segment whose strings are parked in LDS does not ask for its
	// arena offset — the one point where a workgroup can be held up — before the NEXT segment has been loaded (by every
	// wave), analysed and its footprint published.  With ordered placement a round of segments advances at the pace of
	// its slowest workgroup, and a workgroup that waited used to publish its next footprint later still
	// (profiles/r03_encode_lookback.json); now nobody's footprint waits for anybody's offset.  `pend` = the parked
	// segment whose offset and stores are still due; it is drained right after the next segment's analysis.
	struct Pending {
		bool active;
		uint32_t seg, n, align, nchunks, nd, skip, nd_total;
		Analysis a;
	};
	Pending pend {};
	// (8-byte types only: the 4-byte instantiation has no register to spare for the deferred segment's record — 100
	// spills with it — and takes this kernel only under first-come placement, where nothing waits)
	constexpr bool kPublishAhead = sizeof(U) == 8;

	for (;;) {
	// (the thread index is made opaque once per segment: everything derived from it — sixteen rounds of chunk numbers,
	// byte offsets, LDS addresses — is loop invariant, and hoisted out of this loop it costs more registers than the
	// segment leaves free: the compiler then keeps the sixteen chunks in scratch memory)
	tid = tid0;
	asm volatile("" : "+v"(tid));
	const uint32_t wave_chunk0 = tid & ~63u; // first chunk of this wave inside a round
	// rounds [0, kEncPrefetch) of a segment, global -> LDS, fire and forget.  Wave 0 copies nothing: its look-back
	// loads would queue behind its own copies (a wave's vector memory operations return in order) and the whole
	// workgroup waits for that look-back; the last wave copies wave 0's pieces as well.  The barrier of phase 3
	// (__syncthreads: every wave drains its vmcnt first) orders the copies before the reads at the top of the next
	// iteration
	auto prefetch = [&](const EncSegment<U> &g) {
		if (g.seg >= nseg || tid < 64u) return; // uniform per wave
#pragma unroll
		for (int r = 0; r < kEncPrefetch; r++) {
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			__builtin_amdgcn_global_load_lds((gptr_t)(g.base16 + (c < g.last_chunk ? c : g.last_chunk)),
			                                 (lptr_t)(stage + (uint32_t)r * kEncThreads + wave_chunk0), 16, 0, 0);
		}
		if (tid >= kEncThreads - 64u) {
#pragma unroll
			for (int r = 0; r < kEncPrefetch; r++) {
				const uint32_t c = (uint32_t)r * kEncThreads + (tid & 63u);
				__builtin_amdgcn_global_load_lds((gptr_t)(g.base16 + (c < g.last_chunk ? c : g.last_chunk)),
				                                 (lptr_t)(stage + (uint32_t)r * kEncThreads), 16, 0, 0);
			}
		}
	};
	const uint32_t seg = cur.seg;
	const uint32_t n = cur.n, align = cur.align, nchunks = cur.nchunks;
	const uint64_t seg_val_off = cur.val_off;
	ADAC_STAMP(0);
	if (stamps && tid == 0u && seg < kEncStampMax) {
		uint32_t hw, xcc;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
		g_enc_stamps[seg * kEncStampSlots + 7] = ((unsigned long long)xcc << 32) | hw;
	}
	// the next segment's ticket (taken a batch of loads ago); the barrier of phase 2 publishes it
	if (tid == 0) s_ticket = next_ticket;

	// ---- 2. min / max under the rule.  A round whose 1024 chunks all lie inside the segment (wave-uniform test; all
	// rounds but the first of a misplaced segment and the last one) takes a body without row or NULL tests; for the
	// types of up to 32 bits it works on the raw 32-bit patterns: the order of uint64(sign- or zero-extended x) is the
	// unsigned order of x's own bits (negative values extend to the largest numbers), so one v_min_u32 / v_max_u32 per
	// row does, and the result is widened once at the end.
	uint64_t mn = ~0ull, mx = 0;
	uint32_t mn32 = 0xffffffffu, mx32 = 0u;
	const bool widen_signed = rule == ADAC_RULE_APPEND && sign_extend;
#pragma unroll
	for (int r = 0; r < kEncRounds; r++) {
		const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
		if (round_row >= n + align) continue; // (not `break`: a constant trip count keeps the loop unrollable inside the segment loop)
		const bool interior = !validity && round_row >= align && round_row + ROUND_ROWS - align <= n; // uniform
		U v[K];
		__builtin_memcpy(v, &q[r], 16);
		if (interior) {
#pragma unroll
			for (int j = 0; j < K; j++) {
				if (sizeof(U) == 8) {
					const uint64_t x = (uint64_t)v[j]; // 64-bit T: sign extension is the identity
					mn = x < mn ? x : mn;
					mx = x > mx ? x : mx;
				} else {
					mn32 = (uint32_t)v[j] < mn32 ? (uint32_t)v[j] : mn32;
					mx32 = (uint32_t)v[j] > mx32 ? (uint32_t)v[j] : mx32;
				}
			}
		} else {
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			const int32_t row0 = (int32_t)(c * K) - (int32_t)align;
			const uint32_t vbits = chunk_validity(validity, (seg_val_off - align) + (uint64_t)c * K);
#pragma unroll
			for (int j = 0; j < K; j++) {
				if (c >= nchunks || (uint32_t)(row0 + j) >= n) continue;
				const bool valid = (vbits >> j) & 1u;
				uint64_t x;
				if (rule == ADAC_RULE_APPEND) { // succinct.cpp:286-287: uint64_t(sdata[i]); NULL rows do not take part
					if (!valid) continue;
					x = sign_extend ? (uint64_t)(int64_t)(S)v[j] : (uint64_t)v[j];
				} else { // column_segment.cpp:392-399: every slot, zero-extended; NULL slots hold NullValue<T>
					x = valid ? (uint64_t)v[j] : null_bits;
				}
				mn = x < mn ? x : mn;
				mx = x > mx ? x : mx;
			}
		}
	}
	if (sizeof(U) < 8 && mn32 <= mx32) { // some interior row was seen: widen the 32-bit extrema
		const uint64_t a = widen_signed ? (uint64_t)(int64_t)(S)(U)mn32 : (uint64_t)mn32;
		const uint64_t b = widen_signed ? (uint64_t)(int64_t)(S)(U)mx32 : (uint64_t)mx32;
		mn = a < mn ? a : mn;
		mx = b > mx ? b : mx;
	}
	ADAC_STAMP(1);
	mn = wave_min(mn);
	mx = wave_max(mx);
	if ((tid & 63u) == 0u) {
		pmin[tid >> 6] = mn;
		pmax[tid >> 6] = mx;
	}
	lds_barrier();
	// this segment's rows are all in registers: from here to the end of the pack the CU's memory pipe would idle, so
	// the first rounds of the next segment start travelling now
	const EncSegment<U> nxt = enc_segment<U>(descs_in, vals, (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ticket), nseg);
	ADAC_STAMP(2);
#pragma unroll
	for (int i = 0; i < kEncThreads / 64; i++) {
		mn = pmin[i] < mn ? pmin[i] : mn;
		mx = pmax[i] > mx ? pmax[i] : mx;
	}
	// every thread now holds the same pair: tell the compiler (everything derived from it — width, masks, stage
	// geometry — then lives in scalar registers instead of costing a vector register per thread)
	mn = uniform64(mn);
	mx = uniform64(mx);

	// ---- 3. width, flags, stored min (k_plan), footprint, arena offset
	uint32_t w = type_bits;
	uint8_t flags = 0;
	uint64_t stored_min = mn;
	{
		const uint32_t cand = width_for(mn, mx, rule, pad);
		if (type_bits > cand) { // `if (old_width > min_width)` column_segment.cpp:363,420
			w = cand;
			flags = ADAC_SEG_PACKED;
			if (mn == ~0ull && mx == ~0ull) stored_min = ~0ull - mask64(w); // sentinel collision, see k_plan
		}
	}
	const uint64_t footprint = ((((uint64_t)n * w + 64) >> 6) + 15) & ~15ull; // SDSL allocation, rounded to 128 B
	if (tid == 0) { // publish the own footprint first: successors can then pass over this segment
		__hip_atomic_store(&scan_state[seg], (seg == 0 ? kScanFlagPrefix : kScanFlagAggregate) | footprint,
		                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	const Analysis ana {mn, mx, stored_min, footprint, w, flags}; // (what the deferred form keeps of this segment)
	// The arena offset is only needed by the first STORE, and it is obtained as late as that.  The look-back waits for
	// the slowest of the predecessors in flight (256 workgroups with two segments each: an extreme-value wait of
	// 5 - 11 us per segment when it followed the publication at once, profiles/r02_encode_experiments_2.json); every
	// piece of work that fits between publishing the footprint and asking for the prefix shortens it.
	unsigned long long *__restrict__ dst = nullptr;
	const int first_come = placement & 1; // 1: arena order = order of completion (no wait at all; offsets differ run to run)
	// arena offset of segment `pseg` (analysis `a`), its descriptor and min / max written; returns where its words go
	auto place_segment = [&](const Analysis &a, uint32_t pseg, bool drain) __attribute__((always_inline)) -> unsigned long long * {
		const uint32_t seg = pseg; // (the stamp macro names it)
		if (tid < 64) {
			uint64_t excl;
			if (first_come) {
				unsigned long long got = 0;
				if (tid == 0) got = atomicAdd(scan_state + nseg + 1, (unsigned long long)a.footprint);
				excl = uniform64(got);
			} else {
				uint32_t probe = 0;
				excl = seg == 0 ? 0ull : lookback_exclusive(scan_state, seg, &probe);
				if (stamps && tid == 0u && seg < kEncStampMax) g_enc_stamps[seg * kEncStampSlots + 6] = probe;
			}
			if (tid == 0) {
				if (seg != 0 && !first_come) {
					__hip_atomic_store(&scan_state[seg], kScanFlagPrefix | (excl + a.footprint), __ATOMIC_RELAXED,
					                   __HIP_MEMORY_SCOPE_AGENT);
				}
				s_word_off = excl;
				minmax[2 * (uint64_t)seg] = a.mn;
				minmax[2 * (uint64_t)seg + 1] = a.mx;
				descs[seg].word_off = excl;
				descs[seg].min = a.stored_min;
				descs[seg].width = (uint8_t)a.w;
				descs[seg].flags = a.flags;
				descs[seg].reserved = 0;
			}
			ADAC_STAMP(5); // (diagnostic: the look-back itself, before the barrier that also waits for the prefetch)
		}
		if (drain) {
			__syncthreads(); // also drains every wave's vmcnt: the LDS-DMA prefetch has landed when the pack is over
		} else {
			lds_barrier(); // (the parked flow has the next segment's loads in flight: they must not be waited for)
		}
		ADAC_STAMP(3);
		return reinterpret_cast<unsigned long long *>(words) + uniform64(s_word_off);
	};
	auto place = [&](bool drain) {
		if (tid < 64) {
			uint64_t excl;
			if (first_come) {
				unsigned long long got = 0;
				if (tid == 0) got = atomicAdd(scan_state + nseg + 1, (unsigned long long)footprint);
				excl = uniform64(got);
			} else {
				uint32_t probe = 0;
				excl = seg == 0 ? 0ull : lookback_exclusive(scan_state, seg, &probe);
				if (stamps && tid == 0u && seg < kEncStampMax) g_enc_stamps[seg * kEncStampSlots + 6] = probe;
			}
			if (tid == 0) {
				if (seg != 0 && !first_come) {
					__hip_atomic_store(&scan_state[seg], kScanFlagPrefix | (excl + footprint), __ATOMIC_RELAXED,
					                   __HIP_MEMORY_SCOPE_AGENT);
				}
				s_word_off = excl;
				minmax[2 * (uint64_t)seg] = mn;
				minmax[2 * (uint64_t)seg + 1] = mx;
				descs[seg].word_off = excl;
				descs[seg].min = stored_min;
				descs[seg].width = (uint8_t)w;
				descs[seg].flags = flags;
				descs[seg].reserved = 0;
			}
			ADAC_STAMP(5); // (diagnostic: the look-back itself, before the barrier that also waits for the prefetch)
		}
		if (drain) {
			__syncthreads(); // also drains every wave's vmcnt: the LDS-DMA prefetch has landed when the pack is over
		} else {
			lds_barrier(); // (the parked flow has the next segment's loads in flight: they must not be waited for)
		}
		ADAC_STAMP(3);
		dst = reinterpret_cast<unsigned long long *>(words) + uniform64(s_word_off);
	};
	// the parked strings of a segment (park slots of thread-own rounds) -> the arena
	auto store_parked = [&](uint32_t *__restrict__ out32, uint32_t pn, uint32_t palign, uint32_t pnchunks, uint32_t nd,
	                        uint32_t skip, uint32_t nd_total) __attribute__((always_inline)) {
#pragma unroll
		for (int r = 0; r < kEncRounds; r++) {
			const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
			if (round_row >= pn + palign) continue; // uniform
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			const bool interior = round_row >= palign && round_row + ROUND_ROWS - palign <= pn; // uniform
			const uint32_t *slot = park + ((uint32_t)r * kEncThreads + tid) * 2u;
			uint32_t s[2];
			if (nd == 2u) {
				const uint2 t2 = *reinterpret_cast<const uint2 *>(slot);
				s[0] = t2.x, s[1] = t2.y;
			} else {
				s[0] = slot[0], s[1] = 0u;
			}
			const uint32_t g0 = c * nd - skip;
			if (interior) {
				if (nd == 1u) {
					__builtin_nontemporal_store(s[0], out32 + g0);
				} else {
					store_dwords<2>(out32 + g0, s);
				}
			} else if (c < pnchunks) {
#pragma unroll
				for (uint32_t j = 0; j < 2u; j++) {
					if (j < nd && g0 + j < nd_total) out32[g0 + j] = s[j];
				}
				if (c == pnchunks - 1u && g0 + nd < nd_total) out32[nd_total - 1u] = 0u;
			}
		}
	};
	// the parked segment before this one: its successor (this segment) is analysed and published — now its offset
	if constexpr (kPublishAhead) if (pend.active) { // uniform
		unsigned long long *pdst = place_segment(pend.a, pend.seg, false);
		store_parked(reinterpret_cast<uint32_t *>(pdst), pend.n, pend.align, pend.nchunks, pend.nd, pend.skip, pend.nd_total);
		pend.active = false;
		lds_barrier(); // the parking area is free again before this segment's pack may use the pool
	}
	// ... or the segment whose whole image waits in the pool: its offset, its words out of LDS (zeroed as they leave)
	if (ipend_active) { // uniform
		Analysis pa;
		pa.mn = uniform64(s_ipend.mn), pa.mx = uniform64(s_ipend.mx), pa.stored_min = uniform64(s_ipend.stored_min);
		pa.footprint = uniform64(s_ipend.footprint);
		pa.w = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ipend.w);
		pa.flags = (uint8_t)__builtin_amdgcn_readfirstlane((int)s_ipend.flags);
		const uint32_t pseg = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ipend.seg);
		const uint32_t pwords = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_ipend.nwords);
		unsigned long long *__restrict__ pout = place_segment(pa, pseg, false);
		for (uint32_t i = 2u * tid; i < pwords; i += 2u * kEncThreads) {
			const unsigned long long v0 = pool64[i], v1 = pool64[i + 1];
			pool64[i] = 0ull;
			pool64[i + 1] = 0ull;
			if (i + 1 < pwords) {
				uint4 o;
				o.x = (uint32_t)v0;
				o.y = (uint32_t)(v0 >> 32);
				o.z = (uint32_t)v1;
				o.w = (uint32_t)(v1 >> 32);
				nt_store16(pout + i, o);
			} else {
				pout[i] = v0;
			}
		}
		ipend_active = false;
		lds_barrier(); // the pool is all zero again before this segment's pack may use it
	}
	do { // phase 4 (left with `break` where the one-segment form returned)
	// whole-dword strings (4a below)?  With at most two dwords per chunk the strings are PARKED in LDS, the next
	// segment's loads are issued into the freed registers, and only then is the arena offset asked for: the look-back
	// wait overlaps those loads instead of idling the CU's memory pipe.  Other segments prefetch through LDS-DMA.
	const bool whole_dwords = n != 0u && !validity && ((K * w) & 31u) == 0u && ((align * w) & 31u) == 0u &&
	                          (w <= 32u || w == type_bits);
	bool parked = whole_dwords && K * w <= 64u;
	// big image: every other width of a segment whose packed words fit the pool, when another segment follows and the
	// placement is ordered (bit 2 of `placement`: knob encode_big_image)
	const uint32_t seg_words = (uint32_t)(((uint64_t)n * w + 63u) >> 6);
	// (by WIDTH, not by this segment's size: a full segment of kEncRounds rounds at this width must fit, so that the
	// segments of a column take the same flow as long as their width stays — a column whose 65534-row segments do not
	// fit while its shorter ones do alternates between the flows and ran 2x slower: u32 w = 20, profiles/r03_encode_big_image.json)
	// The ONE-dword strings of the parked flow (K w = 32: u64 w = 16, u32 w = 8) take the big image too: at those widths
	// a segment parks only at some placements in the value buffer (a multiple of 32 bits before its first row), so a column
	// alternated between the parked and the staged flow — 0.480 -> 0.444 ms at u64 w = 16, 0.583 -> 0.569 at u32 w = 8.
	// Two-dword strings (u64 w = 32, u32 w = 16) stay parked: 0.248 / 0.305 against 0.252 / 0.310 ms through the image.
	const bool bigimg = (!whole_dwords || (parked && K * w == 32u)) && n != 0u && !validity && w <= 32u && seg_words + 2u <= kPoolWords &&
	                    (uint64_t)kEncRounds * ROUND_ROWS * w + 128u <= (uint64_t)kPoolWords * 64u &&
	                    nxt.seg < nseg && !first_come && (placement & 4);
	if (bigimg) parked = false;
	const bool whole_dwords_now = whole_dwords && !bigimg;
	staged = !parked && !bigimg;
	if (staged) {
		prefetch(nxt);
		stage_dirty = stage_dirty || nxt.seg < nseg;
	}
	if (n == 0) {
		place(true);
		break;
	}

	// ---- 4. pack from the registers, stage by stage
	const U sub = (U)(((flags & ADAC_SEG_PACKED) && stored_min != ADAC_NO_MIN) ? stored_min : 0ull); // column_segment.cpp:371-373
	const U wmask = (U)mask64(w);

	// ---- 4a. whole-dword strings go straight from the registers to the arena.  When a chunk's K fields fill a whole
	// number of dwords (K w a multiple of 32: every byte-multiple width of the 4-byte types, 16 / 32 / 48 / 64 bits of
	// the 8-byte ones — the padded mode's widths (column_segment.cpp:357-359) and C2's w = 32) and the rows before the
	// segment's first one fill whole dwords too, thread t's string IS dwords [c nd - skip, c nd - skip + nd) of the
	// segment: one coalesced store per chunk, no LDS image, no barrier, and the stores of a round overlap the next
	// round's arithmetic.  The image path below cost ~20 vector instructions + 3 ds_or per chunk plus the copy-out.
 	if (whole_dwords_now) {
		const uint32_t nd = (K * w) >> 5;                                              // dwords per chunk: 1..4
		const uint32_t skip = (align * w) >> 5;                                        // dwords of the rows before row 0
		const uint32_t nd_total = 2u * (uint32_t)(((uint64_t)n * w + 63u) >> 6);       // the segment's words, in dwords
		// the strings replace the rows in the registers (the look-back has the time this takes) ...
#pragma unroll
		for (int r = 0; r < kEncRounds; r++) {
			const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
			if (round_row >= n + align) continue; // uniform
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			const bool interior = round_row >= align && round_row + ROUND_ROWS - align <= n; // uniform
			U v[K];
			__builtin_memcpy(v, &q[r], 16);
			if (!interior) { // rows outside the segment leave no bit
				const int32_t row0 = (int32_t)(c * K) - (int32_t)align;
#pragma unroll
				for (int j = 0; j < K; j++) {
					if ((uint32_t)(row0 + j) >= n) v[j] = sub;
				}
			}
			if (w != type_bits) { // (unpacked: the words are the rows)
				const Str128 str = chunk_string32<U>(v, (uint32_t)sub, (uint32_t)wmask, w);
				q[r] = make_uint4(str.s0, str.s1, str.s2, str.s3);
			} else if (!interior) {
				__builtin_memcpy(&q[r], v, 16);
			}
		}
		if (parked) {
			// ... are parked in LDS (every thread in slots of its own: no barrier) ...
#pragma unroll
			for (int r = 0; r < kEncRounds; r++) {
				const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
				if (round_row >= n + align) continue; // uniform
				uint32_t *slot = park + ((uint32_t)r * kEncThreads + tid) * 2u;
				if (nd == 2u) {
					*reinterpret_cast<uint2 *>(slot) = make_uint2(q[r].x, q[r].y);
				} else {
					slot[0] = q[r].x;
				}
			}
			img_dirty = true;
			stage_dirty = true;
			// ... the next segment's rows start travelling into the freed registers (wave 0 later: its look-back loads
			// would queue behind them) ...
			auto load_next = [&]() {
#pragma unroll
				for (int r = 0; r < kEncRounds; r++) {
					const uint32_t c = (uint32_t)r * kEncThreads + tid;
					q[r] = nxt.base16[c < nxt.last_chunk ? c : nxt.last_chunk];
				}
			};
			const bool more = nxt.seg < nseg; // uniform
			// Deferring pays when the NEXT segment parks too (then nothing but its analysis stands between this point and
			// the drain); next to segments of another flow it costs: on a column that alternates between the flows (u64
			// w = 16: a segment at an odd element offset cannot park) it ran 2.5x slower.  The next width is not known
			// yet; at w = 32 every segment parks whatever its placement, so that is where it is done.
			if (kPublishAhead && more && !first_come && (placement & 2) && w == 32u) { // uniform: publish-ahead (see `pend`)
				if (tid == 0) next_ticket = atomicAdd(ticket, 1u); // the segment after the next: before this wave's loads
				load_next();
				pend = Pending {true, seg, n, align, nchunks, nd, skip, nd_total, ana};
				next_loaded = true;
				break;
			}
			if (more && tid >= 64u) load_next();
			place(false);
			// ... and the strings leave for the arena
			if constexpr (kPublishAhead) {
				store_parked(reinterpret_cast<uint32_t *>(dst), n, align, nchunks, nd, skip, nd_total);
			} else {
				uint32_t *__restrict__ out32 = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
				for (int r = 0; r < kEncRounds; r++) {
					const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
					if (round_row >= n + align) continue; // uniform
					const uint32_t c = (uint32_t)r * kEncThreads + tid;
					const bool interior = round_row >= align && round_row + ROUND_ROWS - align <= n; // uniform
					const uint32_t *slot = park + ((uint32_t)r * kEncThreads + tid) * 2u;
					uint32_t s[2];
					if (nd == 2u) {
						const uint2 t2 = *reinterpret_cast<const uint2 *>(slot);
						s[0] = t2.x, s[1] = t2.y;
					} else {
						s[0] = slot[0], s[1] = 0u;
					}
					const uint32_t g0 = c * nd - skip;
					if (interior) {
						if (nd == 1u) {
							__builtin_nontemporal_store(s[0], out32 + g0);
						} else {
							store_dwords<2>(out32 + g0, s);
						}
					} else if (c < nchunks) {
#pragma unroll
						for (uint32_t j = 0; j < 2u; j++) {
							if (j < nd && g0 + j < nd_total) out32[g0 + j] = s[j];
						}
						if (c == nchunks - 1u && g0 + nd < nd_total) out32[nd_total - 1u] = 0u;
					}
				}
			}
			if (more && tid < 64u) {
				if (tid == 0) next_ticket = atomicAdd(ticket, 1u);
				load_next();
			}
			next_loaded = more;
			break;
		}
		place(true);
		// ... and leave for the arena
		uint32_t *__restrict__ dst32 = reinterpret_cast<uint32_t *>(dst);
#pragma unroll
		for (int r = 0; r < kEncRounds; r++) {
			const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
			if (round_row >= n + align) continue; // uniform
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			const bool interior = round_row >= align && round_row + ROUND_ROWS - align <= n; // uniform
			uint32_t s[4];
			__builtin_memcpy(s, &q[r], 16);
			const uint32_t g0 = c * nd - skip; // wraps for the chunk that holds rows before row 0: tested per dword
			if (interior) {
				uint32_t *__restrict__ o = dst32 + g0;
				if (nd == 1u) {
					o[0] = s[0];
				} else if (nd == 2u) {
					store_dwords<2>(o, s);
				} else if (nd == 3u) {
					store_dwords<3>(o, s);
				} else {
					store_dwords<4>(o, s);
				}
			} else if (c < nchunks) {
#pragma unroll
				for (uint32_t j = 0; j < 4u; j++) {
					if (j < nd && g0 + j < nd_total) dst32[g0 + j] = s[j];
				}
				// the upper half of the last word when the strings end one dword short of it
				if (c == nchunks - 1u && g0 + nd < nd_total) dst32[nd_total - 1u] = 0u;
			}
		}
		break;
	}

	unsigned long long *const imgp = bigimg ? pool64 : img; // (uniform)
	if (bigimg && (img_dirty || stage_dirty)) { // uniform: the pool holds prefetched rows or parked strings
		lds_barrier();
		for (uint32_t i = tid; i < kPoolWords + 4; i += kEncThreads) pool64[i] = 0ull;
		lds_barrier();
		img_dirty = false;
		stage_dirty = false;
	}
	if (img_dirty) { // uniform: a parked segment wrote over the image since it was last cleared
		lds_barrier();
		for (uint32_t i = tid; i < kEncImageWords + 4; i += kEncThreads) img[i] = 0ull;
		lds_barrier();
		img_dirty = false;
	}
	uint32_t rps = (kEncImageWords * 64u) / (ROUND_ROWS * w); // whole rounds per stage
	rps = rps < 1u ? 1u : rps;
	if (bigimg) rps = (uint32_t)kEncRounds; // the whole segment is ONE stage
	uint32_t stage_lo = 0; // first row of the current stage (a multiple of ROUND_ROWS)

	// rows [lo, hi) of chunk (row0 .. row0 + K) -> the stage image
	auto emit = [&](const uint4 &chunk, uint32_t c, uint32_t lo, uint32_t hi) {
		const int32_t row0 = (int32_t)(c * K) - (int32_t)align;
		U v[K];
		__builtin_memcpy(v, &chunk, 16);
		if (validity) {
			const uint32_t vbits = chunk_validity(validity, (seg_val_off - align) + (uint64_t)c * K);
#pragma unroll
			for (int j = 0; j < K; j++) v[j] = ((vbits >> j) & 1u) ? v[j] : (U)null_bits; // NullValue<T> (succinct.cpp:288-291)
		}
		const bool inside = row0 >= (int32_t)lo && (uint32_t)row0 + K <= hi;
		if (w <= 32u) { // 32-bit emission (chunk_string32): rows outside [lo, hi) become field 0 and leave no bit
			if (!inside) {
#pragma unroll
				for (int j = 0; j < K; j++) {
					const int32_t row = row0 + j;
					if (row < (int32_t)lo || row >= (int32_t)hi) v[j] = sub;
				}
			}
			// a chunk that starts before the stage (thread 0's, by `align` rows) keeps its string position: the fields
			// of the rows before `lo` are zero, and so are the bits they would have set — but the position must not be
			// negative, so such a chunk goes row by row below
			if (row0 >= (int32_t)stage_lo) {
				const Str128 str = chunk_string32<U>(v, (uint32_t)sub, (uint32_t)wmask, w);
				image_or32(reinterpret_cast<uint32_t *>(imgp), ((uint32_t)row0 - stage_lo) * w, str, (uint32_t)K * w);
				return;
			}
		}
		U f[K];
#pragma unroll
		for (int j = 0; j < K; j++) f[j] = (U)((U)(v[j] - sub) & wmask);
		if (inside) {
			uint64_t s_lo, s_hi;
			concat_fields<U>(f, w, s_lo, s_hi);
			image_or(imgp, ((uint32_t)row0 - stage_lo) * w, s_lo, s_hi);
		} else {
#pragma unroll
			for (int j = 0; j < K; j++) {
				const int32_t row = row0 + j;
				if (row >= (int32_t)lo && row < (int32_t)hi) image_or(imgp, ((uint32_t)row - stage_lo) * w, (uint64_t)f[j], 0ull);
			}
		}
	};
	// the image holds rows [stage_lo, hi): out to the arena, and zero again
	auto flush = [&](uint32_t hi) {
		if (!dst) place(true); // uniform: the first stage is in the image, now the arena offset is needed
		lds_barrier();
		const uint32_t nwords = ((hi - stage_lo) * w + 63u) >> 6;
		unsigned long long *__restrict__ out = dst + (((uint64_t)stage_lo * w) >> 6);
		for (uint32_t i = 2u * tid; i < nwords; i += 2u * kEncThreads) {
			const unsigned long long v0 = img[i], v1 = img[i + 1];
			img[i] = 0ull;
			img[i + 1] = 0ull;
			if (i + 1 < nwords) {
				uint4 o;
				o.x = (uint32_t)v0;
				o.y = (uint32_t)(v0 >> 32);
				o.z = (uint32_t)v1;
				o.w = (uint32_t)(v1 >> 32);
				nt_store16(out + i, o);
			} else {
				out[i] = v0;
			}
		}
		lds_barrier();
	};
	const uint32_t p_thread = (tid * K - align) * w; // bit position of this thread's chunk inside round 0 (wraps for
	                                                  // thread 0 of a misplaced segment: that chunk goes row by row)
#pragma unroll
	for (int r = 0; r < kEncRounds; r++) {
		const uint32_t round_row = (uint32_t)r * ROUND_ROWS;
		if (round_row >= n + align) continue; // uniform: no chunk of this or a later round holds a row
		const uint32_t c = (uint32_t)r * kEncThreads + tid;
		if (r > 0 && ((uint32_t)r % rps) == 0u && round_row < n) { // uniform: a stage ends before this round
			// thread 0's chunk of this round starts `align` rows before the boundary: those rows belong to the stage
			// that is about to be written out
			if (tid == 0 && align) emit(q[r], c, stage_lo, round_row);
			flush(round_row);
			stage_lo = round_row;
		}
		const uint32_t stage_hi = stage_lo + rps * ROUND_ROWS;
		// uniform: every chunk of the round lies inside the segment AND inside the stage (thread 0's chunk starts
		// `align` rows before the round): no row tests, no NULLs, 32-bit emission
		const bool interior = !validity && w <= 32u && round_row >= align && round_row + ROUND_ROWS - align <= n &&
		                      (align == 0u || round_row > stage_lo);
		if (interior) {
			U v[K];
			__builtin_memcpy(v, &q[r], 16);
			const Str128 str = chunk_string32<U>(v, (uint32_t)sub, (uint32_t)wmask, w);
			image_or32(reinterpret_cast<uint32_t *>(imgp), p_thread + (round_row - stage_lo) * w, str, (uint32_t)K * w);
		} else if (c < nchunks) {
			emit(q[r], c, stage_lo, stage_hi < n ? stage_hi : n);
		}
	}
	if (bigimg) { // the image stays in the pool: the next segment is loaded, analysed and published first
		if (tid == 0) {
			s_ipend.mn = mn, s_ipend.mx = mx, s_ipend.stored_min = stored_min, s_ipend.footprint = footprint;
			s_ipend.seg = seg, s_ipend.w = w, s_ipend.flags = flags, s_ipend.nwords = seg_words;
		}
		ipend_active = true;
	} else {
		flush(n);
	}
	} while (0);
	ADAC_STAMP(4);

	// ---- the next segment: its first rounds are in LDS (or on their way), the rest is loaded now, and the segment
	// after it is prefetched behind those loads
	if (nxt.seg >= nseg) break; // uniform: the column is done
	cur = nxt;
	if (next_loaded) { // uniform: the parked flow already issued this segment's loads
		next_loaded = false;
		continue;
	}
	if (tid == 0) next_ticket = atomicAdd(ticket, 1u);
	if (staged) { // uniform
#pragma unroll
		for (int r = 0; r < kEncPrefetch; r++) q[r] = stage[(uint32_t)r * kEncThreads + tid];
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the staged chunks are in registers before LDS-DMA may overwrite them
	} else { // (big image: nothing was prefetched, the pool holds the image)
#pragma unroll
		for (int r = 0; r < kEncPrefetch; r++) {
			const uint32_t c = (uint32_t)r * kEncThreads + tid;
			q[r] = cur.base16[c < cur.last_chunk ? c : cur.last_chunk];
		}
	}
#pragma unroll
	for (int r = kEncPrefetch; r < kEncRounds; r++) {
		const uint32_t c = (uint32_t)r * kEncThreads + tid;
		q[r] = cur.base16[c < cur.last_chunk ? c : cur.last_chunk];
	}
	}
}
