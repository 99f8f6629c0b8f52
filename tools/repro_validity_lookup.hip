// repro_validity_lookup.hip — 60-line stand-alone form of the FIRST k_analyze (round 1, commit 15cc986,
// csrc/adac_kernels.hip:287-330): the validity bit of every row looked up through row_valid() inside the unrolled
// chunk loop.  On the round-1 GPU run that form failed test_nulls_validity_mask (rows in slot 0 of a chunk took part in
// min/max although their validity bit was clear); it was replaced by one mask-word load per chunk (chunk_validity).
// This program checks whether the pattern miscompiles in isolation with the installed hipcc:
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/repro tools/repro_validity_lookup.hip && /tmp/repro
// It prints per-slot counts of wrongly included rows; "PASS" means the stand-alone form is compiled correctly.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int kWorkgroup = 256;
__device__ __forceinline__ bool row_valid(const uint64_t *__restrict__ validity, uint64_t elem) {
	return validity == nullptr || ((validity[elem >> 6] >> (elem & 63)) & 1ull);
}
template <typename U>
__global__ __launch_bounds__(kWorkgroup) void k_analyze_old(const U *__restrict__ vals, uint64_t elem0, uint32_t n,
                                                            const uint64_t *__restrict__ validity, int sign_extend,
                                                            uint64_t null_bits, int rule, uint64_t *__restrict__ minmax) {
	constexpr int K = 16 / (int)sizeof(U);
	using S = typename std::make_signed<U>::type;
	const U *src = vals + elem0;
	const uint32_t align = (uint32_t)(elem0 & (K - 1));
	uint64_t mn = ~0ull, mx = 0;
	for (uint32_t c = threadIdx.x + blockIdx.x * kWorkgroup; c * K < n + align; c += kWorkgroup * gridDim.x) {
		const int32_t base = (int32_t)(c * K) - (int32_t)align;
		U v[K];
		if (base >= 0 && (uint32_t)(base + K) <= n) {
			const uint4 q = *reinterpret_cast<const uint4 *>(src + base);
			__builtin_memcpy(v, &q, 16);
		} else {
#pragma unroll
			for (int j = 0; j < K; j++) v[j] = (uint32_t)(base + j) < n ? src[base + j] : (U)0;
		}
#pragma unroll
		for (int j = 0; j < K; j++) {
			if ((uint32_t)(base + j) >= n) continue;
			const bool valid = row_valid(validity, elem0 + (int64_t)(base + j));
			uint64_t x;
			if (rule == 0) {
				if (!valid) continue;
				x = sign_extend ? (uint64_t)(int64_t)(S)v[j] : (uint64_t)v[j];
			} else {
				x = valid ? (uint64_t)v[j] : null_bits;
			}
			mn = x < mn ? x : mn;
			mx = x > mx ? x : mx;
		}
	}
	atomicMin(reinterpret_cast<unsigned long long *>(minmax), (unsigned long long)mn);
	atomicMax(reinterpret_cast<unsigned long long *>(minmax + 1), (unsigned long long)mx);
}

template <typename U>
static int run(const char *name, int sign_extend) {
	const uint32_t n = 100003;
	const uint64_t elem0 = 3; // misaligned placement, as in the failing test
	std::vector<U> vals(n + 16);
	std::vector<uint64_t> valid((n + 16 + 63) / 64 + 1, ~0ull);
	srand(7);
	int bad = 0;
	for (int slot = 0; slot < (int)(16 / sizeof(U)); slot++) {
		// every row holds a mid-range value; ONE invalid row in chunk slot `slot` holds the extreme: if it takes part
		// the maximum is wrong
		for (auto &v : vals) v = (U)(1000 + rand() % 1000);
		for (auto &w : valid) w = ~0ull;
		const uint32_t K = 16 / sizeof(U);
		const uint32_t row = 4096 * K + ((slot + K - (uint32_t)(elem0 % K)) % K); // element index = elem0 + row in slot `slot`
		vals[elem0 + row] = (U)30000;
		valid[(elem0 + row) >> 6] &= ~(1ull << ((elem0 + row) & 63));
		U *d_vals;
		uint64_t *d_valid, *d_mm, mm[2] = {~0ull, 0};
		hipMalloc(&d_vals, vals.size() * sizeof(U));
		hipMalloc(&d_valid, valid.size() * 8);
		hipMalloc(&d_mm, 16);
		hipMemcpy(d_vals, vals.data(), vals.size() * sizeof(U), hipMemcpyHostToDevice);
		hipMemcpy(d_valid, valid.data(), valid.size() * 8, hipMemcpyHostToDevice);
		hipMemcpy(d_mm, mm, 16, hipMemcpyHostToDevice);
		hipLaunchKernelGGL(k_analyze_old<U>, dim3(8), dim3(kWorkgroup), 0, 0, d_vals, elem0, n, d_valid, sign_extend, 0ull, 0, d_mm);
		hipMemcpy(mm, d_mm, 16, hipMemcpyDeviceToHost);
		const bool wrong = mm[1] >= 30000;
		printf("%s slot %d: max %llu %s\n", name, slot, (unsigned long long)mm[1], wrong ? "WRONG (NULL row took part)" : "ok");
		bad += wrong;
		hipFree(d_vals);
		hipFree(d_valid);
		hipFree(d_mm);
	}
	return bad;
}

int main() {
	int bad = run<uint32_t>("u32", 0) + run<uint16_t>("u16", 0) + run<uint64_t>("u64", 0) + run<uint8_t>("u8", 0) +
	          run<uint32_t>("i32", 1);
	printf(bad ? "FAIL: %d slots wrong\n" : "PASS (%d wrong): the stand-alone form compiles correctly\n", bad);
	return 0;
}
