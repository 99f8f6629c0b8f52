// Read-only HBM bandwidth ceiling probe for MI355X (diagnostic, not part of the library): sums a buffer with
// 16-byte loads in a few launch shapes so the fused scans' GB/s can be read against what a bare streaming read
// achieves on the same box.   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/read_bw_probe tools/read_bw_probe.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                                          \
	do {                                                                                                               \
		hipError_t e = (x);                                                                                            \
		if (e != hipSuccess) {                                                                                         \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                \
			return 1;                                                                                                  \
		}                                                                                                              \
	} while (0)

// each workgroup streams `chunks_per_wg` consecutive 16-byte chunks, 256 lanes wide, UNROLL loads in flight
template <int UNROLL>
__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ src, uint64_t nchunks, uint32_t chunks_per_wg,
                                                uint64_t *__restrict__ out) {
	const uint64_t lo = (uint64_t)blockIdx.x * chunks_per_wg;
	uint64_t hi = lo + chunks_per_wg;
	hi = hi < nchunks ? hi : nchunks;
	uint32_t acc = 0;
	uint64_t c = lo + threadIdx.x;
	for (; c + (uint64_t)(UNROLL - 1) * 256 < hi; c += (uint64_t)UNROLL * 256) {
		uint4 q[UNROLL];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) q[u] = src[c + (uint64_t)u * 256];
#pragma unroll
		for (int u = 0; u < UNROLL; u++) acc += q[u].x ^ q[u].y ^ q[u].z ^ q[u].w;
	}
	for (; c < hi; c += 256) {
		const uint4 q = src[c];
		acc += q.x ^ q.y ^ q.z ^ q.w;
	}
	if (acc == 0x12345678u) out[0] = acc; // keeps the loads alive
}

// decode-shaped traffic: every 16 bytes read become 32 bytes written (w = 32 -> u64), 256 lanes wide
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void put16(uint4 *p, uint4 v) {
	if (NT) {
		v4u q = {v.x, v.y, v.z, v.w};
		__builtin_nontemporal_store(q, reinterpret_cast<v4u *>(p));
	} else {
		*p = v;
	}
}
template <bool NT>
__global__ __launch_bounds__(256) void k_expand(const uint4 *__restrict__ src, uint64_t nchunks, uint32_t chunks_per_wg,
                                                uint4 *__restrict__ dst) {
	const uint64_t lo = (uint64_t)blockIdx.x * chunks_per_wg;
	uint64_t hi = lo + chunks_per_wg;
	hi = hi < nchunks ? hi : nchunks;
	for (uint64_t c = lo + threadIdx.x; c < hi; c += 256) {
		const uint4 q = src[c];
		// lane-contiguous 32 bytes: two dwordx4 stores, like StoreSink's output chunks of two rounds
		put16<NT>(dst + 2 * c, make_uint4(q.x, 0u, q.y, 0u));
		put16<NT>(dst + 2 * c + 1, make_uint4(q.z, 0u, q.w, 0u));
	}
}
// write-only traffic (the store ceiling), plain or non-temporal
template <bool NT>
__global__ __launch_bounds__(256) void k_fill(uint64_t nchunks, uint32_t chunks_per_wg, uint4 *__restrict__ dst) {
	const uint64_t lo = (uint64_t)blockIdx.x * chunks_per_wg;
	uint64_t hi = lo + chunks_per_wg;
	hi = hi < nchunks ? hi : nchunks;
	for (uint64_t c = lo + threadIdx.x; c < hi; c += 256) put16<NT>(dst + c, make_uint4((uint32_t)c, 1u, 2u, 3u));
}

int main() {
	const uint64_t bytes = 1600ull << 20; // beyond the 256 MiB Infinity Cache
	void *d = nullptr;
	uint64_t *d_out = nullptr;
	CK(hipMalloc(&d, bytes));
	CK(hipMalloc((void **)&d_out, 8));
	CK(hipMemset(d, 1, bytes));
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	const uint64_t nchunks = bytes / 16;
	std::printf("{\"bytes\": %llu, \"cases\": [", (unsigned long long)bytes);
	bool first = true;
	for (uint32_t kb_per_wg : {4u, 16u, 32u, 64u, 128u, 512u, 2048u}) {
		for (int unroll : {1, 2, 4}) {
			const uint32_t cpw = kb_per_wg * 1024 / 16;
			const unsigned grid = (unsigned)((nchunks + cpw - 1) / cpw);
			auto launch = [&]() {
				if (unroll == 1) hipLaunchKernelGGL(k_stream<1>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d, nchunks, cpw, d_out);
				if (unroll == 2) hipLaunchKernelGGL(k_stream<2>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d, nchunks, cpw, d_out);
				if (unroll == 4) hipLaunchKernelGGL(k_stream<4>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d, nchunks, cpw, d_out);
			};
			for (int i = 0; i < 3; i++) launch();
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0, 0));
			const int reps = 10;
			for (int i = 0; i < reps; i++) launch();
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			const double gbps = (double)bytes * reps / (ms * 1e-3) / 1e9;
			std::printf("%s{\"kb_per_wg\": %u, \"loads_in_flight\": %d, \"workgroups\": %u, \"GBps\": %.0f}", first ? "" : ", ",
			            kb_per_wg, unroll, grid, gbps);
			first = false;
		}
	}
	for (int nt = 0; nt < 2; nt++) {
	std::printf("], \"%s\": [", nt ? "expand_1_to_2_nontemporal_stores" : "expand_1_to_2");
	{
		const uint64_t rd = 400ull << 20; // C2-like: 400 MB read, 800 MB written
		void *d_dst = nullptr;
		CK(hipMalloc(&d_dst, 2 * rd));
		const uint64_t nch = rd / 16;
		first = true;
		for (uint32_t kb_per_wg : {2u, 4u, 8u, 16u, 64u}) {
			const uint32_t cpw = kb_per_wg * 1024 / 16;
			const unsigned grid = (unsigned)((nch + cpw - 1) / cpw);
			auto go = [&]() {
				if (nt) hipLaunchKernelGGL(k_expand<true>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d, nch, cpw, (uint4 *)d_dst);
				else hipLaunchKernelGGL(k_expand<false>, dim3(grid), dim3(256), 0, 0, (const uint4 *)d, nch, cpw, (uint4 *)d_dst);
			};
			for (int i = 0; i < 3; i++) go();
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0, 0));
			const int reps = 10;
			for (int i = 0; i < reps; i++) go();
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			std::printf("%s{\"read_kb_per_wg\": %u, \"workgroups\": %u, \"total_GBps\": %.0f}", first ? "" : ", ", kb_per_wg,
			            grid, 3.0 * (double)rd * reps / (ms * 1e-3) / 1e9);
			first = false;
		}
		CK(hipFree(d_dst));
	}
	}
	for (int nt = 0; nt < 2; nt++) { // write-only: 800 MB
		std::printf("], \"%s\": [", nt ? "fill_nontemporal" : "fill");
		const uint64_t wr = 800ull << 20;
		const uint64_t nch = wr / 16;
		first = true;
		for (uint32_t kb_per_wg : {4u, 16u, 64u}) {
			const uint32_t cpw = kb_per_wg * 1024 / 16;
			const unsigned grid = (unsigned)((nch + cpw - 1) / cpw);
			auto go = [&]() {
				if (nt) hipLaunchKernelGGL(k_fill<true>, dim3(grid), dim3(256), 0, 0, nch, cpw, (uint4 *)d);
				else hipLaunchKernelGGL(k_fill<false>, dim3(grid), dim3(256), 0, 0, nch, cpw, (uint4 *)d);
			};
			for (int i = 0; i < 3; i++) go();
			CK(hipDeviceSynchronize());
			CK(hipEventRecord(e0, 0));
			const int reps = 10;
			for (int i = 0; i < reps; i++) go();
			CK(hipEventRecord(e1, 0));
			CK(hipEventSynchronize(e1));
			float ms = 0;
			CK(hipEventElapsedTime(&ms, e0, e1));
			std::printf("%s{\"kb_per_wg\": %u, \"GBps\": %.0f}", first ? "" : ", ", kb_per_wg, (double)wr * reps / (ms * 1e-3) / 1e9);
			first = false;
		}
	}
	std::printf("]}\n");
	return 0;
}
