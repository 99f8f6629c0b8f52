// C entry point over the reference's REAL Zipf sampler (benchmark/micro/succinct/zipf.cpp: a self-contained class
// template over <random>), compiled from where it lies into oracle/_ref/libzipf_ref.so.  It pins the INPUT of the
// headline configurations (C2 / C4: Zipf-distributed columns, benchmark/micro/succinct/zipf_distribution.cpp:29-37) —
// the harness's own generator (duckdb-adaptive-compression_amd/csrc/workload.c) must draw the same values from the same
// seed.  Test infrastructure only; contains no reference code — it only includes and calls it.
#include <cstdint>
#include <random>

#include "zipf.cpp"

extern "C" __attribute__((visibility("default"))) void ref_zipf_draws(uint32_t seed, uint32_t n, double q, uint64_t count,
                                                                      uint64_t *out) {
	std::mt19937 gen {seed};               // zipf_distribution.cpp:30 seeds it from std::random_device; the harness fixes it
	Zipf<uint32_t, double> zipf(n, q);     // zipf_distribution.cpp:31
	for (uint64_t i = 0; i < count; i++) out[i] = zipf(gen);
}
