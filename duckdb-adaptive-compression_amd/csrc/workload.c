/*
 * workload.c — synthetic column generators for the benchmark harness (host code, no device).
 *
 * Reproduces the data shapes of the reference's succinct micro-benchmarks
 * (benchmark/micro/succinct/*.cpp): Zipf-distributed integers drawn with the rejection-inversion sampler
 * the reference uses (benchmark/micro/succinct/zipf.cpp:12-100 — W. Hörmann, G. Derflinger,
 * "Rejection-inversion to generate variates from monotone discrete distributions", ACM TOMACS 6(3), 1996)
 * on top of a 32-bit Mersenne Twister (std::mt19937; Matsumoto & Nishimura 1998) feeding a
 * uniform_real_distribution<double> (two 32-bit draws per double, as libstdc++'s generate_canonical does).
 * The reference seeds from std::random_device (zipf_distribution.cpp:29-30); the harness fixes seed 42 so
 * runs are reproducible.  Large columns are generated in blocks of 2^20 values, block b
 * seeded with seed + b, so the result does not depend on the number of threads.
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>

#define WL_API __attribute__((visibility("default")))

/* ---- MT19937 ---------------------------------------------------------------------------------- */
typedef struct {
	uint32_t s[624];
	int idx;
} mt_t;

static void mt_seed(mt_t *m, uint32_t seed) {
	m->s[0] = seed;
	for (int i = 1; i < 624; i++) {
		m->s[i] = 1812433253u * (m->s[i - 1] ^ (m->s[i - 1] >> 30)) + (uint32_t)i;
	}
	m->idx = 624;
}

static uint32_t mt_next(mt_t *m) {
	if (m->idx >= 624) {
		for (int i = 0; i < 624; i++) {
			uint32_t y = (m->s[i] & 0x80000000u) | (m->s[(i + 1) % 624] & 0x7fffffffu);
			uint32_t v = m->s[(i + 397) % 624] ^ (y >> 1);
			if (y & 1u) v ^= 0x9908b0dfu;
			m->s[i] = v;
		}
		m->idx = 0;
	}
	uint32_t y = m->s[m->idx++];
	y ^= y >> 11;
	y ^= (y << 7) & 0x9d2c5680u;
	y ^= (y << 15) & 0xefc60000u;
	y ^= y >> 18;
	return y;
}

/* uniform double in [a, b): 53 bits from two 32-bit draws */
static double mt_uniform(mt_t *m, double a, double b) {
	double lo = (double)mt_next(m);
	double hi = (double)mt_next(m);
	double r = (lo + hi * 4294967296.0) / 18446744073709551616.0;
	if (r >= 1.0) r = nextafter(1.0, 0.0);
	return a + r * (b - a);
}

/* ---- rejection-inversion Zipf ----------------------------------------------------------------- */
typedef struct {
	double n, q, h_x1, h_n;
} zipf_t;

static const double kEps = 1e-8;

/* (exp(x) - 1) / x, stable near 0 */
static double expm1_over_x(double x) {
	return fabs(x) > kEps ? expm1(x) / x : (1.0 + x / 2.0 * (1.0 + x / 3.0 * (1.0 + x / 4.0)));
}
/* log(1 + x) / x, stable near 0 */
static double log1p_over_x(double x) {
	return fabs(x) > kEps ? log1p(x) / x : 1.0 - x * (0.5 - x * (1.0 / 3.0 - x * 0.25));
}
/* H(x): an antiderivative of the hat h(x) = x^-q, written so that q == 1 needs no special case */
static double zipf_H(const zipf_t *z, double x) {
	double lx = log(x);
	return expm1_over_x((1.0 - z->q) * lx) * lx;
}
static double zipf_H_inv(const zipf_t *z, double x) {
	double t = x * (1.0 - z->q);
	if (t < -1.0) t = -1.0;
	return exp(log1p_over_x(t) * x);
}
static double zipf_h(const zipf_t *z, double x) { return exp(-z->q * log(x)); }

static void zipf_init(zipf_t *z, double n, double q) {
	z->n = n;
	z->q = q;
	z->h_x1 = zipf_H(z, 1.5) - 1.0;
	z->h_n = zipf_H(z, n + 0.5);
}

static uint64_t zipf_draw(const zipf_t *z, mt_t *m) {
	for (;;) {
		double u = mt_uniform(m, z->h_x1, z->h_n);
		double x = zipf_H_inv(z, u);
		double k = round(x);
		if (k < 1.0) k = 1.0;
		if (k > z->n) k = z->n;
		if (u >= zipf_H(z, k + 0.5) - zipf_h(z, k)) return (uint64_t)k;
	}
}

/* ---- block-parallel fill ----------------------------------------------------------------------- */
#define WL_BLOCK (1u << 20)

typedef struct {
	void *out;
	uint64_t n;
	unsigned elem_size;
	double domain, q;
	uint32_t seed;
	uint64_t base; /* value = base + draw, draw in [1, domain] */
	int tid, nthreads;
} fill_job;

static void store_val(void *out, uint64_t i, unsigned es, uint64_t v) {
	switch (es) {
	case 1: ((uint8_t *)out)[i] = (uint8_t)v; break;
	case 2: ((uint16_t *)out)[i] = (uint16_t)v; break;
	case 4: ((uint32_t *)out)[i] = (uint32_t)v; break;
	default: ((uint64_t *)out)[i] = v; break;
	}
}

static void *fill_worker(void *arg) {
	fill_job *j = (fill_job *)arg;
	zipf_t z;
	zipf_init(&z, j->domain, j->q);
	uint64_t nblocks = (j->n + WL_BLOCK - 1) / WL_BLOCK;
	for (uint64_t b = (uint64_t)j->tid; b < nblocks; b += (uint64_t)j->nthreads) {
		mt_t m;
		mt_seed(&m, j->seed + (uint32_t)b);
		uint64_t lo = b * WL_BLOCK, hi = lo + WL_BLOCK;
		if (hi > j->n) hi = j->n;
		for (uint64_t i = lo; i < hi; i++) {
			store_val(j->out, i, j->elem_size, j->base + zipf_draw(&z, &m));
		}
	}
	return NULL;
}

/* out[i] = base + Zipf(domain, q) draw in [1, domain]; elem_size in {1,2,4,8}. */
WL_API int adacw_zipf_fill(void *out, uint64_t n, unsigned elem_size, double domain, double q, uint64_t base,
                           uint32_t seed, int threads) {
	if (!out || domain < 1.0 || q < 0.0) return 1;
	if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8) return 1;
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	pthread_t tid[256];
	fill_job jobs[256];
	for (int t = 0; t < threads; t++) {
		jobs[t] = (fill_job){out, n, elem_size, domain, q, seed, base, t, threads};
		if (threads == 1) {
			fill_worker(&jobs[t]);
		} else if (pthread_create(&tid[t], NULL, fill_worker, &jobs[t]) != 0) {
			return 2;
		}
	}
	if (threads > 1) {
		for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
	}
	return 0;
}

/* Rows [row_lo, row_hi) of the GLOBAL column adacw_zipf_fill(seed) defines, written to out[0 ...): a shard's
 * slice of one column (block b of 2^20 rows is its own mt19937 stream seeded seed + b; a slice that starts inside
 * a block replays that block's draws up to its first row, so the values equal the global column's). */
typedef struct {
	void *out;
	uint64_t row_lo, row_hi;
	unsigned elem_size;
	double domain, q;
	uint32_t seed;
	uint64_t base;
	int tid, nthreads;
} range_job;

static void *range_worker(void *arg) {
	range_job *j = (range_job *)arg;
	zipf_t z;
	zipf_init(&z, j->domain, j->q);
	uint64_t b0 = j->row_lo / WL_BLOCK, b1 = (j->row_hi + WL_BLOCK - 1) / WL_BLOCK;
	for (uint64_t b = b0 + (uint64_t)j->tid; b < b1; b += (uint64_t)j->nthreads) {
		mt_t m;
		mt_seed(&m, j->seed + (uint32_t)b);
		uint64_t lo = b * WL_BLOCK, hi = lo + WL_BLOCK;
		if (hi > j->row_hi) hi = j->row_hi;
		for (uint64_t i = lo; i < hi; i++) {
			uint64_t v = j->base + zipf_draw(&z, &m);
			if (i >= j->row_lo) store_val(j->out, i - j->row_lo, j->elem_size, v);
		}
	}
	return NULL;
}

WL_API int adacw_zipf_fill_range(void *out, uint64_t row_lo, uint64_t row_hi, unsigned elem_size, double domain,
                                 double q, uint64_t base, uint32_t seed, int threads) {
	if (!out || domain < 1.0 || q < 0.0 || row_hi < row_lo) return 1;
	if (elem_size != 1 && elem_size != 2 && elem_size != 4 && elem_size != 8) return 1;
	if (threads < 1) threads = 1;
	if (threads > 256) threads = 256;
	pthread_t tid[256];
	range_job jobs[256];
	for (int t = 0; t < threads; t++) {
		jobs[t] = (range_job){out, row_lo, row_hi, elem_size, domain, q, seed, base, t, threads};
		if (threads == 1) {
			range_worker(&jobs[t]);
		} else if (pthread_create(&tid[t], NULL, range_worker, &jobs[t]) != 0) {
			return 2;
		}
	}
	if (threads > 1) {
		for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
	}
	return 0;
}

/* The first `n` raw mt19937 outputs for a seed (self-test against the published reference stream). */
WL_API void adacw_mt19937_stream(uint32_t seed, uint32_t *out, uint64_t n) {
	mt_t m;
	mt_seed(&m, seed);
	for (uint64_t i = 0; i < n; i++) out[i] = mt_next(&m);
}
