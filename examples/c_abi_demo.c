/* c_abi_demo.c — the drop-in boundary used from plain C, no Python, no torch, no HIP headers:
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Lduckdb-adaptive-compression_amd -ladacodec \
 *       -Wl,-rpath,$PWD/duckdb-adaptive-compression_amd -o /tmp/c_abi_demo && /tmp/c_abi_demo
 * Packs a small uint32 column of three ragged segments, scans it back, runs a filter + masked SUM on the packed
 * bytes and materialises the selected rows.  Exits 0 only if every result is right. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "adacodec.h"

#define CHECK(call)                                                                                                    \
	do {                                                                                                               \
		adac_status st_ = (call);                                                                                      \
		if (st_ != ADAC_OK) {                                                                                          \
			fprintf(stderr, "%s -> %s (%s)\n", #call, adac_status_string(st_), adac_last_error());                     \
			return 2;                                                                                                  \
		}                                                                                                              \
	} while (0)

int main(void) {
	enum { NSEG = 3 };
	const uint32_t counts[NSEG] = {70000, 5, 12345};
	uint64_t n = 0;
	for (int s = 0; s < NSEG; s++) n += counts[s];
	uint32_t *vals = (uint32_t *)malloc(n * sizeof(uint32_t));
	for (uint64_t i = 0; i < n; i++) vals[i] = 1000000u + (uint32_t)((i * 2654435761u) % 5000u);

	adac_ctx *ctx = NULL;
	adac_status st = adac_ctx_create(0, NULL, &ctx);
	if (st == ADAC_ERR_NO_DEVICE) {
		fprintf(stderr, "no HIP device: the codec has no CPU fallback\n");
		return 3;
	}
	CHECK(st);
	adac_layout *col = NULL;
	CHECK(adac_layout_create(ctx, ADAC_UINT32, counts, NULL, NSEG, &col));
	void *d_vals = NULL, *d_words = NULL, *d_out = NULL, *d_bitmap = NULL, *d_counts = NULL, *d_sums = NULL, *d_sel = NULL;
	CHECK(adac_dev_alloc(ctx, n * 4 + 16, &d_vals));
	CHECK(adac_dev_alloc(ctx, adac_layout_max_arena_words(col) * 8 + 16, &d_words));
	CHECK(adac_dev_alloc(ctx, n * 4 + 16, &d_out));
	CHECK(adac_dev_alloc(ctx, (n + 63) / 64 * 8, &d_bitmap));
	CHECK(adac_dev_alloc(ctx, NSEG * 8, &d_counts));
	CHECK(adac_dev_alloc(ctx, NSEG * 8, &d_sums));
	CHECK(adac_dev_alloc(ctx, n * 4 + 16, &d_sel));
	CHECK(adac_memcpy_h2d(ctx, d_vals, vals, n * 4));

	/* encode: min/max -> widths -> bit-pack (BitCompressFromSuccinct) */
	CHECK(adac_encode(col, d_vals, NULL, ADAC_RULE_APPEND, 0, (uint64_t *)d_words));
	adac_segment_desc descs[NSEG];
	CHECK(adac_layout_get_descs(col, descs));
	for (int s = 0; s < NSEG; s++) {
		printf("segment %d: %u rows, width %u, min %llu, %llu bytes (sdsl size_in_bytes)\n", s, descs[s].count,
		       (unsigned)descs[s].width, (unsigned long long)descs[s].min,
		       (unsigned long long)adac_size_in_bytes(descs[s].count, descs[s].width));
		if (descs[s].width > 13 || !(descs[s].flags & ADAC_SEG_PACKED)) return 4; /* range 5000 needs <= 13 bits */
	}

	/* full scan (SuccinctScan) */
	CHECK(adac_unpack(col, (const uint64_t *)d_words, d_out));
	uint32_t *back = (uint32_t *)malloc(n * 4);
	CHECK(adac_memcpy_d2h(ctx, back, d_out, n * 4));
	if (memcmp(back, vals, n * 4) != 0) return 5;

	/* WHERE v BETWEEN 1001000 AND 1001999 -> bitmap; SUM(v) under it; the selected rows themselves */
	CHECK(adac_scan_select_between(col, (const uint64_t *)d_words, NULL, 1001000u, 1001999u, (uint64_t *)d_bitmap,
	                               (uint64_t *)d_counts));
	CHECK(adac_scan_sum_valid(col, (const uint64_t *)d_words, (const uint64_t *)d_bitmap, (uint64_t *)d_sums));
	uint64_t nsel = 0;
	CHECK(adac_unpack_selected(col, (const uint64_t *)d_words, (const uint64_t *)d_bitmap, d_sel, NULL, &nsel));
	uint64_t cnt[NSEG], sum[NSEG], exp_cnt = 0, exp_sum = 0, got_cnt = 0, got_sum = 0;
	CHECK(adac_memcpy_d2h(ctx, cnt, d_counts, sizeof cnt));
	CHECK(adac_memcpy_d2h(ctx, sum, d_sums, sizeof sum));
	for (uint64_t i = 0; i < n; i++) {
		if (vals[i] >= 1001000u && vals[i] <= 1001999u) {
			exp_cnt++;
			exp_sum += vals[i];
		}
	}
	for (int s = 0; s < NSEG; s++) {
		got_cnt += cnt[s];
		got_sum += sum[s];
	}
	printf("selected %llu rows (expected %llu), sum %llu (expected %llu), materialised %llu\n",
	       (unsigned long long)got_cnt, (unsigned long long)exp_cnt, (unsigned long long)got_sum,
	       (unsigned long long)exp_sum, (unsigned long long)nsel);
	if (got_cnt != exp_cnt || got_sum != exp_sum || nsel != exp_cnt) return 6;
	CHECK(adac_memcpy_d2h(ctx, back, d_sel, nsel * 4));
	uint64_t k = 0;
	for (uint64_t i = 0; i < n; i++) {
		if (vals[i] >= 1001000u && vals[i] <= 1001999u && back[k++] != vals[i]) return 7;
	}

	adac_dev_free(ctx, d_vals);
	adac_dev_free(ctx, d_words);
	adac_dev_free(ctx, d_out);
	adac_dev_free(ctx, d_bitmap);
	adac_dev_free(ctx, d_counts);
	adac_dev_free(ctx, d_sums);
	adac_dev_free(ctx, d_sel);
	adac_layout_destroy(col);
	adac_ctx_destroy(ctx);
	free(vals);
	free(back);
	printf("ok\n");
	return 0;
}
