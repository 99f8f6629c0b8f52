// adac_capi.cpp — the C ABI of libadacodec (include/adacodec.h): contexts, layouts (tile tables and
// descriptor tables in HBM) and the enqueue functions of the hot path.  Host logic only; the arithmetic
// lives in adac_kernels.hip.  There is deliberately no CPU implementation behind any device entry point.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "adac_internal.h"
#include "adacodec.h"

using adac::RangeArgs;
using adac::TileRef;

static thread_local std::string g_last_error;

static adac_status fail_hip(hipError_t e, const char *what) {
	g_last_error = std::string(what) + ": " + hipGetErrorString(e);
	if (e == hipErrorOutOfMemory) return ADAC_ERR_OUT_OF_MEMORY;
	if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return ADAC_ERR_NO_DEVICE;
	return ADAC_ERR_DEVICE;
}

#define ADAC_HIP(expr)                                                                                                 \
	do {                                                                                                               \
		hipError_t _e = (expr);                                                                                        \
		if (_e != hipSuccess) return fail_hip(_e, #expr);                                                              \
	} while (0)

struct adac_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool owns_stream = false;
	hipEvent_t ev_start = nullptr, ev_stop = nullptr;
	// one reference for the creator + one per layout / plan / graph made on the context: objects may be destroyed
	// in any order, the stream goes away with the last of them
	std::atomic<int> refs {1};
};

static void ctx_retain(adac_ctx *c) { c->refs.fetch_add(1); }
static void ctx_release(adac_ctx *c) {
	if (c->refs.fetch_sub(1) != 1) return;
	(void)hipSetDevice(c->device);
	if (c->ev_start) (void)hipEventDestroy(c->ev_start);
	if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
	if (c->owns_stream && c->stream) (void)hipStreamDestroy(c->stream);
	delete c;
}

struct adac_layout {
	adac_ctx *ctx = nullptr;
	int type = 0;
	uint32_t type_size = 0;
	bool is_signed = false;
	uint64_t null_bits = 0; // NullValue<T>() as the bit pattern of T, zero-extended (null_value.hpp:26-28)
	uint64_t nseg = 0, ntiles = 0, total_values = 0, value_span = 0, max_arena_words = 0;
	bool single_pass_ok = true; // every segment fits the single-pass encode kernel's registers
	void *d_scan_state = nullptr; // its look-back words + ticket (allocated on first use)
	bool dense_values = true; // segments back to back from element 0: no element index between them is unowned
	bool has_empty_segments = false; // a segment without rows has no scan group: its result is cleared, not stored
	// the widest segment among the descriptors that last passed through the host (adac_layout_get_descs / _set_descs), 0 =
	// never seen: a HINT for adac_encode's choice of form on 4-byte columns (performance only: every form is correct)
	uint32_t hint_max_width = 0;
	std::vector<uint32_t> counts;
	std::vector<uint64_t> val_offs;
	adac_segment_desc *d_descs = nullptr;
	TileRef *d_tiles = nullptr;
	uint64_t *d_minmax = nullptr;
	// fused-scan work items (see ScanGroup): built lazily for the current scan_tiles_per_wg, re-expanded when
	// the descriptors changed
	adac::ScanGroupRef *d_group_refs = nullptr;
	adac::ScanGroup *d_groups = nullptr;
	uint64_t ngroups = 0;
	int groups_tiles = 0;
	bool groups_dirty = true;
	adac::TileRec *d_tile_recs = nullptr; // the tiles with their segments' current descriptors folded in (decode side)
	bool tile_recs_dirty = true;
	// the groups of segments at widths 2 and 3, which a scan kernel of their own takes (k_scan_agg<.., NARROW>): the
	// expansion lists them on the device and their number comes back through a page-locked word; the first scan after
	// an expansion waits for that copy (it would wait for the same stream work in its own launch anyway)
	uint32_t *d_narrow_idx = nullptr, *d_narrow_count = nullptr, *h_narrow_count = nullptr;
	hipEvent_t narrow_ev = nullptr;
	bool narrow_pending = false;
	// scratch of adac_unpack_selected: selected rows per tile, their exclusive prefix, block totals + grand total
	uint32_t *d_tile_cnt = nullptr;
	uint64_t *d_tile_off = nullptr;
	uint64_t *d_block_tot = nullptr;
	// arrival cells of the fused scans (ScanGroupRef): zero between calls
	unsigned long long *d_res_cells = nullptr;
	uint32_t *d_edge_cells = nullptr;
	void *d_sel_edges = nullptr;     // shared-word records of the selection scan (two per scan group)
	uint64_t sel_edges_groups = 0;   // ... sized for this many groups
	void *d_group_partial = nullptr; // per-workgroup partials of adac_scan_group_sum (allocated on first use)
	uint32_t group_calls = 0;        // ... and how often it ran: its hand-over word alternates between two slots
};

static adac_status descs_changed(adac_layout *l);
static void note_widths(adac_layout *l, const adac_segment_desc *descs);
static adac_status ensure_scan_groups(adac_layout *l);

// ------------------------------------------------------------------------------------------------
// host-only helpers
// ------------------------------------------------------------------------------------------------

extern "C" int adac_abi_version(void) { return ADAC_ABI_VERSION; }

extern "C" const char *adac_status_string(adac_status s) {
	switch (s) {
	case ADAC_OK: return "ok";
	case ADAC_ERR_INVALID_ARGUMENT: return "invalid argument";
	case ADAC_ERR_UNSUPPORTED_TYPE: return "unsupported physical type for the succinct codec";
	case ADAC_ERR_DEVICE: return "HIP device error";
	case ADAC_ERR_OUT_OF_MEMORY: return "out of device memory";
	case ADAC_ERR_NO_DEVICE: return "no usable gfx950 device";
	}
	return "unknown status";
}

extern "C" const char *adac_last_error(void) { return g_last_error.c_str(); }

extern "C" int adac_type_is_supported(int t) { return t >= ADAC_UINT8 && t <= ADAC_INT64; }

extern "C" uint32_t adac_type_size(int t) {
	switch (t) {
	case ADAC_UINT8:
	case ADAC_INT8: return 1;
	case ADAC_UINT16:
	case ADAC_INT16: return 2;
	case ADAC_UINT32:
	case ADAC_INT32: return 4;
	case ADAC_UINT64:
	case ADAC_INT64: return 8;
	default: return 0;
	}
}

static bool type_is_signed(int t) { return t == ADAC_INT8 || t == ADAC_INT16 || t == ADAC_INT32 || t == ADAC_INT64; }

extern "C" uint32_t adac_hi(uint64_t x) { return x == 0 ? 0u : 63u - (uint32_t)__builtin_clzll(x); }

extern "C" uint8_t adac_width(uint64_t mn, uint64_t mx, int rule, int pad_to_byte) {
	uint32_t w;
	if (rule == ADAC_RULE_APPEND) {
		w = adac_hi(mx - mn) + 1;
	} else {
		if (mx != 0 && mn != UINT64_MAX && mx > mn) mx -= mn;
		w = adac_hi(mx) + 1;
	}
	if (pad_to_byte) w = (w + 7u) & ~7u;
	return (uint8_t)w;
}

extern "C" uint64_t adac_stored_min(uint64_t min, uint64_t max, uint8_t width) {
	// see k_plan: the all-ones segment keeps the reference's packed bits but gets a min that decodes them
	if (min == UINT64_MAX && max == UINT64_MAX && width >= 1 && width < 64) return UINT64_MAX - ((1ull << width) - 1ull);
	return min;
}

extern "C" uint64_t adac_packed_words(uint64_t count, uint8_t width) { return (count * width + 63) >> 6; }

extern "C" uint64_t adac_size_in_bytes(uint64_t count, uint8_t width) { return 9 + (adac_packed_words(count, width) << 3); }

extern "C" uint64_t adac_arena_words(uint64_t count, uint8_t width) {
	return (((count * width + 64) >> 6) + 15) & ~15ull;
}

extern "C" uint64_t adac_block_bytes(uint64_t count, uint8_t width) {
	return adac_size_in_bytes(count, width) + ADAC_BLOCK_TRAILER_BYTES;
}

extern "C" uint64_t adac_block_write(const adac_segment_desc *d, int physical_type, const uint64_t *words, void *out,
                                     uint64_t cap) {
	if (!d || !out || d->width == 0 || d->width > 64 || !adac_type_is_supported(physical_type)) return 0;
	const uint64_t nwords = adac_packed_words(d->count, d->width);
	if (nwords && !words) return 0;
	const uint64_t total = adac_block_bytes(d->count, d->width);
	if (cap < total) return 0;
	uint8_t *p = static_cast<uint8_t *>(out);
	const uint64_t bit_size = (uint64_t)d->count * d->width;
	std::memcpy(p, &bit_size, 8); // int_vector<0>::write_header: m_size, then m_width
	p[8] = d->width;
	if (nwords) std::memcpy(p + 9, words, nwords * 8); // write_data: capacity()/64 words
	uint8_t *t = p + 9 + nwords * 8;
	std::memset(t, 0, ADAC_BLOCK_TRAILER_BYTES);
	std::memcpy(t, &d->min, 8);
	t[8] = d->flags;
	t[9] = (uint8_t)physical_type;
	return total;
}

extern "C" adac_status adac_block_read(const void *block, uint64_t len, adac_segment_desc *d, int *physical_type,
                                       uint64_t *words_out, uint64_t cap_words) {
	if (!block || !d || len < 9 + ADAC_BLOCK_TRAILER_BYTES) return ADAC_ERR_INVALID_ARGUMENT;
	const uint8_t *p = static_cast<const uint8_t *>(block);
	uint64_t bit_size;
	std::memcpy(&bit_size, p, 8);
	const uint8_t width = p[8];
	if (width == 0 || width > 64 || bit_size % width != 0) return ADAC_ERR_INVALID_ARGUMENT;
	const uint64_t count = bit_size / width;
	if (count > 0xffffffffull) return ADAC_ERR_INVALID_ARGUMENT;
	const uint64_t nwords = (bit_size + 63) >> 6;
	if (len != 9 + nwords * 8 + ADAC_BLOCK_TRAILER_BYTES) return ADAC_ERR_INVALID_ARGUMENT;
	const uint8_t *t = p + 9 + nwords * 8;
	const int type = t[9];
	if (!adac_type_is_supported(type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	const uint32_t full_w = 8 * adac_type_size(type);
	const uint8_t flags = t[8];
	if (width > full_w || (!(flags & ADAC_SEG_PACKED) && width != full_w)) return ADAC_ERR_INVALID_ARGUMENT;
	if (nwords > cap_words || (nwords && !words_out)) return ADAC_ERR_INVALID_ARGUMENT;
	if (nwords) std::memcpy(words_out, p + 9, nwords * 8);
	std::memset(d, 0, sizeof(*d));
	std::memcpy(&d->min, t, 8);
	d->count = (uint32_t)count;
	d->width = width;
	d->flags = flags;
	if (physical_type) *physical_type = type;
	return ADAC_OK;
}

extern "C" uint64_t adac_block_stride(uint64_t count, uint8_t width) {
	return 8 * (adac_packed_words(count, width) + 4); // the image zero-padded to whole 8-byte units
}

extern "C" adac_status adac_block_peek(const void *block, uint64_t len, adac_segment_desc *d, int *physical_type) {
	// header + trailer only: what a loader needs to size the arena before the words move (adac_blocks_read)
	if (!block || !d || len < 9 + ADAC_BLOCK_TRAILER_BYTES) return ADAC_ERR_INVALID_ARGUMENT;
	const uint8_t *p = static_cast<const uint8_t *>(block);
	uint64_t bit_size;
	std::memcpy(&bit_size, p, 8);
	const uint8_t width = p[8];
	if (width == 0 || width > 64 || bit_size % width != 0) return ADAC_ERR_INVALID_ARGUMENT;
	const uint64_t count = bit_size / width;
	if (count > 0xffffffffull) return ADAC_ERR_INVALID_ARGUMENT;
	const uint64_t nwords = (bit_size + 63) >> 6;
	// len may be the exact image or the image padded to 8 bytes (adac_block_stride)
	if (len != 9 + nwords * 8 + ADAC_BLOCK_TRAILER_BYTES && len != 8 * (nwords + 4)) return ADAC_ERR_INVALID_ARGUMENT;
	const uint8_t *t = p + 9 + nwords * 8;
	const int type = t[9];
	if (!adac_type_is_supported(type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	const uint32_t full_w = 8 * adac_type_size(type);
	const uint8_t flags = t[8];
	if (width > full_w || (!(flags & ADAC_SEG_PACKED) && width != full_w)) return ADAC_ERR_INVALID_ARGUMENT;
	std::memset(d, 0, sizeof(*d));
	std::memcpy(&d->min, t, 8);
	d->count = (uint32_t)count;
	d->width = width;
	d->flags = flags;
	if (physical_type) *physical_type = type;
	return ADAC_OK;
}

static adac_status blocks_jobs(int physical_type, const adac_segment_desc *descs, const uint64_t *block_offs,
                               uint64_t nseg, std::vector<adac::BlockJob> &jobs, uint32_t &max_units) {
	if (!adac_type_is_supported(physical_type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	if (nseg && (!descs || !block_offs)) return ADAC_ERR_INVALID_ARGUMENT;
	const uint32_t full_w = 8 * adac_type_size(physical_type);
	jobs.resize(nseg);
	max_units = 0;
	for (uint64_t i = 0; i < nseg; i++) {
		const adac_segment_desc &d = descs[i];
		if (d.width == 0 || d.width > full_w || (d.word_off & 15) || (block_offs[i] & 7)) return ADAC_ERR_INVALID_ARGUMENT;
		if (!(d.flags & ADAC_SEG_PACKED) && d.width != full_w) return ADAC_ERR_INVALID_ARGUMENT;
		adac::BlockJob &j = jobs[i];
		std::memset(&j, 0, sizeof j);
		j.word_off = d.word_off;
		j.block_off = block_offs[i];
		j.min = d.min;
		j.bit_size = (uint64_t)d.count * d.width;
		j.nwords = (uint32_t)adac_packed_words(d.count, d.width);
		j.arena_words = (uint32_t)adac_arena_words(d.count, d.width);
		j.width = d.width;
		j.flags = d.flags;
		j.type = (uint8_t)physical_type;
		const uint32_t units = j.arena_words > j.nwords + 4 ? j.arena_words : j.nwords + 4;
		if (units > max_units) max_units = units;
	}
	return ADAC_OK;
}

extern "C" adac_status adac_blocks_write(adac_ctx *ctx, int physical_type, const adac_segment_desc *descs,
                                         const uint64_t *block_offs, uint64_t nseg, const uint64_t *d_words,
                                         void *d_blocks) {
	if (!ctx || (nseg && (!d_words || !d_blocks)) || ((uintptr_t)d_blocks & 7)) return ADAC_ERR_INVALID_ARGUMENT;
	std::vector<adac::BlockJob> jobs;
	uint32_t max_units = 0;
	adac_status st = blocks_jobs(physical_type, descs, block_offs, nseg, jobs, max_units);
	if (st != ADAC_OK || nseg == 0) return st;
	ADAC_HIP(hipSetDevice(ctx->device));
	adac::BlockJob *d_jobs = nullptr;
	ADAC_HIP(hipMalloc((void **)&d_jobs, nseg * sizeof(adac::BlockJob)));
	hipError_t e = hipMemcpyAsync(d_jobs, jobs.data(), nseg * sizeof(adac::BlockJob), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = adac::launch_blocks_write(ctx->stream, d_jobs, nseg, max_units, d_words, d_blocks);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); // the job table is a local: done before it goes
	(void)hipFree(d_jobs);
	ADAC_HIP(e);
	return ADAC_OK;
}

extern "C" adac_status adac_blocks_read(adac_ctx *ctx, int physical_type, const adac_segment_desc *descs,
                                        const uint64_t *block_offs, uint64_t nseg, const void *d_blocks,
                                        uint64_t *d_words) {
	if (!ctx || (nseg && (!d_words || !d_blocks)) || ((uintptr_t)d_blocks & 7) || ((uintptr_t)d_words & 15))
		return ADAC_ERR_INVALID_ARGUMENT;
	std::vector<adac::BlockJob> jobs;
	uint32_t max_units = 0;
	adac_status st = blocks_jobs(physical_type, descs, block_offs, nseg, jobs, max_units);
	if (st != ADAC_OK || nseg == 0) return st;
	ADAC_HIP(hipSetDevice(ctx->device));
	void *d_scratch = nullptr;
	const size_t jobs_bytes = nseg * sizeof(adac::BlockJob);
	ADAC_HIP(hipMalloc(&d_scratch, jobs_bytes + 16));
	adac::BlockJob *d_jobs = static_cast<adac::BlockJob *>(d_scratch);
	uint32_t *d_bad = reinterpret_cast<uint32_t *>(static_cast<uint8_t *>(d_scratch) + jobs_bytes);
	uint32_t bad = 0;
	hipError_t e = hipMemcpyAsync(d_jobs, jobs.data(), jobs_bytes, hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, 16, ctx->stream);
	if (e == hipSuccess) e = adac::launch_blocks_read(ctx->stream, d_jobs, nseg, max_units, d_blocks, d_words, d_bad);
	if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	(void)hipFree(d_scratch);
	ADAC_HIP(e);
	return bad ? ADAC_ERR_INVALID_ARGUMENT : ADAC_OK; // an image whose header is not the one the descriptor was made from
}

extern "C" uint32_t adac_tile_values(int t) {
	uint32_t ts = adac_type_size(t);
	return ts ? adac::tile_values(ts) : 0;
}

// diagnostic: the phase time stamps the single-pass encode recorded (8 x u64 per workgroup ticket) when the tuning knob
// "encode_stamps" is set; not part of the drop-in boundary
extern "C" int adac_debug_encode_stamps(void *host, uint64_t bytes) {
	if (!host) return 1;
	return adac::read_encode_stamps(host, bytes) == hipSuccess ? 0 : 1;
}

extern "C" int adac_set_tuning(const char *name, int value) {
	if (!name) return 1;
	const std::string n(name);
	if (n == "persistent_unpack") adac::g_tuning.persistent_unpack = value;
	else if (n == "scan_probe") adac::g_tuning.scan_probe = value;
	else if (n == "sel_debug") adac::g_tuning.sel_debug = value;
	else if (n == "grouped_repack") adac::g_tuning.grouped_repack = value;
	else if (n == "single_pass_encode") adac::g_tuning.single_pass_encode = value;
	else if (n == "encode_stamps") adac::g_tuning.encode_stamps = value;
	else if (n == "encode_placement") adac::g_tuning.encode_placement = value;
	else if (n == "encode_big_image") adac::g_tuning.encode_big_image = value;
	else if (n == "encode_publish_ahead") adac::g_tuning.encode_publish_ahead = value;
	else if (n == "scan_cells") adac::g_tuning.scan_cells = value;
	else if (n == "tile_records") adac::g_tuning.tile_records = value;
	else if (n == "gather_compact") adac::g_tuning.gather_compact = value;
	else if (n == "group_sum_wide") adac::g_tuning.group_sum_wide = value;
	else if (n == "group_sum_rw") adac::g_tuning.group_sum_rw = value;
	else if (n == "templated_scan") adac::g_tuning.templated_scan = value;
	else if (n == "scan_tiles_per_wg" && value >= 0) adac::g_tuning.scan_tiles_per_wg = value; // 0 = by type
	else if (n == "blocks_per_cu" && value > 0) adac::g_tuning.blocks_per_cu = value;
	else if (n == "num_cus" && value >= 0) adac::g_tuning.num_cus = value;
	else return 1;
	return 0;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------

extern "C" int adac_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

extern "C" adac_status adac_ctx_create(int device, void *external_stream, adac_ctx **out) {
	if (!out) return ADAC_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	int ndev = 0;
	hipError_t e = hipGetDeviceCount(&ndev);
	if (e != hipSuccess || ndev == 0) {
		g_last_error = "hipGetDeviceCount: no HIP device";
		return ADAC_ERR_NO_DEVICE;
	}
	if (device < 0 || device >= ndev) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(device));
	adac_ctx *c = new (std::nothrow) adac_ctx();
	if (!c) return ADAC_ERR_OUT_OF_MEMORY;
	c->device = device;
	if (external_stream) {
		c->stream = static_cast<hipStream_t>(external_stream);
	} else {
		e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
		if (e != hipSuccess) {
			delete c;
			return fail_hip(e, "hipStreamCreateWithFlags");
		}
		c->owns_stream = true;
	}
	(void)hipEventCreate(&c->ev_start);
	(void)hipEventCreate(&c->ev_stop);
	*out = c;
	return ADAC_OK;
}

extern "C" void adac_ctx_destroy(adac_ctx *c) {
	if (c) ctx_release(c);
}

extern "C" adac_status adac_ctx_sync(adac_ctx *c) {
	if (!c) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipStreamSynchronize(c->stream));
	return ADAC_OK;
}

extern "C" void *adac_ctx_stream(adac_ctx *c) { return c ? c->stream : nullptr; }
extern "C" int adac_ctx_device(adac_ctx *c) { return c ? c->device : -1; }

extern "C" adac_status adac_dev_alloc(adac_ctx *c, size_t bytes, void **d_ptr) {
	if (!c || !d_ptr) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipMalloc(d_ptr, bytes ? bytes : 16));
	return ADAC_OK;
}

extern "C" adac_status adac_dev_free(adac_ctx *c, void *d_ptr) {
	if (!c) return ADAC_ERR_INVALID_ARGUMENT;
	if (!d_ptr) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipFree(d_ptr));
	return ADAC_OK;
}

extern "C" adac_status adac_dev_memset(adac_ctx *c, void *d_ptr, int byte, size_t bytes) {
	if (!c || (!d_ptr && bytes)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!bytes) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipMemsetAsync(d_ptr, byte, bytes, c->stream));
	return ADAC_OK;
}

extern "C" adac_status adac_memcpy_h2d(adac_ctx *c, void *d_dst, const void *src, size_t bytes) {
	if (!c || ((!d_dst || !src) && bytes)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!bytes) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c->stream));
	ADAC_HIP(hipStreamSynchronize(c->stream));
	return ADAC_OK;
}

extern "C" adac_status adac_memcpy_d2h(adac_ctx *c, void *dst, const void *d_src, size_t bytes) {
	if (!c || ((!dst || !d_src) && bytes)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!bytes) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
	ADAC_HIP(hipStreamSynchronize(c->stream));
	return ADAC_OK;
}

extern "C" adac_status adac_memcpy_d2h_async(adac_ctx *c, void *dst, const void *d_src, size_t bytes) {
	if (!c || ((!dst || !d_src) && bytes)) return ADAC_ERR_INVALID_ARGUMENT;
	if (bytes == 0) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c->stream));
	return ADAC_OK;
}

extern "C" adac_status adac_host_alloc_pinned(adac_ctx *c, size_t bytes, void **ptr) {
	if (!c || !ptr) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocDefault));
	return ADAC_OK;
}

extern "C" adac_status adac_host_free_pinned(adac_ctx *c, void *ptr) {
	if (!c) return ADAC_ERR_INVALID_ARGUMENT;
	if (!ptr) return ADAC_OK;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipHostFree(ptr));
	return ADAC_OK;
}

extern "C" adac_status adac_timer_start(adac_ctx *c) {
	if (!c) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipEventRecord(c->ev_start, c->stream));
	return ADAC_OK;
}

extern "C" adac_status adac_timer_stop(adac_ctx *c, float *ms) {
	if (!c || !ms) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(c->device));
	ADAC_HIP(hipEventRecord(c->ev_stop, c->stream));
	ADAC_HIP(hipEventSynchronize(c->ev_stop));
	ADAC_HIP(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
	return ADAC_OK;
}

// ------------------------------------------------------------------------------------------------
// layout
// ------------------------------------------------------------------------------------------------

extern "C" adac_status adac_layout_create(adac_ctx *c, int type, const uint32_t *counts, const uint64_t *val_offs,
                                          uint64_t nseg, adac_layout **out) {
	if (!c || !out || (nseg && !counts)) return ADAC_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	if (!adac_type_is_supported(type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	if (nseg >= 0xffffffffull) return ADAC_ERR_INVALID_ARGUMENT;
	adac_layout *l = new (std::nothrow) adac_layout();
	if (!l) return ADAC_ERR_OUT_OF_MEMORY;
	l->ctx = c;
	ctx_retain(c);
	l->type = type;
	l->type_size = adac_type_size(type);
	l->is_signed = type_is_signed(type);
	l->null_bits = l->is_signed ? (1ull << (8 * l->type_size - 1)) : 0ull;
	l->nseg = nseg;
	l->counts.assign(counts, counts + nseg);
	l->val_offs.resize(nseg);
	const uint32_t tile = adac::tile_values(l->type_size);
	const uint8_t full_w = (uint8_t)(8 * l->type_size);
	std::vector<adac_segment_desc> descs(nseg);
	std::vector<TileRef> tiles;
	uint64_t run = 0, arena = 0;
	for (uint64_t s = 0; s < nseg; s++) {
		const uint64_t off = val_offs ? val_offs[s] : run;
		l->val_offs[s] = off;
		if (off != run) l->dense_values = false;
		if (((off & (16 / l->type_size - 1)) + (uint64_t)counts[s]) * l->type_size > adac::kEncodeOnePassBytes) {
			l->single_pass_ok = false;
		}
		run = off + counts[s];
		if (counts[s] == 0) l->has_empty_segments = true;
		if (off + counts[s] > l->value_span) l->value_span = off + counts[s];
		l->total_values += counts[s];
		adac_segment_desc &d = descs[s];
		d.word_off = arena;
		d.val_off = off;
		d.min = ADAC_NO_MIN;
		d.count = counts[s];
		d.width = full_w;
		d.flags = 0;
		d.reserved = 0;
		arena += adac_arena_words(counts[s], full_w);
		for (uint64_t first = 0; first < counts[s]; first += tile) {
			tiles.push_back(TileRef {(uint32_t)s, (uint32_t)first});
		}
	}
	l->max_arena_words = arena;
	l->ntiles = tiles.size();
	if (l->ntiles >= 0x7fffffffull) { // grid.x limit
		adac_layout_destroy(l);
		return ADAC_ERR_INVALID_ARGUMENT;
	}
	hipError_t e = hipSetDevice(c->device);
	if (e == hipSuccess) e = hipMalloc((void **)&l->d_descs, (nseg ? nseg : 1) * sizeof(adac_segment_desc));
	if (e == hipSuccess) e = hipMalloc((void **)&l->d_tiles, (l->ntiles ? l->ntiles : 1) * sizeof(TileRef));
	if (e == hipSuccess) e = hipMalloc((void **)&l->d_minmax, (nseg ? nseg : 1) * 2 * sizeof(uint64_t));
	if (e == hipSuccess && nseg)
		e = hipMemcpyAsync(l->d_descs, descs.data(), nseg * sizeof(adac_segment_desc), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess && l->ntiles)
		e = hipMemcpyAsync(l->d_tiles, tiles.data(), l->ntiles * sizeof(TileRef), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = adac::launch_minmax_init(c->stream, l->d_minmax, nseg);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream); // descs/tiles are stack-local
	if (e != hipSuccess) {
		adac_status st = fail_hip(e, "adac_layout_create");
		adac_layout_destroy(l);
		return st;
	}
	*out = l;
	return ADAC_OK;
}

extern "C" void adac_layout_destroy(adac_layout *l) {
	if (!l) return;
	(void)hipSetDevice(l->ctx->device);
	if (l->d_descs) (void)hipFree(l->d_descs);
	if (l->d_tiles) (void)hipFree(l->d_tiles);
	if (l->d_minmax) (void)hipFree(l->d_minmax);
	if (l->d_tile_cnt) (void)hipFree(l->d_tile_cnt);
	if (l->d_tile_off) (void)hipFree(l->d_tile_off);
	if (l->d_block_tot) (void)hipFree(l->d_block_tot);
	if (l->d_group_refs) (void)hipFree(l->d_group_refs);
	if (l->d_groups) (void)hipFree(l->d_groups);
	if (l->d_narrow_idx) (void)hipFree(l->d_narrow_idx);
	if (l->d_narrow_count) (void)hipFree(l->d_narrow_count);
	if (l->h_narrow_count) (void)hipHostFree(l->h_narrow_count);
	if (l->narrow_ev) (void)hipEventDestroy(l->narrow_ev);
	if (l->d_scan_state) (void)hipFree(l->d_scan_state);
	if (l->d_group_partial) (void)hipFree(l->d_group_partial);
	if (l->d_sel_edges) (void)hipFree(l->d_sel_edges);
	if (l->d_tile_recs) (void)hipFree(l->d_tile_recs);
	if (l->d_res_cells) (void)hipFree(l->d_res_cells);
	if (l->d_edge_cells) (void)hipFree(l->d_edge_cells);
	adac_ctx *c = l->ctx;
	delete l;
	ctx_release(c);
}

extern "C" uint64_t adac_layout_nseg(const adac_layout *l) { return l ? l->nseg : 0; }
extern "C" uint64_t adac_layout_ntiles(const adac_layout *l) { return l ? l->ntiles : 0; }
extern "C" uint64_t adac_layout_total_values(const adac_layout *l) { return l ? l->total_values : 0; }
extern "C" uint64_t adac_layout_value_span(const adac_layout *l) { return l ? l->value_span : 0; }
extern "C" uint64_t adac_layout_max_arena_words(const adac_layout *l) { return l ? l->max_arena_words : 0; }
extern "C" const adac_segment_desc *adac_layout_device_descs(const adac_layout *l) { return l ? l->d_descs : nullptr; }

extern "C" adac_status adac_layout_set_descs(adac_layout *l, const adac_segment_desc *descs) {
	if (!l || (l->nseg && !descs)) return ADAC_ERR_INVALID_ARGUMENT;
	const uint32_t full_w = 8 * l->type_size;
	for (uint64_t s = 0; s < l->nseg; s++) {
		const adac_segment_desc &d = descs[s];
		if (d.count != l->counts[s] || d.val_off != l->val_offs[s]) return ADAC_ERR_INVALID_ARGUMENT;
		if (d.width == 0 || d.width > full_w || (d.word_off & 15)) return ADAC_ERR_INVALID_ARGUMENT;
		if (!(d.flags & ADAC_SEG_PACKED) && d.width != full_w) return ADAC_ERR_INVALID_ARGUMENT;
	}
	ADAC_HIP(hipSetDevice(l->ctx->device));
	if (l->nseg) {
		ADAC_HIP(hipMemcpyAsync(l->d_descs, descs, l->nseg * sizeof(adac_segment_desc), hipMemcpyHostToDevice,
		                        l->ctx->stream));
		ADAC_HIP(hipStreamSynchronize(l->ctx->stream));
	}
	note_widths(l, descs);
	return descs_changed(l);
}

static void note_widths(adac_layout *l, const adac_segment_desc *descs) {
	uint32_t mx = 0;
	for (uint64_t s = 0; s < l->nseg; s++) {
		if (descs[s].count && descs[s].width > mx) mx = descs[s].width;
	}
	l->hint_max_width = mx;
}

extern "C" adac_status adac_layout_get_descs(adac_layout *l, adac_segment_desc *descs) {
	if (!l || (l->nseg && !descs)) return ADAC_ERR_INVALID_ARGUMENT;
	adac_status st = adac_memcpy_d2h(l->ctx, descs, l->d_descs, l->nseg * sizeof(adac_segment_desc));
	if (st == ADAC_OK) note_widths(l, descs);
	return st;
}

extern "C" adac_status adac_layout_get_minmax(adac_layout *l, uint64_t *minmax) {
	if (!l || (l->nseg && !minmax)) return ADAC_ERR_INVALID_ARGUMENT;
	return adac_memcpy_d2h(l->ctx, minmax, l->d_minmax, l->nseg * 2 * sizeof(uint64_t));
}

// ------------------------------------------------------------------------------------------------
// hot path
// ------------------------------------------------------------------------------------------------

static bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Descriptors changed (plan / set_descs): re-expand the scans' work-item records in stream order right away when
// the table exists, so that a captured graph replayed after a re-encode reads current records; otherwise the
// next scan builds the table.
static adac_status expand_groups(adac_layout *l) {
	ADAC_HIP(adac::launch_expand_groups(l->ctx->stream, l->d_descs, l->d_group_refs, l->ngroups, l->d_groups,
	                                    l->d_narrow_idx, l->d_narrow_count));
	ADAC_HIP(hipMemcpyAsync(l->h_narrow_count, l->d_narrow_count, sizeof(uint32_t), hipMemcpyDeviceToHost,
	                        l->ctx->stream));
	ADAC_HIP(hipEventRecord(l->narrow_ev, l->ctx->stream));
	l->narrow_pending = true;
	l->groups_dirty = false;
	return ADAC_OK;
}

// the decode-side tile records, rebuilt (in stream order) when descriptors changed since they were last built
static adac_status tile_records(adac_layout *l, const adac::TileRec **out) {
	*out = nullptr;
	if (!adac::g_tuning.tile_records || l->ntiles == 0) return ADAC_OK;
	if (!l->d_tile_recs) ADAC_HIP(hipMalloc((void **)&l->d_tile_recs, l->ntiles * sizeof(adac::TileRec)));
	if (l->tile_recs_dirty) {
		ADAC_HIP(adac::launch_expand_tiles(l->ctx->stream, l->type_size, l->d_descs, l->d_tiles, l->ntiles, l->d_tile_recs));
		l->tile_recs_dirty = false;
	}
	*out = l->d_tile_recs;
	return ADAC_OK;
}

static adac_status descs_changed(adac_layout *l) {
	l->groups_dirty = true;
	l->tile_recs_dirty = true;
	if (l->d_groups && l->groups_tiles > 0) return expand_groups(l);
	return ADAC_OK;
}

// the scans' view of the work items; waits for the narrow-group count of the last expansion if it is still on its way
static adac_status scan_group_list(adac_layout *l, adac::ScanGroupList *gl) {
	adac_status gst = ensure_scan_groups(l);
	if (gst != ADAC_OK) return gst;
	if (l->narrow_pending) {
		ADAC_HIP(hipEventSynchronize(l->narrow_ev));
		l->narrow_pending = false;
	}
	const bool cells = adac::g_tuning.scan_cells != 0;
	*gl = adac::ScanGroupList {l->d_groups, l->ngroups, l->d_narrow_idx, *l->h_narrow_count,
	                           cells ? l->d_res_cells : nullptr, cells ? l->d_edge_cells : nullptr};
	return ADAC_OK;
}

extern "C" adac_status adac_analyze(adac_layout *l, const void *d_vals, const uint64_t *d_validity, int rule) {
	if (!l || (!d_vals && l->total_values) || (rule != ADAC_RULE_APPEND && rule != ADAC_RULE_RECOMPACT))
		return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_vals)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_minmax_init(l->ctx->stream, l->d_minmax, l->nseg));
	ADAC_HIP(adac::launch_analyze(l->ctx->stream, l->type_size, l->is_signed, l->null_bits, rule, l->d_descs,
	                              l->d_tiles, l->ntiles, d_vals, d_validity, l->d_minmax));
	return ADAC_OK;
}

extern "C" adac_status adac_zonemap(adac_layout *l, const void *d_vals, const uint64_t *d_validity, uint64_t *zonemap) {
	if (!l || (!d_vals && l->total_values) || (l->nseg && !zonemap)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_vals)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	const uint32_t bits = 8 * l->type_size;
	const uint64_t umask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
	const uint64_t sbit = l->is_signed ? (1ull << (bits - 1)) : 0ull;
	ADAC_HIP(adac::launch_minmax_init(l->ctx->stream, l->d_minmax, l->nseg));
	// the kernel's `null_bits` argument carries the sign bit: min/max are taken over bits(v) ^ signbit
	ADAC_HIP(adac::launch_analyze(l->ctx->stream, l->type_size, false, sbit, ADAC_RULE_ZONEMAP, l->d_descs, l->d_tiles,
	                              l->ntiles, d_vals, d_validity, l->d_minmax));
	adac_status st = adac_layout_get_minmax(l, zonemap);
	if (st != ADAC_OK) return st;
	for (uint64_t s = 0; s < l->nseg; s++) {
		uint64_t mn = zonemap[2 * s], mx = zonemap[2 * s + 1];
		if (mn == UINT64_MAX && mx == 0) { // no valid row: empty interval
			mn = umask;
			mx = 0;
		}
		zonemap[2 * s] = (mn ^ sbit) & umask;
		zonemap[2 * s + 1] = (mx ^ sbit) & umask;
	}
	return ADAC_OK;
}

extern "C" adac_status adac_plan(adac_layout *l, int rule, int pad_to_byte) {
	if (!l || (rule != ADAC_RULE_APPEND && rule != ADAC_RULE_RECOMPACT)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_plan(l->ctx->stream, l->type_size, rule, pad_to_byte ? 1 : 0, l->d_descs, l->d_minmax,
	                           l->nseg));
	return descs_changed(l);
}

extern "C" adac_status adac_pack(adac_layout *l, const void *d_vals, const uint64_t *d_validity, uint64_t *d_words) {
	if (!l || ((!d_vals || !d_words) && l->total_values)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_vals) || !aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_pack(l->ctx->stream, l->type_size, l->null_bits, l->d_descs, l->d_tiles, l->ntiles, d_vals,
	                           d_validity, d_words));
	return ADAC_OK;
}

extern "C" adac_status adac_encode(adac_layout *l, const void *d_vals, const uint64_t *d_validity, int rule,
                                   int pad_to_byte, uint64_t *d_words) {
	// Which form?  Measured on 400 MB-of-packed-bytes columns (profiles/r03_encode_forms.json):
	//   8-byte types       single pass wins at every width (0.276 vs 0.363 ms at C2; 8 - 25 %);
	//   4-byte types       single pass wins only where its output is parked (w = 8, 16); at the image-flow widths the
	//                      ORDERED placement makes every round of segments wait for its slowest workgroup and the three
	//                      kernels are 25 - 47 % faster (w = 13: 0.453 vs 0.589 ms) — with first-come placement the single
	//                      pass is the fastest form everywhere (0.378 ms), so it is taken whenever that is selected;
	//   2- / 1-byte types  three kernels (the single-pass instantiations are instruction-bound / spill:
	//                      profiles/r03_encode_small_types.json).
	// Knob "single_pass_encode": 0 never, 1 by this table, 2 always.
	// 4-byte types under ordered placement: the single pass wins (8 - 14 %) while a full segment's packed words fit the LDS
	// pool — widths up to 17 (the big image publishes ahead) — and loses (21 - 36 %) above; the widths are not known before
	// the analysis, so the choice follows the widest segment the host last SAW of this layout (a re-encode of a column, or
	// the adaptive policy's repeated compaction), and stays with the three kernels when it saw none
	// (profiles/r03_encode_big_image.json)
	const bool narrow_hint = l && l->hint_max_width != 0 && l->hint_max_width <= 17;
	const bool one_pass_pays = l && (l->type_size == 8 || (l->type_size == 4 && (adac::g_tuning.encode_placement == 1 || narrow_hint)));
	if (l && adac::g_tuning.single_pass_encode && l->single_pass_ok &&
	    (one_pass_pays || adac::g_tuning.single_pass_encode > 1)) {
		// one kernel, the raw column read once (adac_encode_1p.inl); same descriptors, min/max and words
		if ((!d_vals || !d_words) && l->total_values) return ADAC_ERR_INVALID_ARGUMENT;
		if (rule != ADAC_RULE_APPEND && rule != ADAC_RULE_RECOMPACT) return ADAC_ERR_INVALID_ARGUMENT;
		if (!aligned16(d_vals) || !aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
		ADAC_HIP(hipSetDevice(l->ctx->device));
		if (!l->d_scan_state) ADAC_HIP(hipMalloc(&l->d_scan_state, adac::encode_1p_state_words(l->nseg) * sizeof(uint64_t)));
		ADAC_HIP(adac::launch_encode_1p(l->ctx->stream, l->type_size, l->is_signed, l->null_bits, rule,
		                                pad_to_byte ? 1 : 0, l->d_descs, l->nseg, d_vals, d_validity, l->d_minmax,
		                                l->d_scan_state, d_words));
		return descs_changed(l);
	}
	adac_status st = adac_analyze(l, d_vals, d_validity, rule);
	if (st != ADAC_OK) return st;
	st = adac_plan(l, rule, pad_to_byte);
	if (st != ADAC_OK) return st;
	return adac_pack(l, d_vals, d_validity, d_words);
}

// packed -> packed re-compaction: both layouts describe the same segments (type, counts, value offsets)
static bool same_shape(const adac_layout *a, const adac_layout *b) {
	return a && b && a != b && a->ctx == b->ctx && a->type == b->type && a->counts == b->counts &&
	       a->val_offs == b->val_offs;
}

extern "C" adac_status adac_analyze_packed(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity,
                                           int rule, adac_layout *dst) {
	if (!same_shape(src, dst) || (rule != ADAC_RULE_APPEND && rule != ADAC_RULE_RECOMPACT))
		return ADAC_ERR_INVALID_ARGUMENT;
	if ((!d_src_words && src->total_values) || !aligned16(d_src_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(src->ctx->device));
	ADAC_HIP(adac::launch_minmax_init(src->ctx->stream, dst->d_minmax, dst->nseg));
	if (adac::g_tuning.grouped_repack) {
		adac_status gst = ensure_scan_groups(src);
		if (gst != ADAC_OK) return gst;
		ADAC_HIP(adac::launch_analyze_packed_g(src->ctx->stream, src->type_size, src->is_signed, src->null_bits, rule,
		                                       src->d_groups, src->ngroups, d_src_words, d_validity, dst->d_minmax));
		return ADAC_OK;
	}
	ADAC_HIP(adac::launch_analyze_packed(src->ctx->stream, src->type_size, src->is_signed, src->null_bits, rule,
	                                     src->d_descs, src->d_tiles, src->ntiles, d_src_words, d_validity,
	                                     dst->d_minmax));
	return ADAC_OK;
}

extern "C" adac_status adac_repack(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity,
                                   adac_layout *dst, uint64_t *d_dst_words) {
	if (!same_shape(src, dst)) return ADAC_ERR_INVALID_ARGUMENT;
	if ((!d_src_words || !d_dst_words) && src->total_values) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_src_words) || !aligned16(d_dst_words) || d_src_words == d_dst_words)
		return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(src->ctx->device));
	if (adac::g_tuning.grouped_repack) {
		adac_status gst = ensure_scan_groups(src);
		if (gst != ADAC_OK) return gst;
		ADAC_HIP(adac::launch_repack_g(src->ctx->stream, src->type_size, src->null_bits, src->d_groups, src->ngroups,
		                               dst->d_descs, d_src_words, d_validity, d_dst_words));
		return ADAC_OK;
	}
	ADAC_HIP(adac::launch_repack(src->ctx->stream, src->type_size, src->null_bits, src->d_descs, dst->d_descs,
	                             src->d_tiles, src->ntiles, d_src_words, d_validity, d_dst_words));
	return ADAC_OK;
}

extern "C" adac_status adac_reencode(adac_layout *src, const uint64_t *d_src_words, const uint64_t *d_validity,
                                     int rule, int pad_to_byte, adac_layout *dst, uint64_t *d_dst_words) {
	adac_status st = adac_analyze_packed(src, d_src_words, d_validity, rule, dst);
	if (st != ADAC_OK) return st;
	st = adac_plan(dst, rule, pad_to_byte);
	if (st != ADAC_OK) return st;
	return adac_repack(src, d_src_words, d_validity, dst, d_dst_words);
}

extern "C" adac_status adac_unpack(adac_layout *l, const uint64_t *d_words, void *d_out) {
	if (!l || ((!d_words || !d_out) && l->total_values)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_words) || !aligned16(d_out)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	// (the decode itself keeps the two-hop start — tile entry -> descriptor: with the expanded records it ran 4 - 17 %
	// SLOWER, same box, interleaved (profiles/r03_tile_records.json); knob tile_records = 3 is that A/B form)
	const adac::TileRec *recs = nullptr;
	if (adac::g_tuning.tile_records & 2) {
		adac_status st = tile_records(l, &recs);
		if (st != ADAC_OK) return st;
	}
	ADAC_HIP(adac::launch_unpack(l->ctx->stream, l->type_size, l->d_descs, l->d_tiles, l->ntiles, d_words, d_out, recs));
	return ADAC_OK;
}

extern "C" adac_status adac_unpack_range(adac_layout *l, const uint64_t *d_words, uint64_t seg, uint64_t start,
                                         uint64_t count, void *d_out, uint64_t out_off) {
	if (!l || seg >= l->nseg) return ADAC_ERR_INVALID_ARGUMENT;
	if (start > l->counts[seg] || count > l->counts[seg] - start) return ADAC_ERR_INVALID_ARGUMENT;
	if (count == 0) return ADAC_OK;
	if (!d_words || !d_out || !aligned16(d_words) || !aligned16(d_out)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	RangeArgs r {(uint32_t)seg, (uint32_t)start, (uint32_t)count, out_off};
	ADAC_HIP(adac::launch_unpack_range(l->ctx->stream, l->type_size, l->d_descs, r, d_words, d_out));
	return ADAC_OK;
}

extern "C" adac_status adac_unpack_jobs(adac_ctx *ctx, int physical_type, const adac_unpack_job *jobs, uint64_t njobs,
                                        const uint64_t *d_words, void *d_out) {
	if (!ctx) return ADAC_ERR_INVALID_ARGUMENT;
	if (!adac_type_is_supported(physical_type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	if (njobs == 0) return ADAC_OK;
	if (!jobs || !d_words || !d_out || !aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	const uint32_t ts = adac_type_size(physical_type);
	if ((uintptr_t)d_out % ts) return ADAC_ERR_INVALID_ARGUMENT;
	const uint32_t tile = adac::tile_values(ts);
	for (uint64_t i = 0; i < njobs; i++) {
		const adac_unpack_job &j = jobs[i];
		if (j.width == 0 || j.width > 8 * ts || (j.word_off & 15)) return ADAC_ERR_INVALID_ARGUMENT;
		if ((uint64_t)j.start + j.count > 0xffffffffull) return ADAC_ERR_INVALID_ARGUMENT;
	}
	ADAC_HIP(hipSetDevice(ctx->device));
	// the store path wants chunk alignment relative to a 16-byte aligned base: fold d_out's own misalignment into
	// the element offsets
	const uint64_t bias = ((uintptr_t)d_out & 15) / ts;
	void *base = static_cast<uint8_t *>(d_out) - bias * ts;
	adac::UnpackJobTable table;
	std::memset(&table, 0, sizeof table);
	for (uint64_t i = 0; i < njobs; i++) {
		const adac_unpack_job &j = jobs[i];
		if (j.count) {
			adac::UnpackJob &u = table.jobs[table.njobs++];
			u.word_off = j.word_off;
			u.add = ((j.flags & ADAC_SEG_PACKED) && j.min != ADAC_NO_MIN) ? j.min : 0ull;
			u.out_off = j.out_off + bias;
			u.start = j.start;
			u.count = j.count;
			u.width = j.width;
			u.tile0 = table.ntiles;
			table.ntiles += (j.count + tile - 1) / tile;
		}
		if (table.njobs == (uint32_t)adac::kMaxUnpackJobs || (i + 1 == njobs && table.njobs)) {
			ADAC_HIP(adac::launch_unpack_jobs(ctx->stream, ts, table, d_words, base));
			table.njobs = table.ntiles = 0;
		}
	}
	return ADAC_OK;
}

struct adac_event {
	adac_ctx *ctx = nullptr;
	hipEvent_t ev = nullptr;
};

extern "C" adac_status adac_event_record(adac_ctx *ctx, adac_event **out) {
	if (!ctx || !out) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(ctx->device));
	hipEvent_t ev = nullptr;
	ADAC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	hipError_t e = hipEventRecord(ev, ctx->stream);
	if (e != hipSuccess) {
		(void)hipEventDestroy(ev);
		return fail_hip(e, "hipEventRecord");
	}
	adac_event *h = new (std::nothrow) adac_event();
	if (!h) {
		(void)hipEventDestroy(ev);
		return ADAC_ERR_OUT_OF_MEMORY;
	}
	ctx_retain(ctx);
	h->ctx = ctx;
	h->ev = ev;
	*out = h;
	return ADAC_OK;
}

extern "C" adac_status adac_event_wait(adac_event *ev) {
	if (!ev) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipEventSynchronize(ev->ev));
	return ADAC_OK;
}

extern "C" int adac_event_done(adac_event *ev) { return ev && hipEventQuery(ev->ev) == hipSuccess; }

extern "C" void adac_event_destroy(adac_event *ev) {
	if (!ev) return;
	(void)hipEventDestroy(ev->ev);
	ctx_release(ev->ctx);
	delete ev;
}

extern "C" adac_status adac_fetch_rows(adac_layout *l, const uint64_t *d_words, const uint32_t *d_segs,
                                       const uint32_t *d_rows, uint64_t n, void *d_out) {
	if (!l) return ADAC_ERR_INVALID_ARGUMENT;
	if (n == 0) return ADAC_OK;
	if (!d_words || !d_segs || !d_rows || !d_out) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_fetch(l->ctx->stream, l->type_size, l->d_descs, d_words, d_segs, d_rows, n, d_out));
	return ADAC_OK;
}

// The scans' work items for the current grouping knob; the expansion kernel runs on the codec stream, so it is
// ordered after the plan / descriptor upload that made it necessary.
static adac_status ensure_scan_groups(adac_layout *l) {
	// A segment's tiles are dealt EVENLY to ceil(tiles / target) workgroups.  Default target: 12 tiles of u64, 6 of
	// u32 (~24 K rows), 8 of u16, 4 of u8 (64 K rows): a 16-tile DuckDB segment becomes two to four groups.  Measured (tools/ab_tuning.py,
	// 400 M rows): a whole 32 767-row segment per workgroup puts the workgroups' packed streams and bitmap regions
	// at the same power-of-two stride (32 KiB / 4 KiB) and costs 6-25 %; 16-24 K rows is the best of 4 .. 32 tiles
	const uint32_t tile = adac::tile_values(l->type_size);
	int per = adac::g_tuning.scan_tiles_per_wg;
	if (per < 1) { // by type: best of 1 .. 32 tiles in tools/ab_tuning.py (u8 and u16 want more rows per workgroup)
		per = l->type_size == 8 ? 12 : l->type_size == 4 ? 6 : l->type_size == 2 ? 8 : 4;
	}
	if ((uint64_t)per * tile > 65536) per = (int)(65536 / tile); // the selection scan's LDS image holds 64 Ki rows
	if (l->groups_tiles != per) {
		std::vector<adac::ScanGroupRef> refs;
		for (uint64_t s = 0; s < l->nseg; s++) {
			const uint64_t ntiles = ((uint64_t)l->counts[s] + tile - 1) / tile;
			if (ntiles == 0) continue;
			const uint64_t ngroups = (ntiles + (uint64_t)per - 1) / (uint64_t)per;
			const uint64_t rows_per_group = ((ntiles + ngroups - 1) / ngroups) * tile;
			const size_t seg_first = refs.size();
			for (uint64_t first = 0; first < l->counts[s]; first += rows_per_group) {
				const uint64_t left = l->counts[s] - first;
				adac::ScanGroupRef r {};
				r.seg = (uint32_t)s;
				r.first = (uint32_t)first;
				r.rows = (uint32_t)(left < rows_per_group ? left : rows_per_group);
				refs.push_back(r);
			}
			for (size_t g = seg_first; g < refs.size(); g++) refs[g].seg_groups = (uint32_t)(refs.size() - seg_first);
		}
		if (refs.size() >= 0x7fffffffull) return ADAC_ERR_INVALID_ARGUMENT;
		if (l->dense_values) {
			// Bitmap words that groups cover in part (sel_write_out's test: word 0 of a group unless the group starts on a
			// word boundary and fills it; its last word, if it has more than one, unless it ends on a boundary).  The
			// groups are in element order, so the arrivals at one word are neighbours in this list: a run of equal words
			// shares the cell named after its first arrival and expects as many arrivals as the run is long.
			struct Arrival {
				uint64_t word;
				uint32_t group, which;
			};
			std::vector<Arrival> arr;
			for (size_t g = 0; g < refs.size(); g++) {
				const uint64_t p0 = l->val_offs[refs[g].seg] + refs[g].first, p1 = p0 + refs[g].rows;
				const uint64_t nwords = ((p1 + 31) >> 5) - (p0 >> 5);
				if (!((p0 & 31) == 0 && p0 + 32 <= p1)) arr.push_back(Arrival {p0 >> 5, (uint32_t)g, 0u});
				if (nwords > 1 && (p1 & 31) != 0) arr.push_back(Arrival {(p1 - 1) >> 5, (uint32_t)g, 1u});
			}
			for (size_t a = 0; a < arr.size();) {
				size_t b = a;
				while (b < arr.size() && arr[b].word == arr[a].word) b++;
				const uint32_t cell = 2u * arr[a].group + arr[a].which;
				const uint16_t n = (uint16_t)(b - a); // a word holds 32 rows and every group at least one: n <= 32
				for (size_t i = a; i < b; i++) {
					adac::ScanGroupRef &r = refs[arr[i].group];
					if (arr[i].which == 0) {
						r.cell_first = cell;
						r.n_first = n;
					} else {
						r.cell_last = cell;
						r.n_last = n;
					}
				}
				a = b;
			}
		}
		if (refs.size() >= 0x7fffffffull) return ADAC_ERR_INVALID_ARGUMENT;
		if (l->d_group_refs) (void)hipFree(l->d_group_refs);
		if (l->d_groups) (void)hipFree(l->d_groups);
		if (l->d_narrow_idx) (void)hipFree(l->d_narrow_idx);
		if (l->d_edge_cells) (void)hipFree(l->d_edge_cells);
		l->d_edge_cells = nullptr;
		l->d_group_refs = nullptr;
		l->d_groups = nullptr;
		l->d_narrow_idx = nullptr;
		l->ngroups = refs.size();
		ADAC_HIP(hipMalloc((void **)&l->d_group_refs, (refs.size() ? refs.size() : 1) * sizeof(adac::ScanGroupRef)));
		ADAC_HIP(hipMalloc((void **)&l->d_groups, (refs.size() ? refs.size() : 1) * sizeof(adac::ScanGroup)));
		ADAC_HIP(hipMalloc((void **)&l->d_narrow_idx, (refs.size() ? refs.size() : 1) * sizeof(uint32_t)));
		ADAC_HIP(hipMalloc((void **)&l->d_edge_cells, adac::scan_edge_cell_bytes(refs.size())));
		ADAC_HIP(hipMemsetAsync(l->d_edge_cells, 0, adac::scan_edge_cell_bytes(refs.size()), l->ctx->stream));
		if (!l->d_res_cells) {
			ADAC_HIP(hipMalloc((void **)&l->d_res_cells, adac::scan_res_cell_bytes(l->nseg)));
			ADAC_HIP(hipMemsetAsync(l->d_res_cells, 0, adac::scan_res_cell_bytes(l->nseg), l->ctx->stream));
		}
		if (!l->d_narrow_count) {
			ADAC_HIP(hipMalloc((void **)&l->d_narrow_count, sizeof(uint32_t)));
			ADAC_HIP(hipHostMalloc((void **)&l->h_narrow_count, sizeof(uint32_t), hipHostMallocDefault));
			ADAC_HIP(hipEventCreateWithFlags(&l->narrow_ev, hipEventDisableTiming));
			*l->h_narrow_count = 0;
		}
		if (!refs.empty()) {
			ADAC_HIP(hipMemcpyAsync(l->d_group_refs, refs.data(), refs.size() * sizeof(adac::ScanGroupRef),
			                        hipMemcpyHostToDevice, l->ctx->stream));
			ADAC_HIP(hipStreamSynchronize(l->ctx->stream)); // refs is stack-local
		}
		l->groups_tiles = per;
		l->groups_dirty = true;
	}
	if (l->groups_dirty) return expand_groups(l);
	return ADAC_OK;
}

extern "C" adac_status adac_scan_sum_valid(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity,
                                           uint64_t *d_sums);
extern "C" adac_status adac_scan_sum(adac_layout *l, const uint64_t *d_words, uint64_t *d_sums) {
	return adac_scan_sum_valid(l, d_words, nullptr, d_sums);
}

extern "C" adac_status adac_scan_sum_valid(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity,
                                           uint64_t *d_sums) {
	if (!l || (l->nseg && !d_sums) || (l->total_values && !d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	const uint64_t sbit = l->is_signed ? (1ull << (8 * l->type_size - 1)) : 0ull;
	adac::ScanGroupList gl;
	adac_status gst = scan_group_list(l, &gl);
	if (gst != ADAC_OK) return gst;
	// (with arrival cells every segment's total is STORED by the last of its groups: nothing to clear; a segment
	// without rows has no group and is cleared here)
	if (l->nseg && (!gl.d_res_cells || l->has_empty_segments)) {
		ADAC_HIP(hipMemsetAsync(d_sums, 0, l->nseg * sizeof(uint64_t), l->ctx->stream));
	}
	ADAC_HIP(adac::launch_scan_sum(l->ctx->stream, l->type_size, gl, d_words, d_validity, sbit, d_sums));
	return ADAC_OK;
}

// SUM(value), COUNT(*) GROUP BY key over two packed columns of one table (Q1's shape, TPCH_runtime.txt:2-6)
extern "C" adac_status adac_scan_group_sum(adac_layout *values, const uint64_t *d_value_words, adac_layout *keys,
                                           const uint64_t *d_key_words, uint32_t ngroups, uint64_t *d_sums,
                                           uint64_t *d_counts) {
	if (!values || !keys || values->ctx != keys->ctx || !d_sums || !d_counts) return ADAC_ERR_INVALID_ARGUMENT;
	if (ngroups == 0 || ngroups > adac::group_sum_max_groups()) return ADAC_ERR_INVALID_ARGUMENT;
	if (values->counts != keys->counts) return ADAC_ERR_INVALID_ARGUMENT; // the same rows, segment by segment
	if (values->total_values && (!d_value_words || !d_key_words)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_value_words) || !aligned16(d_key_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(values->ctx->device));
	if (!values->d_group_partial) {
		ADAC_HIP(hipMalloc(&values->d_group_partial, adac::group_sum_partial_bytes()));
		ADAC_HIP(hipMemsetAsync(values->d_group_partial, 0, adac::group_sum_partial_bytes(), values->ctx->stream));
	}
	adac_status gst = ensure_scan_groups(values); // the register-walk kernel's work items
	if (gst != ADAC_OK) return gst;
	ADAC_HIP(adac::launch_group_sum(values->ctx->stream, values->type_size, values->is_signed, keys->type_size,
	                                values->d_descs, values->d_tiles, values->ntiles, values->d_groups, values->ngroups,
	                                d_value_words, keys->d_descs, d_key_words, ngroups, values->d_group_partial,
	                                values->group_calls++, d_sums, d_counts));
	return ADAC_OK;
}

extern "C" adac_status adac_scan_count_between_valid(adac_layout *l, const uint64_t *d_words,
                                                     const uint64_t *d_validity, uint64_t lo, uint64_t hi,
                                                     uint64_t *d_counts);
extern "C" adac_status adac_scan_count_between(adac_layout *l, const uint64_t *d_words, uint64_t lo, uint64_t hi,
                                               uint64_t *d_counts) {
	return adac_scan_count_between_valid(l, d_words, nullptr, lo, hi, d_counts);
}

static adac_status scan_range(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity, uint64_t lo,
                              uint64_t hi, uint64_t *d_counts, uint64_t *d_bitmap, bool want_bitmap) {
	if (!l || (l->nseg && !d_counts) || (l->total_values && !d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	if (want_bitmap && ((l->value_span && !d_bitmap) || d_bitmap == d_validity)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	// order-preserving map of T onto unsigned numbers: flip the sign bit of the signed types
	const uint32_t bits = 8 * l->type_size;
	const uint64_t umask = bits >= 64 ? ~0ull : ((1ull << bits) - 1ull);
	const uint64_t sbit = l->is_signed ? (1ull << (bits - 1)) : 0ull;
	const uint64_t blo = (lo & umask) ^ sbit, bhi = (hi & umask) ^ sbit;
	// The scan writes every bitmap word that lies inside a group whole (zero words included).  With the segments back
	// to back in the value space the words two groups share are all that is left: every group leaves its bits of
	// them in a record, and a tiny kernel ORs the records of a word and stores it afterwards — no clearing pass
	// (12.5 MB at C2) and no global atomics.  Value spaces with gaps between segments, and the empty range, take the
	// full memset and atomicOr.
	// (A/B: sel_debug 5 takes the memset + atomicOr form on a dense value space too)
	// (the diagnostic forms that skip the write-out — sel_debug 1, 2 — leave no records: the merge kernel must not run
	// over them, it would store through uninitialised word indices)
	const bool edges_only = want_bitmap && l->dense_values && bhi >= blo && l->ntiles &&
	                        (adac::g_tuning.sel_debug == 0 || adac::g_tuning.sel_debug >= 6);
	if (want_bitmap && l->value_span && !edges_only) {
		ADAC_HIP(hipMemsetAsync(d_bitmap, 0, ((l->value_span + 63) / 64) * sizeof(uint64_t), l->ctx->stream));
	}
	if (bhi < blo) { // empty range: all counts (and bits) are zero
		if (l->nseg) ADAC_HIP(hipMemsetAsync(d_counts, 0, l->nseg * sizeof(uint64_t), l->ctx->stream));
		return ADAC_OK;
	}
	adac::ScanGroupList gl;
	adac_status gst = scan_group_list(l, &gl);
	if (gst != ADAC_OK) return gst;
	// (with arrival cells every segment's count is STORED by the last of its groups: nothing to clear)
	if (l->nseg && (!gl.d_res_cells || l->has_empty_segments)) {
		ADAC_HIP(hipMemsetAsync(d_counts, 0, l->nseg * sizeof(uint64_t), l->ctx->stream));
	}
	const uint64_t words32 = (l->value_span + 31) / 32; // the odd half of the last 64-bit word, if there is one
	const uint64_t tail_word = (words32 & 1) ? words32 : ~0ull;
	const bool edge_cells = edges_only && gl.d_edge_cells != nullptr; // shared words finished inside the scan kernel
	if (edges_only && !edge_cells && l->sel_edges_groups < l->ngroups) {
		if (l->d_sel_edges) ADAC_HIP(hipFree(l->d_sel_edges));
		l->d_sel_edges = nullptr;
		l->sel_edges_groups = 0;
		ADAC_HIP(hipMalloc(&l->d_sel_edges, adac::sel_edge_bytes(l->ngroups)));
		l->sel_edges_groups = l->ngroups;
	}
	ADAC_HIP(adac::launch_scan_count_range(l->ctx->stream, l->type_size, gl, d_words, d_validity, blo, bhi - blo, sbit,
	                                       d_counts, want_bitmap ? d_bitmap : nullptr,
	                                       edges_only && !edge_cells ? l->d_sel_edges : nullptr, edge_cells, tail_word));
	if (edges_only && !edge_cells) {
		ADAC_HIP(adac::launch_sel_merge_edges(l->ctx->stream, l->d_sel_edges, l->ngroups, d_bitmap, tail_word));
	}
	return ADAC_OK;
}

extern "C" adac_status adac_scan_count_between_valid(adac_layout *l, const uint64_t *d_words,
                                                     const uint64_t *d_validity, uint64_t lo, uint64_t hi,
                                                     uint64_t *d_counts) {
	return scan_range(l, d_words, d_validity, lo, hi, d_counts, nullptr, false);
}

extern "C" adac_status adac_scan_select_between(adac_layout *l, const uint64_t *d_words, const uint64_t *d_validity,
                                                uint64_t lo, uint64_t hi, uint64_t *d_bitmap, uint64_t *d_counts) {
	return scan_range(l, d_words, d_validity, lo, hi, d_counts, d_bitmap, true);
}

extern "C" adac_status adac_unpack_selected(adac_layout *l, const uint64_t *d_words, const uint64_t *d_bitmap,
                                            void *d_out, uint64_t *d_out_ids, uint64_t *total_out) {
	if (!l || (l->total_values && (!d_words || !d_bitmap || !d_out))) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_words)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	if (!l->d_tile_cnt) {
		const uint64_t nt = l->ntiles ? l->ntiles : 1;
		ADAC_HIP(hipMalloc((void **)&l->d_tile_cnt, nt * sizeof(uint32_t)));
		ADAC_HIP(hipMalloc((void **)&l->d_tile_off, nt * sizeof(uint64_t)));
		ADAC_HIP(hipMalloc((void **)&l->d_block_tot, ((nt + 1023) / 1024 + 1) * sizeof(uint64_t)));
	}
	const uint64_t nblocks = (l->ntiles + 1023) / 1024;
	uint64_t *d_total = l->d_block_tot + nblocks; // the spare slot after the block totals
	const adac::TileRec *recs = nullptr;
	adac_status rst = tile_records(l, &recs);
	if (rst != ADAC_OK) return rst;
	ADAC_HIP(adac::launch_gather_selected(l->ctx->stream, l->type_size, l->d_descs, l->d_tiles, l->ntiles, d_words, recs,
	                                      d_bitmap, (l->value_span + 63) / 64, l->d_tile_cnt, l->d_tile_off, l->d_block_tot,
	                                      d_out, d_out_ids, d_total));
	if (total_out) return adac_memcpy_d2h(l->ctx, total_out, d_total, sizeof(uint64_t));
	return ADAC_OK;
}

extern "C" adac_status adac_scan_count_eq(adac_layout *l, const uint64_t *d_words, uint64_t key, uint64_t *d_counts) {
	return adac_scan_count_between(l, d_words, key, key, d_counts);
}

// ------------------------------------------------------------------------------------------------
// DuckDB BITPACKING segments: the persistent counterpart (decode side)
// ------------------------------------------------------------------------------------------------

struct adac_bp_layout {
	adac_ctx *ctx = nullptr;
	uint32_t type_size = 0;
	uint64_t nseg = 0, ngroups = 0, total_values = 0;
	std::vector<uint32_t> counts;
	std::vector<uint64_t> seg_first_group; // index of each segment's first metadata group in the group table
	void *d_groups = nullptr;
	uint64_t *d_block_offs = nullptr;
	const void *bound_blocks = nullptr; // blocks buffer whose group headers are parsed into d_groups
};

extern "C" adac_status adac_bp_layout_create(adac_ctx *c, int physical_type, const uint64_t *block_offs,
                                             const uint32_t *counts, const uint64_t *out_offs, uint64_t nseg,
                                             adac_bp_layout **out) {
	if (!c || !out || (nseg && (!block_offs || !counts))) return ADAC_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	if (!adac_type_is_supported(physical_type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	adac_bp_layout *l = new (std::nothrow) adac_bp_layout();
	if (!l) return ADAC_ERR_OUT_OF_MEMORY;
	l->ctx = c;
	ctx_retain(c);
	l->type_size = adac_type_size(physical_type);
	l->nseg = nseg;
	l->counts.assign(counts, counts + nseg);
	std::vector<adac::BpGroupHost> groups;
	uint64_t run = 0;
	for (uint64_t s = 0; s < nseg; s++) {
		if (block_offs[s] & 15) {
			adac_bp_layout_destroy(l);
			return ADAC_ERR_INVALID_ARGUMENT;
		}
		const uint64_t off = out_offs ? out_offs[s] : run;
		l->seg_first_group.push_back(groups.size());
		for (uint64_t r = 0; r < counts[s]; r += 2048) {
			const uint32_t rows = (uint32_t)(counts[s] - r < 2048 ? counts[s] - r : 2048);
			groups.push_back(adac::BpGroupHost {block_offs[s], off + r, (uint32_t)(r / 2048), rows, 0, 0, 0, 0, 0});
		}
		run = off + counts[s];
		l->total_values += counts[s];
	}
	l->ngroups = groups.size();
	if (l->ngroups >= 0x7fffffffull) {
		adac_bp_layout_destroy(l);
		return ADAC_ERR_INVALID_ARGUMENT;
	}
	hipError_t e = hipSetDevice(c->device);
	if (e == hipSuccess) e = hipMalloc(&l->d_groups, (l->ngroups ? l->ngroups : 1) * sizeof(adac::BpGroupHost));
	if (e == hipSuccess) e = hipMalloc((void **)&l->d_block_offs, (nseg ? nseg : 1) * sizeof(uint64_t));
	if (e == hipSuccess && l->ngroups)
		e = hipMemcpyAsync(l->d_groups, groups.data(), l->ngroups * sizeof(adac::BpGroupHost), hipMemcpyHostToDevice,
		                   c->stream);
	if (e == hipSuccess && nseg)
		e = hipMemcpyAsync(l->d_block_offs, block_offs, nseg * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (e != hipSuccess) {
		adac_status st = fail_hip(e, "adac_bp_layout_create");
		adac_bp_layout_destroy(l);
		return st;
	}
	*out = l;
	return ADAC_OK;
}

extern "C" void adac_bp_layout_destroy(adac_bp_layout *l) {
	if (!l) return;
	(void)hipSetDevice(l->ctx->device);
	if (l->d_groups) (void)hipFree(l->d_groups);
	if (l->d_block_offs) (void)hipFree(l->d_block_offs);
	adac_ctx *c = l->ctx;
	delete l;
	ctx_release(c);
}

extern "C" uint64_t adac_bp_layout_ngroups(const adac_bp_layout *l) { return l ? l->ngroups : 0; }
extern "C" uint64_t adac_bp_layout_total_values(const adac_bp_layout *l) { return l ? l->total_values : 0; }

extern "C" adac_status adac_bp_bind(adac_bp_layout *l, const void *d_blocks) {
	if (!l || (!d_blocks && l->total_values) || !aligned16(d_blocks)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_bp_prepare(l->ctx->stream, l->type_size, l->d_groups, l->ngroups, d_blocks));
	l->bound_blocks = d_blocks;
	return ADAC_OK;
}

extern "C" adac_status adac_bp_unpack(adac_bp_layout *l, const void *d_blocks, void *d_out) {
	if (!l || ((!d_blocks || !d_out) && l->total_values)) return ADAC_ERR_INVALID_ARGUMENT;
	if (!aligned16(d_blocks) || !aligned16(d_out)) return ADAC_ERR_INVALID_ARGUMENT;
	if (l->bound_blocks != d_blocks) {
		adac_status st = adac_bp_bind(l, d_blocks);
		if (st != ADAC_OK) return st;
	}
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_bp_unpack(l->ctx->stream, l->type_size, l->d_groups, l->ngroups, d_blocks, d_out));
	return ADAC_OK;
}

extern "C" adac_status adac_bp_unpack_range(adac_bp_layout *l, const void *d_blocks, uint64_t seg, uint64_t start,
                                            uint64_t count, void *d_out, uint64_t out_off) {
	if (!l || seg >= l->nseg) return ADAC_ERR_INVALID_ARGUMENT;
	if (start > l->counts[seg] || count > l->counts[seg] - start) return ADAC_ERR_INVALID_ARGUMENT;
	if (count == 0) return ADAC_OK;
	if (!d_blocks || !d_out || !aligned16(d_blocks) || !aligned16(d_out)) return ADAC_ERR_INVALID_ARGUMENT;
	if (l->bound_blocks != d_blocks) {
		adac_status st = adac_bp_bind(l, d_blocks);
		if (st != ADAC_OK) return st;
	}
	ADAC_HIP(hipSetDevice(l->ctx->device));
	const uint64_t g0 = l->seg_first_group[seg] + start / 2048;
	ADAC_HIP(adac::launch_bp_unpack_range(l->ctx->stream, l->type_size, l->d_groups, (uint32_t)g0,
	                                      (uint32_t)(start % 2048), count, out_off, d_blocks, d_out));
	return ADAC_OK;
}

extern "C" adac_status adac_bp_fetch_rows(adac_bp_layout *l, const void *d_blocks, const uint32_t *d_segs,
                                          const uint32_t *d_rows, uint64_t n, void *d_out) {
	if (!l) return ADAC_ERR_INVALID_ARGUMENT;
	if (n == 0) return ADAC_OK;
	if (!d_blocks || !d_segs || !d_rows || !d_out || !aligned16(d_blocks)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(l->ctx->device));
	ADAC_HIP(adac::launch_bp_fetch(l->ctx->stream, l->type_size, l->d_block_offs, d_blocks, d_segs, d_rows, n, d_out));
	return ADAC_OK;
}

// ---- BITPACKING compress: device statistics + host decisions (BitpackingState::Flush) + device group writes ----

namespace {

struct BpType {
	uint32_t ts, bits;
	bool sg;
	uint64_t mask, sbit;
};

inline int64_t bp_s64(const BpType &t, uint64_t x) {
	x &= t.mask;
	if (t.ts == 8) return (int64_t)x;
	const uint64_t sb = 1ull << (t.bits - 1);
	return (int64_t)((x ^ sb) - sb);
}

// TrySubtractOperator::Operation (src/function/scalar/operators/subtract.cpp:82-160) in T (as_signed = T's
// signedness) or in T_S (as_signed = true)
inline bool bp_try_sub(const BpType &t, bool as_signed, uint64_t l, uint64_t r, uint64_t *res) {
	if (!as_signed) {
		if ((r & t.mask) > (l & t.mask)) return false;
		*res = (l - r) & t.mask;
		return true;
	}
	if (t.ts == 8) {
		int64_t o;
		if (__builtin_sub_overflow((int64_t)l, (int64_t)r, &o)) return false;
		*res = (uint64_t)o;
		return true;
	}
	const int64_t d = bp_s64(t, l) - bp_s64(t, r);
	const int64_t lim = (int64_t)(t.mask >> 1);
	if (d < -lim - 1 || d > lim) return false;
	*res = (uint64_t)d & t.mask;
	return true;
}

inline uint32_t bp_eff_width(const BpType &t, uint32_t w) { // GetEffectiveWidth, bitpacking.hpp:208-216
	return (w + t.ts > t.bits) ? t.bits : w;
}
inline uint32_t bp_width_unsigned(const BpType &t, uint64_t v) { // FindMinimumBitWidth<T_U>
	v &= t.mask;
	if (v == 0) return 0;
	uint32_t w = 0;
	while (v) {
		w++;
		v >>= 1;
	}
	return bp_eff_width(t, w);
}
inline uint32_t bp_width_signed(const BpType &t, uint64_t v) { // FindMinimumBitWidth<T> for a signed T
	const uint64_t tmin = (~(t.mask >> 1)) & t.mask;
	if ((v & t.mask) == tmin) return t.bits;
	const int64_t s = bp_s64(t, v);
	uint64_t mag = (uint64_t)(s < 0 ? -s : s);
	if (mag == 0) return 0;
	uint32_t w = 1;
	while (mag) {
		w++;
		mag >>= 1;
	}
	return bp_eff_width(t, w);
}
inline uint64_t bp_required_size(uint64_t count, uint32_t w) { // GetRequiredSize, bitpacking.hpp:99-102
	return ((count + 31) / 32 * 32) * w / 8;
}

constexpr uint64_t kBpBlock = 262144 - 8; // Storage::BLOCK_SIZE

} // namespace

struct adac_bp_plan {
	adac_ctx *ctx = nullptr;
	uint32_t type_size = 0;
	bool is_signed = false;
	uint64_t n = 0, ngroups = 0;
	bool encodable = true;
	struct Seg {
		uint64_t start, count, total_size;
	};
	std::vector<Seg> segs;
	uint64_t by_mode[5] = {0, 0, 0, 0, 0};
	void *d_recs = nullptr;
};

extern "C" void adac_bp_plan_destroy(adac_bp_plan *p) {
	if (!p) return;
	(void)hipSetDevice(p->ctx->device);
	if (p->d_recs) (void)hipFree(p->d_recs);
	adac_ctx *c = p->ctx;
	delete p;
	ctx_release(c);
}

extern "C" adac_status adac_bp_plan_create(adac_ctx *c, int physical_type, const void *d_vals, const uint64_t *d_validity,
                                           uint64_t n, int force_mode, adac_bp_plan **out) {
	if (!c || !out || (n && !d_vals) || force_mode < 0 || force_mode > 4) return ADAC_ERR_INVALID_ARGUMENT;
	*out = nullptr;
	if (!adac_type_is_supported(physical_type)) return ADAC_ERR_UNSUPPORTED_TYPE;
	adac_bp_plan *p = new (std::nothrow) adac_bp_plan();
	if (!p) return ADAC_ERR_OUT_OF_MEMORY;
	p->ctx = c;
	ctx_retain(c);
	p->type_size = adac_type_size(physical_type);
	p->is_signed = type_is_signed(physical_type);
	p->n = n;
	p->ngroups = (n + 2047) / 2048;
	BpType t;
	t.ts = p->type_size;
	t.bits = 8 * t.ts;
	t.sg = p->is_signed;
	t.mask = t.ts == 8 ? ~0ull : ((1ull << t.bits) - 1ull);
	t.sbit = t.sg ? (1ull << (t.bits - 1)) : 0ull;
	const uint64_t sbit_s = 1ull << (t.bits - 1);
	const uint64_t ts_max = t.mask >> 1;                  // NumericLimits<T_S>::Maximum()
	const uint64_t t_max = t.sg ? ts_max : t.mask;
	const uint64_t t_min = t.sg ? ((~ts_max) & t.mask) : 0ull;

	std::vector<adac::BpStatsHost> stats(p->ngroups);
	void *d_stats = nullptr;
	hipError_t e = hipSetDevice(c->device);
	if (e == hipSuccess && p->ngroups) e = hipMalloc(&d_stats, p->ngroups * sizeof(adac::BpStatsHost));
	if (e == hipSuccess) e = adac::launch_bp_stats(c->stream, p->type_size, p->is_signed, d_vals, d_validity, n, d_stats);
	if (e == hipSuccess && p->ngroups)
		e = hipMemcpyAsync(stats.data(), d_stats, p->ngroups * sizeof(adac::BpStatsHost), hipMemcpyDeviceToHost, c->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
	if (d_stats) (void)hipFree(d_stats);
	if (e != hipSuccess) {
		adac_status st = fail_hip(e, "adac_bp_plan_create");
		adac_bp_plan_destroy(p);
		return st;
	}

	// BitpackingState::Flush per group (bitpacking.cpp:229-294) + ReserveSpace / FlushSegment placement (:453-512)
	std::vector<adac::BpWriteHost> recs(p->ngroups);
	uint64_t data_ptr = 8, meta_ptr = kBpBlock, seg_first_group = 0, seg_start = 0, seg_count = 0;
	auto finish_segment = [&](uint64_t end_group) {
		const uint64_t metadata_offset = (data_ptr + 7) & ~7ull;
		const uint64_t k = end_group - seg_first_group;
		const uint64_t total = metadata_offset + 4 * k;
		for (uint64_t g = seg_first_group; g < end_group; g++) {
			recs[g].meta_off = (uint32_t)(total - 4 * (g - seg_first_group + 1));
			recs[g].first = total;
			recs[g].first_of_segment = g == seg_first_group;
		}
		p->segs.push_back(adac_bp_plan::Seg {seg_start, seg_count, total});
	};
	for (uint64_t g = 0; g < p->ngroups; g++) {
		const adac::BpStatsHost &st = stats[g];
		adac::BpWriteHost &r = recs[g];
		std::memset(&r, 0, sizeof(r));
		const uint64_t rows = st.rows;
		const bool all_invalid = st.nvalid == 0, all_valid = st.nvalid == rows;
		const uint64_t minimum = all_invalid ? t_max : ((st.bmin ^ t.sbit) & t.mask);
		const uint64_t maximum = all_invalid ? t_min : ((st.bmax ^ t.sbit) & t.mask);
		uint64_t bytes = 0;
		bool done = false;
		if ((all_invalid || maximum == minimum) && (force_mode == 0 || force_mode == 1)) {
			r.mode = 1;
			r.frame = maximum;
			bytes = t.ts;
			done = true;
		}
		uint64_t mmd = 0, mmdd = 0, delta_offset = 0, min_delta = 0, max_delta = 0;
		bool can_do_for = false, can_do_delta = false;
		if (!done) {
			can_do_for = bp_try_sub(t, t.sg, maximum, minimum, &mmd);
			const bool above = !t.sg && maximum > ts_max; // maximum > (T)NumericLimits<T_S>::Maximum()
			if (!above && rows >= 2 && all_valid) {
				bool can_do_all = true;
				if (t.sg) {
					uint64_t bogus;
					can_do_all = bp_try_sub(t, true, minimum, maximum, &bogus) && bp_try_sub(t, true, maximum, minimum, &bogus);
				}
				if (can_do_all || !st.delta_overflow) {
					min_delta = (st.bdmin ^ sbit_s) & t.mask;
					max_delta = (st.bdmax ^ sbit_s) & t.mask;
					can_do_delta = bp_try_sub(t, true, max_delta, min_delta, &mmdd);
					can_do_delta = can_do_delta && bp_try_sub(t, true, st.v0, min_delta, &delta_offset);
				}
			}
			if (can_do_delta) {
				if (max_delta == min_delta && force_mode != 4 && force_mode != 3) {
					r.mode = 2;
					r.frame = st.v0;
					r.extra = max_delta;
					bytes = 2 * t.ts;
					done = true;
				} else {
					const uint32_t dw = bp_width_unsigned(t, mmdd);
					const uint32_t rw = t.sg ? bp_width_signed(t, mmd) : bp_width_unsigned(t, mmd);
					if (dw < rw && force_mode != 4) {
						r.mode = 3;
						r.frame = min_delta;
						r.width = dw;
						r.extra = delta_offset;
						bytes = bp_required_size(rows, dw) + 3 * t.ts;
						done = true;
					}
				}
			}
			if (!done && can_do_for) {
				r.mode = 4;
				r.frame = minimum;
				r.width = bp_width_unsigned(t, mmd);
				bytes = bp_required_size(rows, r.width) + 2 * t.ts;
				done = true;
			}
		}
		if (!done) {
			p->encodable = false; // Flush() returned false: BitpackingFinalAnalyze reports INVALID_INDEX
			break;
		}
		if (meta_ptr - data_ptr < bytes + 4) { // FlushAndCreateSegmentIfFull
			finish_segment(g);
			seg_first_group = g;
			seg_start += seg_count;
			seg_count = 0;
			data_ptr = 8;
			meta_ptr = kBpBlock;
		}
		r.seg = (uint32_t)p->segs.size();
		r.data_off = (uint32_t)data_ptr;
		r.rows = (uint32_t)rows;
		data_ptr += bytes;
		meta_ptr -= 4;
		seg_count += rows;
		p->by_mode[r.mode]++;
	}
	if (p->encodable) {
		finish_segment(p->ngroups);
		if (p->ngroups) {
			e = hipMalloc(&p->d_recs, p->ngroups * sizeof(adac::BpWriteHost));
			if (e == hipSuccess)
				e = hipMemcpyAsync(p->d_recs, recs.data(), p->ngroups * sizeof(adac::BpWriteHost), hipMemcpyHostToDevice,
				                   c->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
			if (e != hipSuccess) {
				adac_status st2 = fail_hip(e, "adac_bp_plan_create(upload)");
				adac_bp_plan_destroy(p);
				return st2;
			}
		}
	} else {
		p->segs.clear();
	}
	*out = p;
	return ADAC_OK;
}

extern "C" int adac_bp_plan_encodable(const adac_bp_plan *p) { return p && p->encodable ? 1 : 0; }
extern "C" uint64_t adac_bp_plan_nseg(const adac_bp_plan *p) { return p ? p->segs.size() : 0; }
extern "C" uint64_t adac_bp_plan_groups_by_mode(const adac_bp_plan *p, int mode) {
	return (p && mode >= 1 && mode <= 4) ? p->by_mode[mode] : 0;
}
extern "C" adac_status adac_bp_plan_segment(const adac_bp_plan *p, uint64_t i, uint64_t *start, uint64_t *count,
                                            uint64_t *total_size) {
	if (!p || i >= p->segs.size()) return ADAC_ERR_INVALID_ARGUMENT;
	if (start) *start = p->segs[i].start;
	if (count) *count = p->segs[i].count;
	if (total_size) *total_size = p->segs[i].total_size;
	return ADAC_OK;
}

extern "C" adac_status adac_bp_write(adac_bp_plan *p, const void *d_vals, const uint64_t *d_validity, void *d_blocks,
                                     uint64_t block_stride) {
	if (!p || !p->encodable || (p->n && (!d_vals || !d_blocks))) return ADAC_ERR_INVALID_ARGUMENT;
	if (block_stride < kBpBlock || (block_stride & 15) || !aligned16(d_blocks)) return ADAC_ERR_INVALID_ARGUMENT;
	ADAC_HIP(hipSetDevice(p->ctx->device));
	// alignment gaps and unused tails are zero (the reference leaves them as the buffer manager handed them out)
	if (!p->segs.empty()) ADAC_HIP(hipMemsetAsync(d_blocks, 0, p->segs.size() * block_stride, p->ctx->stream));
	ADAC_HIP(adac::launch_bp_write(p->ctx->stream, p->type_size, p->d_recs, p->ngroups, d_vals, d_validity, block_stride,
	                               d_blocks));
	return ADAC_OK;
}
