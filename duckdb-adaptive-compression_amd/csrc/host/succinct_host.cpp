// succinct_host.cpp — see succinct_host.hpp.  Device work goes through include/adacodec.h only; the host
// side moves bytes (memcpy, NULL-slot fill) and keeps the reference's state machine and accounting.
#include "succinct_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "adacodec_host.h"

namespace adacodec {

static void Check(adac_status st, const char *what) {
	if (st != ADAC_OK) {
		std::string msg = std::string("adacodec: ") + what + ": " + adac_status_string(st);
		if (st == ADAC_ERR_DEVICE) msg += std::string(" (") + adac_last_error() + ")";
		throw InternalException(msg);
	}
}

static bool IsSigned(PhysicalType t) {
	return t == PhysicalType::INT8 || t == PhysicalType::INT16 || t == PhysicalType::INT32 || t == PhysicalType::INT64;
}

// NullValue<T>() = numeric_limits<T>::min() as the bit pattern of T (null_value.hpp:26-28)
static uint64_t NullBits(PhysicalType t, idx_t type_size) {
	return IsSigned(t) ? (1ull << (8 * type_size - 1)) : 0ull;
}

// ------------------------------------------------------------------------------------------------
// SegmentPool
// ------------------------------------------------------------------------------------------------

SegmentPool::SegmentPool(int device, size_t arena_bytes) {
	Check(adac_ctx_create(device, nullptr, &ctx), "adac_ctx_create");
	arena_words = ((arena_bytes / 8) + 15) & ~15ull;
	if (arena_words < 16) arena_words = 16;
	void *p = nullptr;
	adac_status st = adac_dev_alloc(ctx, arena_words * 8, &p);
	if (st != ADAC_OK) {
		adac_ctx_destroy(ctx);
		Check(st, "adac_dev_alloc(arena)");
	}
	d_arena = static_cast<uint64_t *>(p);
	adac_dev_memset(ctx, d_arena, 0, arena_words * 8);
	adac_ctx_sync(ctx);
	free_list[0] = arena_words;
}

const uint8_t *SegmentPool::CacheLookup(const void *key) {
	auto it = cache.find(key);
	if (it == cache.end()) {
		cache_misses++;
		return nullptr;
	}
	cache_hits++;
	it->second.stamp = ++cache_clock;
	CacheSettle(it->second);
	return it->second.data;
}

void SegmentPool::CacheReserve() {
	if (cache_slab || cache_capacity < kCacheSlotBytes) return;
	const size_t nslots = cache_capacity / kCacheSlotBytes;
	void *p = nullptr;
	if (adac_host_alloc_pinned(ctx, nslots * kCacheSlotBytes, &p) != ADAC_OK) return; // entries fall back to own blocks
	cache_slab = static_cast<uint8_t *>(p);
	cache_free_slots.reserve(nslots);
	for (size_t i = nslots; i-- > 0;) cache_free_slots.push_back((int32_t)i);
}

void SegmentPool::CacheSettle(CacheEntry &e) {
	if (e.pending) {
		Check(adac_ctx_sync(ctx), "adac_ctx_sync(prefetch)"); // one stream: everything queued before it is done too
		for (auto &kv : cache) kv.second.pending = false;
	}
}

static void CacheRelease(SegmentPool &pool, SegmentPool::CacheEntry &e) {
	pool.CacheSettle(e); // the DMA may still be writing the block
	if (e.slot >= 0) {
		pool.cache_free_slots.push_back(e.slot);
	} else {
		adac_host_free_pinned(pool.ctx, e.data);
	}
	pool.cache_used -= e.slot >= 0 ? SegmentPool::kCacheSlotBytes : e.bytes;
}

uint8_t *SegmentPool::CacheInsert(const void *key, size_t bytes) {
	if (bytes > cache_capacity) return nullptr;
	CacheReserve();
	const bool slotted = cache_slab && bytes <= kCacheSlotBytes;
	const size_t charge = slotted ? kCacheSlotBytes : bytes;
	while ((cache_used + charge > cache_capacity || (slotted && cache_free_slots.empty())) && !cache.empty()) {
		auto victim = cache.begin(); // least recently used
		for (auto it = cache.begin(); it != cache.end(); ++it) {
			if (it->second.stamp < victim->second.stamp) victim = it;
		}
		CacheRelease(*this, victim->second);
		cache.erase(victim);
	}
	CacheEntry e;
	e.bytes = bytes;
	e.stamp = ++cache_clock;
	if (slotted && !cache_free_slots.empty()) {
		e.slot = cache_free_slots.back();
		cache_free_slots.pop_back();
		e.data = cache_slab + (size_t)e.slot * kCacheSlotBytes;
	} else {
		void *p = nullptr;
		if (adac_host_alloc_pinned(ctx, bytes, &p) != ADAC_OK) return nullptr;
		e.data = static_cast<uint8_t *>(p);
	}
	cache[key] = e;
	cache_used += e.slot >= 0 ? kCacheSlotBytes : bytes;
	return e.data;
}

void SegmentPool::CacheDrop(const void *key) {
	auto it = cache.find(key);
	if (it == cache.end()) return;
	CacheRelease(*this, it->second);
	cache.erase(it);
}

SegmentPool::~SegmentPool() {
	for (auto &e : cache) {
		if (e.second.slot < 0) adac_host_free_pinned(ctx, e.second.data);
	}
	if (cache_slab) adac_host_free_pinned(ctx, cache_slab);
	if (h_pinned) adac_host_free_pinned(ctx, h_pinned);
	if (d_prefetch) adac_dev_free(ctx, d_prefetch);
	if (d_staging) adac_dev_free(ctx, d_staging);
	if (d_staging2) adac_dev_free(ctx, d_staging2);
	if (d_arena) adac_dev_free(ctx, d_arena);
	adac_ctx_destroy(ctx);
}

uint64_t SegmentPool::Allocate(uint64_t words) {
	words = (words + 15) & ~15ull;
	for (auto it = free_list.begin(); it != free_list.end(); ++it) {
		if (it->second >= words) {
			uint64_t off = it->first, len = it->second;
			free_list.erase(it);
			if (len > words) free_list[off + words] = len - words;
			used_words += words;
			return off;
		}
	}
	throw InternalException("adacodec: segment pool arena exhausted");
}

void SegmentPool::Free(uint64_t off, uint64_t words) {
	words = (words + 15) & ~15ull;
	used_words -= words;
	auto next = free_list.lower_bound(off);
	if (next != free_list.begin()) {
		auto prev = std::prev(next);
		if (prev->first + prev->second == off) { // merge with the block before
			off = prev->first;
			words += prev->second;
			free_list.erase(prev);
		}
	}
	if (next != free_list.end() && off + words == next->first) { // and with the block after
		words += next->second;
		free_list.erase(next);
	}
	free_list[off] = words;
}

static void *Grow(adac_ctx *ctx, void *&buf, size_t &have, size_t want) {
	if (want > have) {
		if (buf) adac_dev_free(ctx, buf);
		buf = nullptr;
		size_t n = std::max(want, have * 2);
		n = (n + 255) & ~size_t(255);
		Check(adac_dev_alloc(ctx, n, &buf), "adac_dev_alloc(staging)");
		have = n;
	}
	return buf;
}

uint8_t *SegmentPool::PinnedStaging(size_t bytes) {
	if (bytes > pinned_bytes) {
		if (h_pinned) adac_host_free_pinned(ctx, h_pinned);
		h_pinned = nullptr;
		size_t n = std::max(bytes, pinned_bytes * 2);
		n = (n + 4095) & ~size_t(4095);
		Check(adac_host_alloc_pinned(ctx, n, &h_pinned), "adac_host_alloc_pinned(staging)");
		pinned_bytes = n;
	}
	return static_cast<uint8_t *>(h_pinned);
}

void *SegmentPool::Staging(size_t bytes) {
	return Grow(ctx, d_staging, staging_bytes, bytes + 64);
}
void *SegmentPool::PrefetchStaging(size_t bytes) {
	return Grow(ctx, d_prefetch, prefetch_bytes, bytes + 64);
}
void *SegmentPool::Staging2(size_t bytes) {
	return Grow(ctx, d_staging2, staging2_bytes, bytes + 64);
}

// ADACH_TRACE=1: wall time of the phases of a batched compaction on stderr (diagnostic only).
struct PhaseTrace {
	bool on;
	std::chrono::steady_clock::time_point t0;
	std::string line;
	PhaseTrace() : on(std::getenv("ADACH_TRACE") != nullptr), t0(std::chrono::steady_clock::now()) {
	}
	void mark(const char *what) {
		if (!on) return;
		auto t1 = std::chrono::steady_clock::now();
		char buf[96];
		std::snprintf(buf, sizeof buf, " %s=%.3fms", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
		line += buf;
		t0 = t1;
	}
	~PhaseTrace() {
		if (on && !line.empty()) std::fprintf(stderr, "[adach]%s\n", line.c_str());
	}
};

// A batch layout shared by the segments compacted together (one adac_layout, many segments).
struct LayoutHandle {
	adac_layout *layout = nullptr;
	~LayoutHandle() {
		if (layout) adac_layout_destroy(layout);
	}
};

static adac_layout *LayoutOf(const std::shared_ptr<void> &handle) {
	return static_cast<LayoutHandle *>(handle.get())->layout;
}

// ------------------------------------------------------------------------------------------------
// CompressionFunction tables
// ------------------------------------------------------------------------------------------------

static void CodecScanPartial(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result,
                             idx_t result_offset) {
	// succinct.cpp:123-144 / fixed_size_uncompressed.cpp FixedSizeScanPartial
	auto start = segment.GetRelativeIndex(state.row_index);
	result.flat = true; // SetVectorType(FLAT_VECTOR)
	segment.ScanRows(start, scan_count, result.data + result_offset * segment.type_size);
}
static void CodecScan(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result) {
	CodecScanPartial(segment, state, scan_count, result, 0); // succinct.cpp:232-240
}
static void CodecFetchRow(ColumnSegment &segment, ColumnFetchState &, row_t row_id, Vector &result, idx_t result_idx) {
	// intended semantics of SuccinctFetchRow (succinct.cpp:244-260): one value at row_id
	segment.ScanRows((idx_t)row_id, 1, result.data + result_idx * segment.type_size);
}
static void EmptySkip(ColumnSegment &, ColumnScanState &, idx_t) {
}
static idx_t CodecAppend(ColumnSegment &segment, UnifiedVectorFormat &data, idx_t offset, idx_t count) {
	return segment.AppendRows(data, offset, count); // succinct.cpp:308-322 / FixedSizeAppend
}
static idx_t CodecFinalizeAppend(ColumnSegment &segment) {
	return segment.count * segment.type_size; // succinct.cpp:324-330
}

// FixedSizeInitAnalyze / FixedSizeAnalyze / FixedSizeFinalAnalyze<T> (fixed_size_uncompressed.cpp:20-41): the
// succinct codec borrows them unchanged (succinct.cpp:337-338) — the score is the uncompressed size.
struct FixedSizeAnalyzeState : public AnalyzeState {
	idx_t count = 0;
	idx_t type_size = 0;
};
static std::unique_ptr<AnalyzeState> FixedSizeInitAnalyze(PhysicalType type) {
	auto state = std::unique_ptr<FixedSizeAnalyzeState>(new FixedSizeAnalyzeState());
	state->type_size = adac_type_size((int)type);
	return std::move(state);
}
static bool FixedSizeAnalyze(AnalyzeState &state_p, Vector &, idx_t count) {
	static_cast<FixedSizeAnalyzeState &>(state_p).count += count;
	return true;
}
static idx_t FixedSizeFinalAnalyze(AnalyzeState &state_p) {
	auto &state = static_cast<FixedSizeAnalyzeState &>(state_p);
	return state.type_size * state.count;
}

// SuccinctCompressState and its three slots (succinct.cpp:52-119; UncompressedFunctions for the other codec)
ColumnDataCheckpointer::ColumnDataCheckpointer(DatabaseInstance &db_p, PhysicalType type_p, idx_t row_group_start_p)
    : db(db_p), type(type_p), row_group_start(row_group_start_p) {
}
ColumnDataCheckpointer::~ColumnDataCheckpointer() = default;

struct SuccinctCompressState : public CompressionState {
	explicit SuccinctCompressState(ColumnDataCheckpointer &checkpointer_p) : checkpointer(checkpointer_p) {
		CreateEmptySegment(checkpointer.row_group_start);
	}
	void CreateEmptySegment(idx_t row_start) {
		current_segment = ColumnSegment::CreateTransientSegment(checkpointer.db, checkpointer.type, row_start);
	}
	void FlushSegment(idx_t segment_size) {
		checkpointer.flushed_sizes.push_back(segment_size);
		checkpointer.flushed_segments.push_back(std::move(current_segment));
	}
	void Finalize(idx_t segment_size) {
		FlushSegment(segment_size);
		current_segment.reset();
	}
	ColumnDataCheckpointer &checkpointer;
	std::unique_ptr<ColumnSegment> current_segment;
};
static std::unique_ptr<CompressionState> InitCompression(ColumnDataCheckpointer &checkpointer,
                                                         std::unique_ptr<AnalyzeState>) {
	return std::unique_ptr<CompressionState>(new SuccinctCompressState(checkpointer));
}
static void Compress(CompressionState &state_p, Vector &data, idx_t count) {
	auto &state = static_cast<SuccinctCompressState &>(state_p);
	UnifiedVectorFormat vdata; // data.ToUnifiedFormat(count, vdata)
	vdata.data = data.data;
	vdata.validity = data.validity;
	idx_t offset = 0;
	while (count > 0) {
		idx_t appended = state.current_segment->Append(vdata, offset, count);
		if (appended == count) return; // appended everything: finished
		auto next_start = state.current_segment->start + state.current_segment->count;
		state.FlushSegment(state.current_segment->FinalizeAppend()); // the segment is full
		state.CreateEmptySegment(next_start);
		offset += appended;
		count -= appended;
	}
}
static void FinalizeCompress(CompressionState &state_p) {
	auto &state = static_cast<SuccinctCompressState &>(state_p);
	state.Finalize(state.current_segment->FinalizeAppend());
}

// FixedSizeInitScan / SuccinctInitAppend pin the segment's block (fixed_size_uncompressed.cpp:125-130,
// succinct.cpp:264-269); here the bits are owned by the segment / the pool arena, so the states carry nothing.
static std::unique_ptr<SegmentScanState> CodecInitScan(ColumnSegment &) {
	return std::unique_ptr<SegmentScanState>(new SegmentScanState());
}
static std::unique_ptr<CompressionAppendState> CodecInitAppend(ColumnSegment &) {
	return std::unique_ptr<CompressionAppendState>(new CompressionAppendState());
}

bool SuccinctFun::TypeIsSupported(PhysicalType type) {
	return adac_type_is_supported((int)type) != 0;
}

static CompressionFunction MakeFunction(CompressionType type, PhysicalType data_type) {
	return CompressionFunction {type,           data_type,        FixedSizeInitAnalyze, FixedSizeAnalyze,
	                            FixedSizeFinalAnalyze, InitCompression, Compress,        FinalizeCompress,
	                            CodecInitScan,  CodecScan,        CodecScanPartial,     CodecFetchRow,
	                            EmptySkip,      nullptr,          CodecInitAppend,      CodecAppend,
	                            CodecFinalizeAppend, nullptr};
}

CompressionFunction SuccinctFun::GetFunction(PhysicalType data_type) {
	if (!TypeIsSupported(data_type)) throw InternalException("Unsupported type for FixedSizeSuccinct::GetFunction");
	return MakeFunction(CompressionType::COMPRESSION_SUCCINCT, data_type);
}

CompressionFunction UncompressedFun::GetFunction(PhysicalType data_type) {
	if (!SuccinctFun::TypeIsSupported(data_type)) throw InternalException("Unsupported type for FixedSizeUncompressed");
	return MakeFunction(CompressionType::COMPRESSION_UNCOMPRESSED, data_type);
}

// ------------------------------------------------------------------------------------------------
// DatabaseInstance
// ------------------------------------------------------------------------------------------------

DatabaseInstance::DatabaseInstance(int device, const DBConfig &config_p, size_t arena_bytes)
    : config(config_p), pool(device, arena_bytes), catalog(*this) {
	pool.cache_capacity = config.decoded_cache_bytes;
	pool.CacheReserve();
}

const CompressionFunction *DatabaseInstance::GetCompressionFunction(CompressionType type, PhysicalType data_type) {
	std::lock_guard<std::mutex> g(fn_lock);
	auto key = std::make_pair((uint8_t)type, (uint8_t)data_type);
	auto it = functions.find(key);
	if (it == functions.end()) {
		CompressionFunction fn = type == CompressionType::COMPRESSION_SUCCINCT ? SuccinctFun::GetFunction(data_type)
		                                                                       : UncompressedFun::GetFunction(data_type);
		it = functions.emplace(key, fn).first;
	}
	return &it->second;
}

// ------------------------------------------------------------------------------------------------
// ColumnSegment
// ------------------------------------------------------------------------------------------------

std::unique_ptr<ColumnSegment> ColumnSegment::CreateTransientSegment(DatabaseInstance &db, PhysicalType type,
                                                                     idx_t start, idx_t segment_size) {
	// column_segment.cpp:45-82
	if (!SuccinctFun::TypeIsSupported(type)) throw InternalException("Unsupported type for the succinct codec");
	auto &config = db.config;
	const CompressionFunction *function;
	bool succinct_possible;
	if (config.succinct_enabled && !config.adaptive_succinct_compression_enabled) {
		succinct_possible = true;
		function = db.GetCompressionFunction(CompressionType::COMPRESSION_SUCCINCT, type);
	} else {
		succinct_possible = config.succinct_enabled;
		function = db.GetCompressionFunction(CompressionType::COMPRESSION_UNCOMPRESSED, type);
	}
	return std::unique_ptr<ColumnSegment>(new ColumnSegment(db, type, start, segment_size, function, succinct_possible,
	                                                        config.adaptive_succinct_compression_enabled));
}

ColumnSegment::ColumnSegment(DatabaseInstance &db_p, PhysicalType type_p, idx_t start_p, idx_t segment_size_p,
                             const CompressionFunction *fn, bool succinct_possible_p, bool background_p)
    : db(db_p), type(type_p), type_size(adac_type_size((int)type_p)), start(start_p), function(fn),
      succinct_possible(succinct_possible_p), segment_size(segment_size_p), background_compaction_enabled(background_p) {
	raw.assign(segment_size, 0);
	if (function->type == CompressionType::COMPRESSION_SUCCINCT) {
		// column_segment.cpp:101-105: succinct_vec.width(8*type_size); resize(segment_size / type_size)
		vec_width = (uint8_t)(8 * type_size);
		vec_slots = segment_size / type_size;
		appended_via_succinct = true;
		db.data_size += (int64_t)adac_size_in_bytes(vec_slots, vec_width);
	} else {
		vec_width = 64;
		vec_slots = 0;
		db.data_size += (int64_t)segment_size; // AddOnlyToDataSize (column_segment.cpp:74)
	}
	db.catalog.AddColumnSegment(this);
}

void ColumnSegment::SetNext(ColumnSegment *next) {
	std::lock_guard<std::mutex> g(db.pool.lock); // the prefetch path follows the hint under this lock
	if (next_hint) next_hint->prev_hint = nullptr;
	next_hint = next;
	if (next) {
		if (next->prev_hint) next->prev_hint->next_hint = nullptr;
		next->prev_hint = this;
	}
}

ColumnSegment::~ColumnSegment() {
	db.catalog.RemoveColumnSegment(this);
	{
		std::lock_guard<std::mutex> g(db.pool.lock);
		if (prev_hint) prev_hint->next_hint = nullptr;
		if (next_hint) next_hint->prev_hint = nullptr;
	}
	if (packed_on_device) {
		std::lock_guard<std::mutex> g(db.pool.lock);
		db.pool.Free(word_off, arena_words);
	}
	if (db.pool.cache_capacity) {
		std::lock_guard<std::mutex> g(db.pool.lock);
		db.pool.CacheDrop(this);
	}
}

idx_t ColumnSegment::GetDataSize() const {
	// column_segment.cpp:204-214
	if (!is_data_segment) return 0;
	if (function->type == CompressionType::COMPRESSION_SUCCINCT) return adac_size_in_bytes(vec_slots, vec_width);
	return segment_size;
}

idx_t ColumnSegment::SuccinctSize() const {
	return function->type == CompressionType::COMPRESSION_SUCCINCT ? adac_size_in_bytes(vec_slots, vec_width) : 0;
}

void ColumnSegment::Scan(ColumnScanState &state, idx_t scan_count, Vector &result, idx_t result_offset,
                         bool entire_vector) {
	// column_segment.cpp:137-188
	db.catalog.AddReadAccess(this);
	if (!compacted && !background_compaction_enabled) Compact();
	{
		std::lock_guard<std::mutex> g(bit_compression_lock);
		force_reinitializing_scan_state = false; // scan states carry nothing in this codec
	}
	if (entire_vector) {
		function->scan_vector(*this, state, scan_count, result);
	} else {
		function->scan_partial(*this, state, scan_count, result, result_offset);
	}
}

void ColumnSegment::Skip(ColumnScanState &state) {
	function->skip(*this, state, 0);
}

void ColumnSegment::FetchRow(ColumnFetchState &state, row_t row_id, Vector &result, idx_t result_idx) {
	function->fetch_row(*this, state, row_id - (row_t)start, result, result_idx); // column_segment.cpp:193-195
}

void ColumnSegment::ScanRows(idx_t start_row, idx_t scan_count, data_ptr_t target) {
	if (start_row > count || scan_count > count - start_row) throw InternalException("scan beyond the segment");
	if (scan_count == 0) return;
	std::lock_guard<std::mutex> g(bit_compression_lock);
	if (function->type == CompressionType::COMPRESSION_SUCCINCT && packed_on_device) {
		std::lock_guard<std::mutex> pg(db.pool.lock);
		if (db.pool.cache_capacity) {
			// vector-serving cache: decode the WHOLE segment once, serve this and the following vectors by memcpy
			auto found = db.pool.cache.find(this);
			const bool was_prefetched = found != db.pool.cache.end() && found->second.pending;
			const uint8_t *hit = db.pool.CacheLookup(this); // waits for a prefetch in flight
			bool fresh = was_prefetched;
			if (!hit) {
				uint8_t *block = db.pool.CacheInsert(this, count * type_size);
				if (block) {
					void *d_all = db.pool.Staging(count * type_size);
					Check(adac_unpack_range(LayoutOf(device_layout), db.pool.d_arena, layout_index, 0, count, d_all, 0),
					      "adac_unpack_range");
					Check(adac_memcpy_d2h(db.pool.ctx, block, d_all, count * type_size), "adac_memcpy_d2h");
					for (auto &kv : db.pool.cache) kv.second.pending = false; // synchronous copy on the one stream
					hit = block;
					fresh = true;
				}
			}
			if (hit) {
				// first touch of this segment's image: once the rows are out, start the next segment's decode + copy,
				// so that it runs while the consumer reads this one (a sequential scan then waits for PCIe once)
				std::memcpy(target, hit + start_row * type_size, scan_count * type_size);
				if (fresh && next_hint) next_hint->PrefetchIntoCache(this);
				return;
			}
		}
		void *d_out = db.pool.Staging(scan_count * type_size);
		Check(adac_unpack_range(LayoutOf(device_layout), db.pool.d_arena, layout_index, start_row, scan_count, d_out, 0),
		      "adac_unpack_range");
		// the engine's result vector is pageable memory: copy down into the page-locked staging block (a direct
		// DMA) and move the few KiB from there, instead of letting the runtime stage the copy itself
		uint8_t *bounce = db.pool.PinnedStaging(scan_count * type_size);
		Check(adac_memcpy_d2h(db.pool.ctx, bounce, d_out, scan_count * type_size), "adac_memcpy_d2h");
		std::memcpy(target, bounce, scan_count * type_size);
	} else {
		// unpacked slots / uncompressed block: the bytes ARE the values (no min add: SURVEY.md §8a (iii))
		std::memcpy(target, raw.data() + start_row * type_size, scan_count * type_size);
	}
}

void ColumnSegment::PrefetchIntoCache(const ColumnSegment *reader) {
	// called with pool.lock held by a scan of the PREVIOUS segment (`reader`).  This segment's state is read under
	// its own lock, taken with try_lock only: a flip in progress on it holds that lock and may be waiting for
	// pool.lock.
	std::unique_lock<std::mutex> g(bit_compression_lock, std::try_to_lock);
	if (!g.owns_lock()) return;
	if (!function || function->type != CompressionType::COMPRESSION_SUCCINCT || !packed_on_device || count == 0) return;
	auto &pool = db.pool;
	if (pool.cache.count(this)) return;
	for (auto &kv : pool.cache) {
		if (kv.second.pending) return; // one prefetch in flight at a time (one staging buffer)
	}
	const size_t bytes = count * type_size;
	const bool slotted = pool.cache_slab && bytes <= SegmentPool::kCacheSlotBytes;
	const size_t charge = slotted ? SegmentPool::kCacheSlotBytes : bytes;
	const bool evicts = pool.cache_used + charge > pool.cache_capacity || (slotted && pool.cache_free_slots.empty());
	// never at the reader's expense: its block was touched last, so with two or more entries the victim is another
	if (evicts && (pool.cache.size() < 2 || !pool.cache.count(reader))) return;
	uint8_t *block = pool.CacheInsert(this, bytes);
	if (!block) return;
	void *d_all = pool.PrefetchStaging(bytes);
	Check(adac_unpack_range(LayoutOf(device_layout), pool.d_arena, layout_index, 0, count, d_all, 0), "adac_unpack_range");
	Check(adac_memcpy_d2h_async(pool.ctx, block, d_all, bytes), "adac_memcpy_d2h_async");
	pool.cache[this].pending = true;
	pool.cache_misses++; // a device decode of this segment, as a miss would have been; the look-up will be a hit
	pool.cache_prefetches++;
}

idx_t ColumnSegment::AppendRows(UnifiedVectorFormat &data, idx_t offset, idx_t append_count) {
	// SuccinctAppend (succinct.cpp:308-322) / FixedSizeAppend: no arithmetic here — the running min/max of
	// SuccinctAppendLoop is produced by adac_analyze when the segment compacts.
	idx_t max_tuple_count = segment_size / type_size;
	idx_t copy_count = std::min<idx_t>(append_count, max_tuple_count - count);
	const bool track_validity = function->type == CompressionType::COMPRESSION_SUCCINCT;
	const uint64_t null_bits = NullBits(type, type_size);
	if (track_validity && validity.empty()) validity.assign((max_tuple_count + 63) / 64 + 1, ~0ull);
	for (idx_t i = 0; i < copy_count; i++) {
		idx_t source_idx = data.sel ? data.sel[offset + i] : offset + i;
		idx_t target_idx = count + i;
		bool valid = !data.validity || ((data.validity[source_idx >> 6] >> (source_idx & 63)) & 1);
		if (valid) {
			std::memcpy(raw.data() + target_idx * type_size, data.data + source_idx * type_size, type_size);
		} else {
			std::memcpy(raw.data() + target_idx * type_size, &null_bits, type_size); // NullValue<T>()
			if (track_validity) {
				validity[target_idx >> 6] &= ~(1ull << (target_idx & 63));
				any_null = true;
			}
		}
	}
	count += copy_count;
	return copy_count;
}

idx_t ColumnSegment::Append(UnifiedVectorFormat &append_data, idx_t offset, idx_t append_count) {
	// column_segment.cpp:247-271.  The whole append is one flip-free section: the background policy thread must
	// not compact the segment between the Uncompact below and the write into the unpacked image (in the
	// reference that window is open: its TSan suppressions cover it)
	std::lock_guard<std::recursive_mutex> flips(db.flip_lock);
	bool uncompacted = false;
	if (IsBitCompressed()) {
		Uncompact();
		uncompacted = true;
	}
	idx_t copy_count = function->append(*this, append_data, offset, append_count);
	num_elements += append_count; // sic: the requested count
	if (!compacted && !background_compaction_enabled && (num_elements >= vec_slots || uncompacted)) {
		Compact();
	}
	return copy_count;
}

idx_t ColumnSegment::FinalizeAppend() {
	return function->finalize_append(*this);
}

bool ColumnSegment::NeedsCompaction() const {
	// column_segment.cpp:278-280
	return !(compacted || !function || num_elements == 0 || !succinct_possible);
}

void ColumnSegment::Compact() {
	std::vector<ColumnSegment *> one {this};
	CompactMany(db, one);
}

void ColumnSegment::CompactMany(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments) {
	// Batched ColumnSegment::Compact (column_segment.cpp:273-322): per (type, rule) group one upload, one
	// adac_analyze, the width decision on the host from the downloaded min/max, one adac_pack.
	std::lock_guard<std::recursive_mutex> flips(db.flip_lock);
	std::map<std::pair<uint8_t, int>, std::vector<ColumnSegment *>> groups;
	for (auto *s : segments) {
		if (!s->NeedsCompaction()) continue;
		int rule = s->function->type == CompressionType::COMPRESSION_SUCCINCT ? ADAC_RULE_APPEND : ADAC_RULE_RECOMPACT;
		groups[{(uint8_t)s->type, rule}].push_back(s);
	}
	const bool padded = db.config.succinct_padded_to_next_byte_enabled;
	for (auto &g : groups) {
		PhaseTrace trace;
		const int ptype = g.first.first;
		const int rule = g.first.second;
		auto &segs = g.second;
		const idx_t ts = adac_type_size(ptype);
		const idx_t per16 = 16 / ts;
		std::vector<uint32_t> counts(segs.size());
		std::vector<uint64_t> offs(segs.size());
		uint64_t span = 0;
		bool any_null = false;
		for (size_t i = 0; i < segs.size(); i++) {
			counts[i] = (uint32_t)segs[i]->count;
			offs[i] = span;
			span += (segs[i]->count + per16 - 1) / per16 * per16; // keep every segment 16-byte aligned
			any_null |= (rule == ADAC_RULE_APPEND && segs[i]->any_null);
		}
		const size_t host_bytes = span * ts + 16;
		std::vector<uint64_t> vmask;
		if (any_null) vmask.assign(span / 64 + 2, ~0ull);
		if (any_null) {
			for (size_t i = 0; i < segs.size(); i++) {
				if (!segs[i]->any_null) continue;
				for (idx_t r = 0; r < segs[i]->count; r++) {
					if (!((segs[i]->validity[r >> 6] >> (r & 63)) & 1)) {
						uint64_t e = offs[i] + r;
						vmask[e >> 6] &= ~(1ull << (e & 63));
					}
				}
			}
		}
		std::vector<uint64_t> mm(2 * segs.size());
		std::vector<adac_segment_desc> descs;
		std::vector<size_t> pidx;
		std::vector<uint8_t> widths(segs.size());
		std::shared_ptr<LayoutHandle> handle;
		{ // device work under the pool lock; representation flips (bit_compression_lock) after it is released
		std::lock_guard<std::mutex> pg(db.pool.lock);
		adac_ctx *ctx = db.pool.ctx;
		trace.mark("prep");
		// gather the segments' rows into page-locked staging (a few host threads for big batches: the gather,
		// not PCIe, bounds a re-compaction round) and upload them with one copy at PCIe rate
		uint8_t *host = db.pool.PinnedStaging(host_bytes);
		trace.mark("pinned_alloc");
		{
			auto gather = [&](size_t lo, size_t hi) {
				for (size_t i = lo; i < hi; i++) {
					std::memcpy(host + offs[i] * ts, segs[i]->raw.data(), segs[i]->count * ts);
					const size_t end = (offs[i] + segs[i]->count) * ts;
					const size_t next = i + 1 < segs.size() ? offs[i + 1] * ts : host_bytes;
					std::memset(host + end, 0, next - end); // alignment gap
				}
			};
			size_t nthreads = host_bytes > (8u << 20) ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
			nthreads = std::min(nthreads, segs.size());
			if (nthreads <= 1) {
				gather(0, segs.size());
			} else {
				std::vector<std::thread> th;
				for (size_t k = 0; k < nthreads; k++) {
					th.emplace_back(gather, segs.size() * k / nthreads, segs.size() * (k + 1) / nthreads);
				}
				for (auto &t : th) t.join();
			}
		}
		trace.mark("gather");
		void *d_vals = db.pool.Staging(host_bytes);
		trace.mark("dev_alloc");
		Check(adac_memcpy_h2d(ctx, d_vals, host, host_bytes), "upload rows");
		trace.mark("h2d");
		uint64_t *d_valid = nullptr;
		if (any_null) {
			d_valid = static_cast<uint64_t *>(db.pool.Staging2(vmask.size() * 8));
			Check(adac_memcpy_h2d(ctx, d_valid, vmask.data(), vmask.size() * 8), "upload validity");
		}
		adac_layout *probe = nullptr;
		Check(adac_layout_create(ctx, ptype, counts.data(), offs.data(), segs.size(), &probe), "adac_layout_create");
		adac_status st = adac_analyze(probe, d_vals, d_valid, rule);
		if (st == ADAC_OK) st = adac_layout_get_minmax(probe, mm.data());
		adac_layout_destroy(probe);
		Check(st, "adac_analyze");
		trace.mark("analyze");
		// width decision (column_segment.cpp:351-363 / :404-420) and arena placement
		std::vector<uint32_t> pcounts;
		std::vector<uint64_t> poffs;
		for (size_t i = 0; i < segs.size(); i++) {
			uint8_t w = adac_width(mm[2 * i], mm[2 * i + 1], rule, padded);
			widths[i] = w;
			if (8 * ts > w) {
				adac_segment_desc d;
				d.word_off = db.pool.Allocate(adac_arena_words(counts[i], w));
				d.val_off = offs[i];
				d.min = adac_stored_min(mm[2 * i], mm[2 * i + 1], w); // all-ones segments: see adacodec.h
				d.count = counts[i];
				d.width = w;
				d.flags = ADAC_SEG_PACKED;
				d.reserved = 0;
				descs.push_back(d);
				pcounts.push_back(counts[i]);
				poffs.push_back(offs[i]);
				pidx.push_back(i);
			}
		}
		if (!descs.empty()) {
			handle = std::make_shared<LayoutHandle>();
			Check(adac_layout_create(ctx, ptype, pcounts.data(), poffs.data(), descs.size(), &handle->layout),
			      "adac_layout_create");
			Check(adac_layout_set_descs(handle->layout, descs.data()), "adac_layout_set_descs");
			trace.mark("alloc+layout");
			Check(adac_pack(handle->layout, d_vals, d_valid, db.pool.d_arena), "adac_pack");
			Check(adac_ctx_sync(ctx), "adac_ctx_sync");
			trace.mark("pack");
		}
		} // pool lock released
		size_t p = 0;
		// the unpacked images of a big batch go back to the OS off the critical path: unmapping a 256 KiB block
		// costs ~18 us, 6.5 ms for the 360 segments of a first policy round
		std::vector<std::vector<uint8_t>> graveyard;
		if (segs.size() > 8) graveyard.reserve(segs.size());
		for (size_t i = 0; i < segs.size(); i++) {
			bool packed = p < pidx.size() && pidx[p] == i;
			if (packed) {
				segs[i]->device_layout = handle;
				segs[i]->layout_index = p;
			}
			segs[i]->FinishCompaction(packed, widths[i], mm[2 * i], mm[2 * i + 1], rule,
			                          packed ? descs[p].word_off : 0, segs.size() > 8 ? &graveyard : nullptr);
			if (packed) p++;
		}
		if (!graveyard.empty()) {
			std::thread([g = std::move(graveyard)]() mutable { g.clear(); }).detach();
		}
		trace.mark("finish");
	}
}

void ColumnSegment::FinishCompaction(bool packed, uint8_t width, uint64_t mn, uint64_t mx, int rule, uint64_t off,
                                     std::vector<std::vector<uint8_t>> *graveyard) {
	std::lock_guard<std::mutex> g(bit_compression_lock);
	const idx_t before = GetDataSize();
	if (rule == ADAC_RULE_APPEND) {
		// what UpdateMinFactor/UpdateMaxFactor accumulated during the appends (succinct.cpp:317-318)
		min_factor = std::min(min_factor, mn);
		max_factor = std::max(max_factor, mx);
	} else {
		// succinct_vec.width(8*type_size); resize(segment_size / type_size)  (column_segment.cpp:304-305)
		vec_width = (uint8_t)(8 * type_size);
		vec_slots = segment_size / type_size;
		// the product stores the frame of reference of the recompaction (reference defect 2)
		min_factor = packed ? mn : UINT64_MAX;
	}
	if (packed) {
		vec_slots = count; // bit_resize(count * w)
		vec_width = width;
		packed_on_device = true;
		word_off = off;
		arena_words = adac_arena_words(count, width);
		// the unpacked image is gone, as after SDSL's realloc shrink
		if (graveyard) {
			graveyard->emplace_back(std::move(raw));
			raw = std::vector<uint8_t>();
		} else {
			std::vector<uint8_t>().swap(raw);
		}
	}
	std::vector<uint64_t>().swap(validity);
	any_null = false;
	function = db.GetCompressionFunction(CompressionType::COMPRESSION_SUCCINCT, type);
	compacted = true;
	db.data_size += (int64_t)GetDataSize() - (int64_t)before;
}

void ColumnSegment::Uncompact() {
	// column_segment.cpp:324-346 + UncompressSuccinct :458-506
	std::lock_guard<std::recursive_mutex> flips(db.flip_lock);
	if (!compacted || !function || function->type != CompressionType::COMPRESSION_SUCCINCT) return;
	std::lock_guard<std::mutex> g(bit_compression_lock);
	const idx_t compressed_size = adac_size_in_bytes(vec_slots, vec_width);
	if (packed_on_device) {
		raw.assign(segment_size, 0);
		std::lock_guard<std::mutex> pg(db.pool.lock);
		if (count) {
			void *d_out = db.pool.Staging(count * type_size);
			Check(adac_unpack_range(LayoutOf(device_layout), db.pool.d_arena, layout_index, 0, count, d_out, 0),
			      "adac_unpack_range");
			uint8_t *bounce = db.pool.PinnedStaging(count * type_size);
			Check(adac_memcpy_d2h(db.pool.ctx, bounce, d_out, count * type_size), "adac_memcpy_d2h");
			std::memcpy(raw.data(), bounce, count * type_size);
		}
		db.pool.Free(word_off, arena_words);
		db.pool.CacheDrop(this);
		packed_on_device = false;
		device_layout.reset();
	}
	function = db.GetCompressionFunction(CompressionType::COMPRESSION_UNCOMPRESSED, type);
	compacted = false;
	vec_slots = 0; // succinct_vec.resize(0)
	force_reinitializing_scan_state = true;
	db.data_size += (int64_t)segment_size - (int64_t)compressed_size;
}

// ------------------------------------------------------------------------------------------------
// ColumnSegmentCatalog
// ------------------------------------------------------------------------------------------------

ColumnSegmentCatalog::ColumnSegmentCatalog(DatabaseInstance &db_p) : db(db_p) {
}

ColumnSegmentCatalog::~ColumnSegmentCatalog() {
	DisableBackgroundThreadCompaction();
}

void ColumnSegmentCatalog::AddColumnSegment(ColumnSegment *segment) {
	if (!segment->is_data_segment) return;
	std::lock_guard<std::mutex> g(lock);
	statistics[segment] = AccessStatistics {0};
}

void ColumnSegmentCatalog::RemoveColumnSegment(ColumnSegment *segment) {
	std::lock_guard<std::mutex> g(lock);
	statistics.erase(segment);
}

void ColumnSegmentCatalog::AddReadAccess(ColumnSegment *segment) {
	// column_segment_catalog.cpp:37-54
	if (segment == nullptr || !segment->is_data_segment) return;
	std::lock_guard<std::mutex> g(lock);
	auto it = statistics.find(segment);
	if (it == statistics.end()) {
		statistics[segment] = AccessStatistics {1};
	} else {
		it->second.num_reads++;
		event_counter++;
	}
}

idx_t ColumnSegmentCatalog::NumSegments() {
	std::lock_guard<std::mutex> g(lock);
	return statistics.size();
}

void ColumnSegmentCatalog::CompactAllSegments() {
	std::vector<ColumnSegment *> all;
	{
		std::lock_guard<std::mutex> g(lock);
		for (auto &e : statistics) all.push_back(e.first);
	}
	ColumnSegment::CompactMany(db, all);
}

size_t ColumnSegmentCatalog::GetTotalDataSize() {
	std::lock_guard<std::mutex> g(lock);
	size_t data_size = 0;
	for (auto &e : statistics) data_size += e.first->GetDataSize();
	return data_size;
}

void ColumnSegmentCatalog::CompressLowestKSegmentsOnce(double compression_rate) {
	// column_segment_catalog.cpp:79-112.  The reference sorts by num_reads only (ties in unordered_map
	// order); ties are broken here by the segment's start row, then address, to be deterministic.
	std::vector<std::pair<ColumnSegment *, AccessStatistics>> v;
	{
		std::lock_guard<std::mutex> g(lock);
		v.assign(statistics.begin(), statistics.end());
	}
	std::sort(v.begin(), v.end(), [](const std::pair<ColumnSegment *, AccessStatistics> &l,
	                                 const std::pair<ColumnSegment *, AccessStatistics> &r) {
		if (l.second.num_reads != r.second.num_reads) return l.second.num_reads < r.second.num_reads;
		if (l.first->start != r.first->start) return l.first->start < r.first->start;
		return l.first < r.first;
	});
	std::vector<ColumnSegment *> to_compact, to_uncompact;
	float cum_sum = 0;
	idx_t curr_counter = v.size();
	for (auto &e : v) {
		cum_sum += 1;
		if (cum_sum / curr_counter < compression_rate) {
			to_compact.push_back(e.first);
		} else {
			to_uncompact.push_back(e.first);
		}
	}
	ColumnSegment::CompactMany(db, to_compact);
	for (auto *s : to_uncompact) s->Uncompact();
	{
		std::lock_guard<std::mutex> g(lock);
		for (auto &e : v) {
			auto it = statistics.find(e.first);
			if (it != statistics.end()) it->second.num_reads = 0;
		}
	}
	event_counter = 0;
}

void ColumnSegmentCatalog::EnableBackgroundThreadCompaction(unsigned period_ms) {
	// column_segment_catalog.cpp:13-22 (the reference detaches the thread and sleeps 10 s per round)
	if (background_compaction_enabled.exchange(true)) return;
	stop = false;
	worker = std::thread([this, period_ms]() {
		while (!stop) {
			for (unsigned waited = 0; waited < period_ms && !stop; waited += 5) {
				std::this_thread::sleep_for(std::chrono::milliseconds(5));
			}
			if (stop) break;
			CompressLowestKSegmentsOnce(0.90);
		}
	});
}

void ColumnSegmentCatalog::DisableBackgroundThreadCompaction() {
	stop = true;
	if (worker.joinable()) worker.join();
	background_compaction_enabled = false;
}

} // namespace adacodec

// ------------------------------------------------------------------------------------------------
// C wrappers (include/adacodec_host.h): let non-C++ hosts and the Python tests drive the mirror.
// ------------------------------------------------------------------------------------------------

using namespace adacodec;

struct adach_db {
	std::unique_ptr<DatabaseInstance> db;
};
struct adach_segment {
	std::unique_ptr<ColumnSegment> seg;
};

static thread_local std::string g_host_error;

template <typename F>
static int Guard(F &&f) {
	try {
		f();
		return 0;
	} catch (const std::exception &e) {
		g_host_error = e.what();
		return 1;
	}
}

extern "C" const char *adach_last_error(void) {
	return g_host_error.c_str();
}

extern "C" adach_db *adach_db_create(int device, int succinct_enabled, int adaptive, int padded, uint64_t arena_bytes) {
	adach_db *h = nullptr;
	Guard([&]() {
		DBConfig cfg;
		cfg.succinct_enabled = succinct_enabled != 0;
		cfg.adaptive_succinct_compression_enabled = adaptive != 0;
		cfg.succinct_padded_to_next_byte_enabled = padded != 0;
		auto db = std::unique_ptr<DatabaseInstance>(new DatabaseInstance(device, cfg, arena_bytes));
		h = new adach_db {std::move(db)};
	});
	return h;
}

extern "C" adach_db *adach_db_create_cached(int device, int succinct_enabled, int adaptive, int padded,
                                            uint64_t arena_bytes, uint64_t decoded_cache_bytes) {
	adach_db *h = nullptr;
	Guard([&]() {
		DBConfig cfg;
		cfg.succinct_enabled = succinct_enabled != 0;
		cfg.adaptive_succinct_compression_enabled = adaptive != 0;
		cfg.succinct_padded_to_next_byte_enabled = padded != 0;
		cfg.decoded_cache_bytes = decoded_cache_bytes;
		auto db = std::unique_ptr<DatabaseInstance>(new DatabaseInstance(device, cfg, arena_bytes));
		h = new adach_db {std::move(db)};
	});
	return h;
}

extern "C" void adach_db_cache_stats(adach_db *h, uint64_t *hits, uint64_t *misses, uint64_t *bytes) {
	std::lock_guard<std::mutex> g(h->db->pool.lock);
	if (hits) *hits = h->db->pool.cache_hits;
	if (misses) *misses = h->db->pool.cache_misses;
	if (bytes) *bytes = h->db->pool.cache_used;
}

// Full scan of a list of segments in the engine's call pattern — ColumnSegment::Scan on vector_size-row vectors
// (ColumnData::ScanVector, column_data.cpp:92-139) — timed on the host; checksum = wrapping sum of all rows.
template <typename T>
static uint64_t SumTyped(const uint8_t *p, idx_t n) {
	uint64_t acc = 0;
	for (idx_t k = 0; k < n; k++) { // unaligned-safe, vectorises
		T x;
		std::memcpy(&x, p + k * sizeof(T), sizeof(T));
		acc += x;
	}
	return acc;
}
static uint64_t SumVector(const uint8_t *p, idx_t n, idx_t type_size) {
	switch (type_size) {
	case 1: return SumTyped<uint8_t>(p, n);
	case 2: return SumTyped<uint16_t>(p, n);
	case 4: return SumTyped<uint32_t>(p, n);
	default: return SumTyped<uint64_t>(p, n);
	}
}

extern "C" int adach_full_scan(adach_segment **segs, uint64_t nseg, uint64_t vector_size, uint64_t *checksum,
                               double *seconds, uint64_t *rows_out) {
	return Guard([&]() {
		std::vector<uint8_t> vec(vector_size * 8 + 64);
		uint64_t sum = 0, rows = 0;
		auto t0 = std::chrono::steady_clock::now();
		for (uint64_t i = 0; i < nseg; i++) {
			ColumnSegment &s = *segs[i]->seg;
			for (idx_t r = 0; r < s.count; r += vector_size) {
				idx_t c = std::min<idx_t>(vector_size, s.count - r);
				ColumnScanState st;
				st.row_index = s.start + r;
				Vector v;
				v.data = vec.data();
				s.Scan(st, c, v, 0, true);
				sum += SumVector(vec.data(), c, s.type_size); // the consumer touches every value
				rows += c;
			}
		}
		auto t1 = std::chrono::steady_clock::now();
		if (checksum) *checksum = sum;
		if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
		if (rows_out) *rows_out = rows;
	});
}

extern "C" void adach_db_destroy(adach_db *h) {
	delete h;
}

extern "C" adach_segment *adach_segment_create(adach_db *h, int physical_type, uint64_t start, uint64_t segment_size) {
	adach_segment *s = nullptr;
	Guard([&]() {
		if (!adac_type_is_supported(physical_type)) throw InternalException("Unsupported type for the succinct codec");
		auto seg = ColumnSegment::CreateTransientSegment(*h->db, (PhysicalType)physical_type, start, segment_size);
		s = new adach_segment {std::move(seg)};
	});
	return s;
}

extern "C" int adach_compress_column(adach_db *h, int compression_type, int physical_type, uint64_t row_group_start,
                                     const void *values, const uint64_t *validity, uint64_t n,
                                     adach_segment **out_segments, uint64_t max_segments, uint64_t *out_nseg,
                                     uint64_t *out_sizes, uint64_t *out_score) {
	return Guard([&]() {
		// the checkpoint pipeline of ColumnDataCheckpointer::WriteToDisk (column_data_checkpointer.cpp): analyze
		// every vector, score, then compress every vector through the winning function's slots
		auto type = (PhysicalType)physical_type;
		const CompressionFunction *fn = h->db->GetCompressionFunction((CompressionType)compression_type, type);
		const idx_t tsize = adac_type_size(physical_type);
		auto astate = fn->init_analyze(type);
		for (idx_t off = 0; off < n; off += STANDARD_VECTOR_SIZE) {
			idx_t c = std::min<idx_t>(STANDARD_VECTOR_SIZE, n - off);
			Vector v;
			v.data = (data_ptr_t)values + off * tsize;
			if (!fn->analyze(*astate, v, c)) throw InternalException("analyze refused the column");
		}
		if (out_score) *out_score = fn->final_analyze(*astate);
		ColumnDataCheckpointer checkpointer(*h->db, type, row_group_start);
		auto cstate = fn->init_compression(checkpointer, std::move(astate));
		std::vector<uint64_t> shifted;
		for (idx_t off = 0; off < n; off += STANDARD_VECTOR_SIZE) {
			idx_t c = std::min<idx_t>(STANDARD_VECTOR_SIZE, n - off);
			Vector v;
			v.data = (data_ptr_t)values + off * tsize;
			if (validity) { // the vector's own mask starts at its first row (2048 | 64: whole words)
				v.validity = validity + off / 64;
			}
			fn->compress(*cstate, v, c);
		}
		fn->compress_finalize(*cstate);
		if (checkpointer.flushed_segments.size() > max_segments) throw InternalException("segment array too small");
		*out_nseg = checkpointer.flushed_segments.size();
		for (size_t i = 0; i < checkpointer.flushed_segments.size(); i++) {
			out_segments[i] = new adach_segment {std::move(checkpointer.flushed_segments[i])};
			if (out_sizes) out_sizes[i] = checkpointer.flushed_sizes[i];
		}
	});
}

extern "C" int adach_db_reserve_staging(adach_db *h, uint64_t bytes) {
	return Guard([&]() {
		// page-locking the upload buffer is the one slow step of a first compaction round (hipHostMalloc: ~20 ms
		// per 100 MB): an engine sizes it once, here, instead of inside the first policy step
		std::lock_guard<std::mutex> pg(h->db->pool.lock);
		h->db->pool.PinnedStaging(bytes);
		h->db->pool.Staging(bytes);
	});
}

extern "C" int adach_function_slots(adach_db *h, int compression_type, int physical_type, int *present) {
	return Guard([&]() {
		const CompressionFunction *fn =
		    h->db->GetCompressionFunction((CompressionType)compression_type, (PhysicalType)physical_type);
		const void *slots[16] = {(const void *)fn->init_analyze, (const void *)fn->analyze,
		                         (const void *)fn->final_analyze, (const void *)fn->init_compression,
		                         (const void *)fn->compress, (const void *)fn->compress_finalize,
		                         (const void *)fn->init_scan, (const void *)fn->scan_vector,
		                         (const void *)fn->scan_partial, (const void *)fn->fetch_row,
		                         (const void *)fn->skip, (const void *)fn->init_segment,
		                         (const void *)fn->init_append, (const void *)fn->append,
		                         (const void *)fn->finalize_append, (const void *)fn->revert_append};
		for (int i = 0; i < 16; i++) present[i] = slots[i] != nullptr;
	});
}

extern "C" int adach_segment_set_next(adach_segment *s, adach_segment *next) {
	return Guard([&]() { s->seg->SetNext(next ? next->seg.get() : nullptr); });
}

extern "C" void adach_segment_destroy(adach_segment *s) {
	delete s;
}

extern "C" int64_t adach_segment_append(adach_segment *s, const void *vals, const uint64_t *validity, const uint32_t *sel,
                                        uint64_t offset, uint64_t count) {
	int64_t copied = -1;
	Guard([&]() {
		UnifiedVectorFormat f;
		f.data = static_cast<const uint8_t *>(vals);
		f.validity = validity;
		f.sel = sel;
		copied = (int64_t)s->seg->Append(f, offset, count);
	});
	return copied;
}

extern "C" int adach_segment_scan(adach_segment *s, uint64_t row_index, uint64_t count, void *result,
                                  uint64_t result_offset, int entire_vector) {
	return Guard([&]() {
		ColumnScanState st;
		st.row_index = row_index;
		Vector v;
		v.data = static_cast<data_ptr_t>(result);
		s->seg->Scan(st, count, v, result_offset, entire_vector != 0);
	});
}

extern "C" int adach_segment_fetch_row(adach_segment *s, int64_t row_id, void *result, uint64_t result_idx) {
	return Guard([&]() {
		ColumnFetchState st;
		Vector v;
		v.data = static_cast<data_ptr_t>(result);
		s->seg->FetchRow(st, row_id, v, result_idx);
	});
}

extern "C" int adach_segment_compact(adach_segment *s) {
	return Guard([&]() { s->seg->Compact(); });
}
extern "C" int adach_segment_uncompact(adach_segment *s) {
	return Guard([&]() { s->seg->Uncompact(); });
}
extern "C" uint64_t adach_segment_count(adach_segment *s) {
	return s->seg->count;
}
extern "C" uint64_t adach_segment_min(adach_segment *s) {
	return s->seg->GetMinFactor();
}
extern "C" uint64_t adach_segment_max(adach_segment *s) {
	return s->seg->GetMax();
}
extern "C" uint32_t adach_segment_width(adach_segment *s) {
	return s->seg->Width();
}
extern "C" int adach_segment_compacted(adach_segment *s) {
	return s->seg->IsBitCompressed();
}
extern "C" int adach_segment_function(adach_segment *s) {
	return (int)s->seg->function->type;
}
extern "C" uint64_t adach_segment_data_size(adach_segment *s) {
	return s->seg->GetDataSize();
}

extern "C" int adach_catalog_compact_all(adach_db *h) {
	return Guard([&]() { h->db->catalog.CompactAllSegments(); });
}
extern "C" uint64_t adach_catalog_total_data_size(adach_db *h) {
	return h->db->catalog.GetTotalDataSize();
}
extern "C" uint64_t adach_catalog_num_segments(adach_db *h) {
	return h->db->catalog.NumSegments();
}
extern "C" int adach_catalog_policy_step(adach_db *h, double compression_rate) {
	return Guard([&]() { h->db->catalog.CompressLowestKSegmentsOnce(compression_rate); });
}
extern "C" int adach_catalog_enable_background(adach_db *h, unsigned period_ms) {
	return Guard([&]() { h->db->catalog.EnableBackgroundThreadCompaction(period_ms); });
}
extern "C" int adach_catalog_disable_background(adach_db *h) {
	return Guard([&]() { h->db->catalog.DisableBackgroundThreadCompaction(); });
}
extern "C" int64_t adach_db_data_size(adach_db *h) {
	return h->db->data_size.load();
}
extern "C" uint64_t adach_db_arena_used_bytes(adach_db *h) {
	return h->db->pool.UsedWords() * 8;
}
extern "C" int adach_type_is_supported(int physical_type) {
	return SuccinctFun::TypeIsSupported((PhysicalType)physical_type) ? 1 : 0;
}
