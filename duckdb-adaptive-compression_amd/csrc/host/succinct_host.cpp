// succinct_host.cpp — see succinct_host.hpp.  Device work goes through include/adacodec.h only; the host
// side moves bytes (memcpy, NULL-slot fill) and keeps the reference's state machine and accounting.
#include "succinct_host.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <tuple>

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
// DecodeBatch
// ------------------------------------------------------------------------------------------------

DecodeBatch::~DecodeBatch() {
	if (ev) adac_event_destroy(ev);
}

void DecodeBatch::Publish(adac_event *e, bool ok) {
	{
		std::lock_guard<std::mutex> g(m);
		ev = e;
		state = ok ? 1 : -1;
	}
	cv.notify_all();
}

bool DecodeBatch::Wait() {
	std::unique_lock<std::mutex> g(m);
	cv.wait(g, [&] { return state != 0; });
	if (state == 2) return true;
	if (state < 0) return false;
	adac_event *e = ev;
	g.unlock();
	const bool ok = adac_event_wait(e) == ADAC_OK; // any number of threads may wait on the same event
	g.lock();
	if (state == 1) state = ok ? 2 : -1;
	return state == 2;
}

bool DecodeBatch::Finished() {
	std::lock_guard<std::mutex> g(m);
	if (state == 2 || state < 0) return true;
	if (state == 1 && adac_event_done(ev)) {
		state = 2;
		return true;
	}
	return false;
}

// ------------------------------------------------------------------------------------------------
// SegmentPool
// ------------------------------------------------------------------------------------------------

SegmentPool::SegmentPool(int index_p, int device_p, size_t arena_bytes, const DBConfig &config)
    : index(index_p), device(device_p), prefetch_segments(std::max<uint32_t>(1, std::min<uint32_t>(config.prefetch_segments, 48))) {
	Check(adac_ctx_create(device, nullptr, &ctx), "adac_ctx_create");
	try {
		arena_words = ((arena_bytes / 8) + 15) & ~15ull;
		if (arena_words < 16) arena_words = 16;
		void *p = nullptr;
		Check(adac_dev_alloc(ctx, arena_words * 8, &p), "adac_dev_alloc(arena)");
		d_arena = static_cast<uint64_t *>(p);
		Check(adac_dev_memset(ctx, d_arena, 0, arena_words * 8), "adac_dev_memset(arena)");
		Check(adac_ctx_sync(ctx), "adac_ctx_sync");
		free_list[0] = arena_words;
		cache_capacity = config.decoded_cache_bytes;
		if (cache_capacity >= kCacheSlotBytes) {
			// the cache's page-locked memory is ONE slab cut into block-sized slots: page-locking costs ~50 us per
			// 256 KiB block, which made every cold segment pay more for its buffer than for its decode and copy
			const size_t nslots = cache_capacity / kCacheSlotBytes;
			void *slab = nullptr;
			Check(adac_host_alloc_pinned(ctx, nslots * kCacheSlotBytes, &slab), "adac_host_alloc_pinned(cache)");
			cache_slab = static_cast<uint8_t *>(slab);
			for (size_t i = 0; i < nslots; i++) free_slots.insert((int32_t)i);
			const uint32_t nlanes = std::max<uint32_t>(1, std::min<uint32_t>(config.scan_lanes, 16));
			for (uint32_t i = 0; i < nlanes; i++) {
				std::unique_ptr<ScanLane> lane(new ScanLane());
				Check(adac_ctx_create(device, nullptr, &lane->ctx), "adac_ctx_create(lane)");
				lanes.push_back(std::move(lane)); // owned from here on: the destructor releases what exists
				Check(adac_dev_alloc(lanes.back()->ctx, (size_t)prefetch_segments * kCacheSlotBytes + 256, &lanes.back()->d_stage),
				      "adac_dev_alloc(lane staging)");
			}
		} else {
			cache_capacity = 0;
		}
	} catch (...) {
		Release();
		throw;
	}
}

SegmentPool::~SegmentPool() {
	Release();
}

void SegmentPool::Release() {
	for (auto &lane : lanes) {
		if (!lane || !lane->ctx) continue;
		adac_ctx_sync(lane->ctx); // no copy may still be writing the slab
		if (lane->d_stage) adac_dev_free(lane->ctx, lane->d_stage);
		adac_ctx_destroy(lane->ctx);
		lane->ctx = nullptr;
	}
	lanes.clear();
	for (auto &kv : cache) delete kv.second;
	cache.clear();
	for (auto *z : zombies) delete z;
	zombies.clear();
	if (!ctx) return;
	if (cache_slab) adac_host_free_pinned(ctx, cache_slab);
	if (h_pinned) adac_host_free_pinned(ctx, h_pinned);
	if (d_staging) adac_dev_free(ctx, d_staging);
	if (d_staging2) adac_dev_free(ctx, d_staging2);
	if (d_arena) adac_dev_free(ctx, d_arena);
	adac_ctx_destroy(ctx);
	ctx = nullptr;
	cache_slab = nullptr;
	h_pinned = nullptr;
	d_staging = d_staging2 = nullptr;
	d_arena = nullptr;
}

bool SegmentPool::TryAllocate(uint64_t words, uint64_t &word_off) {
	words = (words + 15) & ~15ull;
	std::lock_guard<std::mutex> g(arena_lock);
	for (auto it = free_list.begin(); it != free_list.end(); ++it) {
		if (it->second >= words) {
			uint64_t off = it->first, len = it->second;
			free_list.erase(it);
			if (len > words) free_list[off + words] = len - words;
			used_words += words;
			word_off = off;
			return true;
		}
	}
	return false;
}

void SegmentPool::Free(uint64_t off, uint64_t words) {
	words = (words + 15) & ~15ull;
	std::lock_guard<std::mutex> g(arena_lock);
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
	free_generation++;
}

uint64_t SegmentPool::UsedWords() {
	std::lock_guard<std::mutex> g(arena_lock);
	return used_words;
}

static void *Grow(adac_ctx *ctx, void *&buf, size_t &have, size_t want) {
	if (want > have) {
		size_t n = std::max(want, have * 2);
		n = (n + 255) & ~size_t(255);
		if (buf) adac_dev_free(ctx, buf);
		buf = nullptr;
		have = 0;
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
		pinned_bytes = 0;
		n = (n + 4095) & ~size_t(4095);
		Check(adac_host_alloc_pinned(ctx, n, &h_pinned), "adac_host_alloc_pinned(staging)");
		pinned_bytes = n;
	}
	return static_cast<uint8_t *>(h_pinned);
}

void *SegmentPool::Staging(size_t bytes) {
	return Grow(ctx, d_staging, staging_bytes, bytes + 64);
}
void *SegmentPool::Staging2(size_t bytes) {
	return Grow(ctx, d_staging2, staging2_bytes, bytes + 64);
}

ScanLane &SegmentPool::AcquireLane(std::unique_lock<std::mutex> &held) {
	// a lane nobody is enqueueing on, starting from a rotating position; otherwise wait for that one
	const uint32_t n = (uint32_t)lanes.size();
	const uint32_t first = next_lane.fetch_add(1) % n;
	for (uint32_t k = 0; k < n; k++) {
		ScanLane &lane = *lanes[(first + k) % n];
		held = std::unique_lock<std::mutex>(lane.lock, std::try_to_lock);
		if (held.owns_lock()) return lane;
	}
	held = std::unique_lock<std::mutex>(lanes[first]->lock);
	return *lanes[first];
}

uint64_t SegmentPool::CacheUsedBytes() {
	std::lock_guard<std::mutex> g(cache_lock);
	return (uint64_t)(cache.size() + zombies.size()) * kCacheSlotBytes;
}

void SegmentPool::CacheReclaimLocked() {
	size_t k = 0;
	for (size_t i = 0; i < zombies.size(); i++) {
		CacheEntry *z = zombies[i];
		if (z->pins == 0 && z->batch->Finished()) {
			free_slots.insert(z->slot);
			delete z;
		} else {
			zombies[k++] = z;
		}
	}
	zombies.resize(k);
}

void SegmentPool::CacheDropLocked(uint64_t key) {
	auto it = cache.find(key);
	if (it == cache.end()) return;
	CacheEntry *e = it->second;
	cache.erase(it);
	lru.erase(e->lru);
	e->dropped = true;
	if (e->pins == 0 && e->batch->Finished()) {
		free_slots.insert(e->slot);
		delete e;
	} else {
		zombies.push_back(e); // a reader still holds the block, or its copy is still in flight
	}
}

void SegmentPool::CacheDrop(uint64_t key) {
	if (!cache_capacity) return;
	std::lock_guard<std::mutex> g(cache_lock);
	CacheDropLocked(key);
}

void SegmentPool::CacheUnpin(CacheEntry *e) {
	std::lock_guard<std::mutex> g(cache_lock);
	e->pins--;
	if (e->dropped && e->pins == 0) CacheReclaimLocked();
}

CacheEntry *SegmentPool::CacheInsertLocked(uint64_t key, uint64_t version, size_t bytes) {
	if (!cache_slab || bytes > kCacheSlotBytes) return nullptr;
	if (free_slots.empty() && !zombies.empty()) CacheReclaimLocked();
	if (free_slots.empty()) {
		// least recently used entry nobody reads and no transfer writes
		for (auto it = lru.rbegin(); it != lru.rend(); ++it) {
			CacheEntry *v = *it;
			if (v->pins == 0 && v->batch->Finished()) {
				CacheDropLocked(v->key);
				break;
			}
		}
	}
	if (free_slots.empty()) return nullptr;
	CacheEntry *e = new CacheEntry();
	e->key = key;
	e->version = version;
	e->bytes = bytes;
	e->slot = *free_slots.begin();
	free_slots.erase(free_slots.begin());
	e->data = cache_slab + (size_t)e->slot * kCacheSlotBytes;
	lru.push_front(e);
	e->lru = lru.begin();
	cache[key] = e;
	return e;
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

// A transient adac_layout (analyze / pack of one batch); destroyed with the scope.
struct LayoutGuard {
	adac_layout *layout = nullptr;
	~LayoutGuard() {
		if (layout) adac_layout_destroy(layout);
	}
};

// ------------------------------------------------------------------------------------------------
// CompressionFunction tables
// ------------------------------------------------------------------------------------------------

// What init_scan hands to the engine for a segment of this codec: the pin on the segment's decoded image in the
// pool's page-locked cache (the analogue of the buffer handle FixedSizeInitScan pins,
// fixed_size_uncompressed.cpp:125-130).  Taken lazily by the first scan_vector call.
struct SuccinctScanState : public SegmentScanState {
	SegmentPool *pool = nullptr;
	CacheEntry *pin = nullptr;
	uint64_t segment_id = 0, version = 0;
	void Release() {
		if (pin) pool->CacheUnpin(pin);
		pin = nullptr;
	}
	~SuccinctScanState() override {
		Release();
	}
};

static void CodecScanPartial(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result,
                             idx_t result_offset) {
	// succinct.cpp:123-144 / fixed_size_uncompressed.cpp FixedSizeScanPartial
	auto start = segment.GetRelativeIndex(state.row_index);
	result.flat = true; // SetVectorType(FLAT_VECTOR)
	segment.ScanRows(&state, start, scan_count, result.data + result_offset * segment.type_size);
}
static void CodecScan(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result) {
	CodecScanPartial(segment, state, scan_count, result, 0); // succinct.cpp:232-240
}
static void CodecFetchRow(ColumnSegment &segment, ColumnFetchState &, row_t row_id, Vector &result, idx_t result_idx) {
	// intended semantics of SuccinctFetchRow (succinct.cpp:244-260): one value at row_id
	segment.ScanRows(nullptr, (idx_t)row_id, 1, result.data + result_idx * segment.type_size);
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
		// ColumnCheckpointState::FlushSegment: the engine takes the segment here and calls ConvertToPersistent on it;
		// the images of all flushed segments are produced together in Finalize (one device pass per pool)
		if (!checkpointer.flushed_segments.empty()) checkpointer.flushed_segments.back()->SetNext(current_segment.get());
		checkpointer.flushed_sizes.push_back(segment_size);
		checkpointer.flushed_segments.push_back(std::move(current_segment));
	}
	void Finalize(idx_t segment_size) {
		FlushSegment(segment_size);
		current_segment.reset();
		if (succinct) {
			std::vector<ColumnSegment *> segs;
			for (auto &s : checkpointer.flushed_segments) segs.push_back(s.get());
			ColumnSegment::ConvertManyToPersistent(checkpointer.db, segs, checkpointer.flushed_blocks);
		}
	}
	ColumnDataCheckpointer &checkpointer;
	std::unique_ptr<ColumnSegment> current_segment;
	bool succinct = true;
};
template <bool SUCCINCT>
static std::unique_ptr<CompressionState> InitCompression(ColumnDataCheckpointer &checkpointer,
                                                         std::unique_ptr<AnalyzeState>) {
	auto state = std::unique_ptr<SuccinctCompressState>(new SuccinctCompressState(checkpointer));
	state->succinct = SUCCINCT;
	return std::move(state);
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
// succinct.cpp:264-269); here the scan state pins the decoded image (lazily), the append state carries nothing.
static std::unique_ptr<SegmentScanState> CodecInitScan(ColumnSegment &) {
	return std::unique_ptr<SegmentScanState>(new SuccinctScanState());
}
static std::unique_ptr<CompressionAppendState> CodecInitAppend(ColumnSegment &) {
	return std::unique_ptr<CompressionAppendState>(new CompressionAppendState());
}

bool SuccinctFun::TypeIsSupported(PhysicalType type) {
	return adac_type_is_supported((int)type) != 0;
}

static CompressionFunction MakeFunction(CompressionType type, PhysicalType data_type) {
	const bool succinct = type == CompressionType::COMPRESSION_SUCCINCT;
	return CompressionFunction {type,           data_type,        FixedSizeInitAnalyze, FixedSizeAnalyze,
	                            FixedSizeFinalAnalyze, succinct ? InitCompression<true> : InitCompression<false>,
	                            Compress,        FinalizeCompress,
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

DatabaseInstance::DatabaseInstance(const std::vector<int> &devices, const DBConfig &config_p, size_t arena_bytes)
    : config(config_p), catalog(*this) {
	if (devices.empty()) throw InternalException("adacodec: a database needs at least one segment pool");
	for (size_t i = 0; i < devices.size(); i++) {
		pools.push_back(std::unique_ptr<SegmentPool>(new SegmentPool((int)i, devices[i], arena_bytes, config)));
	}
}

DatabaseInstance::~DatabaseInstance() {
	catalog.DisableBackgroundThreadCompaction(); // the policy thread uses the pools
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
    : db(db_p), segment_id(db_p.next_segment_id.fetch_add(1)), pool(db_p.PoolFor(segment_id)), type(type_p),
      type_size(adac_type_size((int)type_p)), start(start_p), function(fn), succinct_possible(succinct_possible_p),
      segment_size(segment_size_p), background_compaction_enabled(background_p) {
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
	std::lock_guard<std::mutex> g(db.chain_lock);
	if (next_hint) next_hint->prev_hint = nullptr;
	next_hint = next;
	if (next) {
		if (next->prev_hint) next->prev_hint->next_hint = nullptr;
		next->prev_hint = this;
	}
}

ColumnSegment *ColumnSegment::Next() {
	std::lock_guard<std::mutex> g(db.chain_lock);
	return next_hint;
}

ColumnSegment::~ColumnSegment() {
	// a policy round works on this pool's segments under the pool's flip_lock and re-checks the catalog under it:
	// once the segment is out of the catalog here, no round touches it any more
	std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
	db.catalog.RemoveColumnSegment(this);
	{
		std::lock_guard<std::mutex> g(db.chain_lock);
		if (prev_hint) prev_hint->next_hint = nullptr;
		if (next_hint) next_hint->prev_hint = nullptr;
	}
	pool.CacheDrop(segment_id);
	if (packed_on_device) pool.Free(word_off, arena_words);
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

void ColumnSegment::InitializeScan(ColumnScanState &state) {
	state.scan_state = function->init_scan(*this); // column_segment.cpp:133-135
}

void ColumnSegment::Scan(ColumnScanState &state, idx_t scan_count, Vector &result, idx_t result_offset,
                         bool entire_vector) {
	// column_segment.cpp:137-188
	db.catalog.AddReadAccess(this);
	if (!compacted && !background_compaction_enabled) Compact();
	// the reference re-runs init_scan under bit_compression_lock when force_reinitializing_scan_state is set; here
	// the scan state carries the representation version it pinned, and ScanRows re-pins on a mismatch
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

adac_segment_desc ColumnSegment::DeviceDesc() const {
	adac_segment_desc d;
	d.word_off = word_off;
	d.val_off = 0;
	d.min = device_min;
	d.count = (uint32_t)count;
	d.width = vec_width;
	d.flags = ADAC_SEG_PACKED;
	d.reserved = 0;
	return d;
}

// One row range of this (packed) segment as a layout-free decode job.
static adac_unpack_job MakeJob(const adac_segment_desc &d, idx_t start_row, idx_t rows, uint64_t out_off) {
	adac_unpack_job j;
	std::memset(&j, 0, sizeof j);
	j.word_off = d.word_off;
	j.min = d.min;
	j.out_off = out_off;
	j.start = (uint32_t)start_row;
	j.count = (uint32_t)rows;
	j.width = d.width;
	j.flags = d.flags;
	return j;
}

void ColumnSegment::DecodeRowsDirect(idx_t start_row, idx_t scan_count, data_ptr_t target) {
	// uncached read (bit_compression_lock held): one decode launch on the pool's main stream + a copy down into the
	// page-locked bounce block (a direct DMA), then the few KiB move into the engine's pageable vector
	std::lock_guard<std::mutex> pg(pool.lock);
	void *d_out = pool.Staging(scan_count * type_size);
	adac_unpack_job job = MakeJob(DeviceDesc(), start_row, scan_count, 0);
	Check(adac_unpack_jobs(pool.ctx, (int)type, &job, 1, pool.d_arena, d_out), "adac_unpack_jobs");
	uint8_t *bounce = pool.PinnedStaging(scan_count * type_size);
	Check(adac_memcpy_d2h(pool.ctx, bounce, d_out, scan_count * type_size), "adac_memcpy_d2h");
	std::memcpy(target, bounce, scan_count * type_size);
}

CacheEntry *ColumnSegment::PinDecoded(bool may_schedule) {
	// bit_compression_lock of this segment is held; the segment is packed on the device
	SegmentPool &p = pool;
	const size_t my_bytes = count * type_size;
	if (!p.cache_capacity || my_bytes == 0 || my_bytes > SegmentPool::kCacheSlotBytes) return nullptr;
	struct Item {
		CacheEntry *e;
		adac_unpack_job job;
	};
	std::vector<Item> items;
	std::shared_ptr<DecodeBatch> batch;
	CacheEntry *mine = nullptr;
	{
		std::lock_guard<std::mutex> chain(db.chain_lock); // the hint walk; keeps the followers from being destroyed
		std::lock_guard<std::mutex> cl(p.cache_lock);
		auto it = p.cache.find(segment_id);
		if (it != p.cache.end() && it->second->version != version) {
			p.CacheDropLocked(segment_id); // decoded from a representation that is gone
			it = p.cache.end();
		}
		bool walk = false;
		if (it != p.cache.end()) {
			mine = it->second;
			mine->pins++;
			p.lru.splice(p.lru.begin(), p.lru, mine->lru);
			p.cache_hits++;
			if (mine->trigger && may_schedule) {
				mine->trigger = false;
				walk = true;
			}
		} else {
			p.cache_misses++;
			if (!may_schedule) return nullptr;
			walk = true;
		}
		if (walk) {
			// this segment (on a miss) and the next ones of the column that live in this pool, decoded as ONE batch:
			// one launch, one copy down per run of adjacent blocks.  The first prefetched segment is the trigger for
			// the batch after this one, so a sequential scan always has about one batch in flight ahead of it.
			const uint32_t B = p.prefetch_segments;
			const uint32_t window = 2 * B * (uint32_t)db.pools.size(); // consecutive segments alternate pools
			uint32_t looked = 0;
			bool first_prefetched = true;
			for (ColumnSegment *s = this; s && looked < window && items.size() < B; s = s->next_hint, looked++) {
				if (s == this) {
					if (mine) continue;
				} else if (&s->pool != &p || s->type_size != type_size) {
					continue;
				}
				std::unique_lock<std::mutex> sl;
				if (s != this) {
					// try only: a flip in progress holds that lock and may be waiting for one of ours
					sl = std::unique_lock<std::mutex>(s->bit_compression_lock, std::try_to_lock);
					if (!sl.owns_lock()) continue;
					if (!s->function || s->function->type != CompressionType::COMPRESSION_SUCCINCT || !s->packed_on_device ||
					    s->count == 0 || s->count * s->type_size > SegmentPool::kCacheSlotBytes) {
						continue;
					}
					auto have = p.cache.find(s->segment_id);
					if (have != p.cache.end()) {
						if (have->second->version == s->version) continue;
						p.CacheDropLocked(s->segment_id);
					}
				}
				CacheEntry *e = p.CacheInsertLocked(s->segment_id, s->version, s->count * s->type_size);
				if (!e) break; // every block is pinned or in flight
				if (!batch) batch = std::make_shared<DecodeBatch>();
				e->batch = batch;
				if (s == this) {
					e->pins = 1;
					mine = e;
				} else {
					if (first_prefetched) e->trigger = true;
					first_prefetched = false;
					p.cache_prefetched++;
				}
				items.push_back(Item {e, MakeJob(s->DeviceDesc(), 0, s->count, 0)});
			}
		}
	}
	if (!items.empty()) {
		// enqueue outside the cache lock: other consumers keep hitting while this thread talks to the device
		std::sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.e->slot < b.e->slot; });
		std::vector<adac_unpack_job> jobs(items.size());
		const uint64_t slot_elems = SegmentPool::kCacheSlotBytes / type_size;
		for (size_t i = 0; i < items.size(); i++) {
			jobs[i] = items[i].job;
			jobs[i].out_off = i * slot_elems; // staging mirrors the slab: adjacent slots are adjacent in staging
		}
		std::unique_lock<std::mutex> held;
		ScanLane &lane = p.AcquireLane(held);
		adac_status st;
		if (db.config.zero_copy_decode) {
			// the decode kernel stores straight into the page-locked blocks (they are mapped into the GPU's address
			// space): the rows cross PCIe as the kernel's own 16-byte stores — no device staging, no copy engine
			for (size_t i = 0; i < items.size(); i++) jobs[i].out_off = (uint64_t)items[i].e->slot * slot_elems;
			st = adac_unpack_jobs(lane.ctx, (int)type, jobs.data(), jobs.size(), p.d_arena, p.cache_slab);
		} else {
			st = adac_unpack_jobs(lane.ctx, (int)type, jobs.data(), jobs.size(), p.d_arena, lane.d_stage);
			for (size_t i = 0; st == ADAC_OK && i < items.size();) {
				size_t j = i + 1;
				while (j < items.size() && items[j].e->slot == items[j - 1].e->slot + 1) j++;
				// whole slots of the run but the last, whose tail past its rows is not needed
				const size_t bytes = (j - 1 - i) * SegmentPool::kCacheSlotBytes + items[j - 1].e->bytes;
				st = adac_memcpy_d2h_async(lane.ctx, items[i].e->data,
				                           static_cast<uint8_t *>(lane.d_stage) + i * SegmentPool::kCacheSlotBytes, bytes);
				i = j;
			}
		}
		adac_event *ev = nullptr;
		if (st == ADAC_OK) st = adac_event_record(lane.ctx, &ev);
		held.unlock();
		batch->Publish(ev, st == ADAC_OK);
		p.cache_batches++;
		if (st != ADAC_OK) {
			std::lock_guard<std::mutex> cl(p.cache_lock);
			for (auto &it : items) {
				if (!it.e->dropped) p.CacheDropLocked(it.e->key); // a pinned one (mine) becomes a zombie until unpinned
			}
		}
	}
	if (!mine) return nullptr;
	if (!mine->batch->Wait()) {
		p.CacheUnpin(mine);
		return nullptr;
	}
	return mine;
}

void ColumnSegment::ScanRows(ColumnScanState *state, idx_t start_row, idx_t scan_count, data_ptr_t target) {
	if (start_row > count || scan_count > count - start_row) throw InternalException("scan beyond the segment");
	if (scan_count == 0) return;
	std::unique_lock<std::mutex> g(bit_compression_lock);
	if (function->type == CompressionType::COMPRESSION_SUCCINCT && packed_on_device) {
		if (pool.cache_capacity) {
			// vector-serving cache: the WHOLE segment is decoded once (together with the next few of the column, ahead
			// of the consumer); this and the following vectors are memcpys out of the block the scan state pins
			SuccinctScanState *ss = state ? dynamic_cast<SuccinctScanState *>(state->scan_state.get()) : nullptr;
			if (ss && ss->pin && (ss->segment_id != segment_id || ss->version != version)) ss->Release();
			CacheEntry *e = ss ? ss->pin : nullptr;
			if (!e) {
				e = PinDecoded(/*may_schedule=*/state != nullptr); // a one-off read (fetch_row) only uses what is there
				if (e && ss) {
					ss->pool = &pool;
					ss->pin = e;
					ss->segment_id = segment_id;
					ss->version = version;
				}
			}
			if (e) {
				g.unlock(); // the pin keeps the block; a flip from here on is seen by the next call's version check
				std::memcpy(target, e->data + start_row * type_size, scan_count * type_size);
				if (!ss) pool.CacheUnpin(e);
				return;
			}
		}
		DecodeRowsDirect(start_row, scan_count, target);
	} else {
		// unpacked slots / uncompressed block: the bytes ARE the values (no min add: SURVEY.md §8a (iii))
		std::memcpy(target, raw.data() + start_row * type_size, scan_count * type_size);
	}
}

idx_t ColumnSegment::AppendRows(UnifiedVectorFormat &data, idx_t offset, idx_t append_count) {
	// SuccinctAppend (succinct.cpp:308-322) / FixedSizeAppend: no arithmetic here — the running min/max of
	// SuccinctAppendLoop is produced by adac_analyze when the segment compacts.
	idx_t max_tuple_count = segment_size / type_size;
	idx_t copy_count = std::min<idx_t>(append_count, max_tuple_count - count);
	const bool track_validity = function->type == CompressionType::COMPRESSION_SUCCINCT;
	const uint64_t null_bits = NullBits(type, type_size);
	if (track_validity && validity.empty()) validity.assign((max_tuple_count + 63) / 64 + 1, ~0ull);
	if (!data.sel && !data.validity) {
		std::memcpy(raw.data() + count * type_size, data.data + offset * type_size, copy_count * type_size);
	} else {
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
	}
	count += copy_count;
	return copy_count;
}

idx_t ColumnSegment::Append(UnifiedVectorFormat &append_data, idx_t offset, idx_t append_count) {
	// column_segment.cpp:247-271.  The whole append is one flip-free section: the background policy thread must
	// not compact the segment between the Uncompact below and the write into the unpacked image (in the
	// reference that window is open: its TSan suppressions cover it)
	std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
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
	if (compacted || !function || num_elements == 0 || !succinct_possible) return false;
	// a segment the arena had no room for is not tried again (by every scan call, in the non-adaptive mode) until
	// some block of its pool has been freed
	return !(arena_refused && refused_generation == pool.free_generation.load());
}

void ColumnSegment::Compact() {
	std::vector<ColumnSegment *> one {this};
	CompactMany(db, one);
}

// Frees the arena blocks a batch took if the batch does not reach the point where its segments own them.
struct ArenaRollback {
	SegmentPool &pool;
	std::vector<std::pair<uint64_t, uint64_t>> blocks; // word_off, words
	bool armed = true;
	explicit ArenaRollback(SegmentPool &p) : pool(p) {
	}
	~ArenaRollback() {
		if (armed) {
			for (auto &b : blocks) pool.Free(b.first, b.second);
		}
	}
};

void ColumnSegment::CompactMany(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments) {
	// Batched ColumnSegment::Compact (column_segment.cpp:273-322): per (pool, type, rule) group one upload, one
	// adac_analyze, the width decision on the host from the downloaded min/max, one adac_pack.
	std::map<std::tuple<int, uint8_t, int>, std::vector<ColumnSegment *>> groups;
	for (auto *s : segments) {
		int rule = s->function->type == CompressionType::COMPRESSION_SUCCINCT ? ADAC_RULE_APPEND : ADAC_RULE_RECOMPACT;
		groups[std::make_tuple(s->pool.index, (uint8_t)s->type, rule)].push_back(s);
	}
	for (auto &g : groups) {
		SegmentPool &pool = *db.pools[std::get<0>(g.first)];
		std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
		std::vector<ColumnSegment *> todo;
		for (auto *s : g.second) {
			// decided under the flip lock; the rule is re-read there too (an Uncompact may have come in between)
			const int rule = s->function->type == CompressionType::COMPRESSION_SUCCINCT ? ADAC_RULE_APPEND : ADAC_RULE_RECOMPACT;
			if (s->NeedsCompaction() && rule == std::get<2>(g.first)) todo.push_back(s);
		}
		if (!todo.empty()) CompactGroup(db, pool, std::get<1>(g.first), std::get<2>(g.first), todo);
	}
}

void ColumnSegment::CompactGroup(DatabaseInstance &db, SegmentPool &pool, int ptype, int rule,
                                 const std::vector<ColumnSegment *> &segs) {
	// pool.flip_lock is held by the caller
	PhaseTrace trace;
	const bool padded = db.config.succinct_padded_to_next_byte_enabled;
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
	if (any_null) {
		vmask.assign(span / 64 + 2, ~0ull);
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
	std::vector<adac_segment_desc> descs; // of the segments that get packed, in order
	std::vector<size_t> pidx;             // their indices in segs
	std::vector<uint8_t> widths(segs.size());
	std::vector<char> skipped(segs.size(), 0); // no arena space: stays as it is (still needs compaction)
	ArenaRollback rollback(pool);
	{ // device work under the pool lock; representation flips (bit_compression_lock) after it is released
		std::lock_guard<std::mutex> pg(pool.lock);
		adac_ctx *ctx = pool.ctx;
		trace.mark("prep");
		// gather the segments' rows into page-locked staging (a few host threads for big batches: the gather,
		// not PCIe, bounds a re-compaction round) and upload them with one copy at PCIe rate
		uint8_t *host = pool.PinnedStaging(host_bytes);
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
		void *d_vals = pool.Staging(host_bytes);
		trace.mark("dev_alloc");
		Check(adac_memcpy_h2d(ctx, d_vals, host, host_bytes), "upload rows");
		trace.mark("h2d");
		uint64_t *d_valid = nullptr;
		if (any_null) {
			d_valid = static_cast<uint64_t *>(pool.Staging2(vmask.size() * 8));
			Check(adac_memcpy_h2d(ctx, d_valid, vmask.data(), vmask.size() * 8), "upload validity");
		}
		{
			LayoutGuard probe;
			Check(adac_layout_create(ctx, ptype, counts.data(), offs.data(), segs.size(), &probe.layout), "adac_layout_create");
			Check(adac_analyze(probe.layout, d_vals, d_valid, rule), "adac_analyze");
			Check(adac_layout_get_minmax(probe.layout, mm.data()), "adac_layout_get_minmax");
		}
		trace.mark("analyze");
		// width decision (column_segment.cpp:351-363 / :404-420) and arena placement
		std::vector<uint32_t> pcounts;
		std::vector<uint64_t> poffs;
		for (size_t i = 0; i < segs.size(); i++) {
			uint8_t w = adac_width(mm[2 * i], mm[2 * i + 1], rule, padded);
			widths[i] = w;
			if (8 * ts > w) {
				adac_segment_desc d;
				const uint64_t need = adac_arena_words(counts[i], w);
				if (!pool.TryAllocate(need, d.word_off)) {
					// the arena is full: the segment keeps its unpacked form; it is tried again once space was freed
					skipped[i] = 1;
					pool.exhausted_events++;
					segs[i]->arena_refused = true;
					segs[i]->refused_generation = pool.free_generation.load();
					continue;
				}
				rollback.blocks.emplace_back(d.word_off, need);
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
			LayoutGuard packer;
			Check(adac_layout_create(ctx, ptype, pcounts.data(), poffs.data(), descs.size(), &packer.layout),
			      "adac_layout_create");
			Check(adac_layout_set_descs(packer.layout, descs.data()), "adac_layout_set_descs");
			trace.mark("alloc+layout");
			Check(adac_pack(packer.layout, d_vals, d_valid, pool.d_arena), "adac_pack");
			Check(adac_ctx_sync(ctx), "adac_ctx_sync");
			trace.mark("pack");
		}
	} // pool lock released
	rollback.armed = false; // from here on the segments own their blocks
	size_t p = 0;
	// the unpacked images of a big batch go back to the OS off the critical path: unmapping a 256 KiB block
	// costs ~18 us, 6.5 ms for the 360 segments of a first policy round
	std::vector<std::vector<uint8_t>> graveyard;
	if (segs.size() > 8) graveyard.reserve(segs.size());
	for (size_t i = 0; i < segs.size(); i++) {
		if (skipped[i]) continue;
		bool packed = p < pidx.size() && pidx[p] == i;
		segs[i]->FinishCompaction(packed, widths[i], mm[2 * i], mm[2 * i + 1], rule, packed ? descs[p].word_off : 0,
		                          packed ? descs[p].min : UINT64_MAX, segs.size() > 8 ? &graveyard : nullptr);
		if (packed) p++;
	}
	if (!graveyard.empty()) {
		std::thread([g = std::move(graveyard)]() mutable { g.clear(); }).detach();
	}
	trace.mark("finish");
}

void ColumnSegment::FinishCompaction(bool packed, uint8_t width, uint64_t mn, uint64_t mx, int rule, uint64_t off,
                                     uint64_t stored_min, std::vector<std::vector<uint8_t>> *graveyard) {
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
		device_min = stored_min;
		arena_refused = false;
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
	version++;
	db.data_size += (int64_t)GetDataSize() - (int64_t)before;
}

void ColumnSegment::Uncompact() {
	// column_segment.cpp:324-346 + UncompressSuccinct :458-506
	std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
	if (!compacted || !function || function->type != CompressionType::COMPRESSION_SUCCINCT) return;
	std::lock_guard<std::mutex> g(bit_compression_lock);
	const idx_t compressed_size = adac_size_in_bytes(vec_slots, vec_width);
	if (packed_on_device) {
		raw.assign(segment_size, 0);
		if (count) {
			std::lock_guard<std::mutex> pg(pool.lock);
			void *d_out = pool.Staging(count * type_size);
			adac_unpack_job job = MakeJob(DeviceDesc(), 0, count, 0);
			Check(adac_unpack_jobs(pool.ctx, (int)type, &job, 1, pool.d_arena, d_out), "adac_unpack_jobs");
			uint8_t *bounce = pool.PinnedStaging(count * type_size);
			Check(adac_memcpy_d2h(pool.ctx, bounce, d_out, count * type_size), "adac_memcpy_d2h");
			std::memcpy(raw.data(), bounce, count * type_size);
		}
		pool.Free(word_off, arena_words);
		pool.CacheDrop(segment_id);
		packed_on_device = false;
	}
	function = db.GetCompressionFunction(CompressionType::COMPRESSION_UNCOMPRESSED, type);
	compacted = false;
	vec_slots = 0; // succinct_vec.resize(0)
	version++;     // force_reinitializing_scan_state = true: scan states re-pin on their next call
	segment_type = ColumnSegmentType::TRANSIENT;
	db.data_size += (int64_t)segment_size - (int64_t)compressed_size;
}

// ------------------------------------------------------------------------------------------------
// Persistence
// ------------------------------------------------------------------------------------------------

void ColumnSegment::ConvertToPersistent(std::vector<uint8_t> &image) {
	std::vector<ColumnSegment *> one {this};
	std::vector<std::vector<uint8_t>> images;
	ConvertManyToPersistent(db, one, images);
	image = std::move(images[0]);
}

void ColumnSegment::ConvertManyToPersistent(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments,
                                            std::vector<std::vector<uint8_t>> &images) {
	images.assign(segments.size(), std::vector<uint8_t>());
	CompactMany(db, segments); // a segment that was never scanned or filled is still in its append form
	std::map<std::pair<int, uint8_t>, std::vector<size_t>> groups; // (pool, type) -> indices into segments
	for (size_t i = 0; i < segments.size(); i++) {
		groups[{segments[i]->pool.index, (uint8_t)segments[i]->type}].push_back(i);
	}
	for (auto &g : groups) {
		SegmentPool &pool = *db.pools[g.first.first];
		const int ptype = g.first.second;
		std::lock_guard<std::recursive_mutex> flips(pool.flip_lock); // no flip while the images are taken
		std::vector<adac_segment_desc> descs;
		std::vector<uint64_t> offs;
		std::vector<size_t> on_device;
		uint64_t total = 0;
		for (size_t i : g.second) {
			ColumnSegment &s = *segments[i];
			std::lock_guard<std::mutex> sl(s.bit_compression_lock);
			const bool empty = s.count == 0; // e.g. the last segment FinalizeCompress flushes: nothing to compact
			if (!empty && (!s.compacted || s.function->type != CompressionType::COMPRESSION_SUCCINCT)) {
				throw InternalException("ConvertToPersistent: the segment is not in its succinct form");
			}
			if (!empty && s.packed_on_device) {
				descs.push_back(s.DeviceDesc());
				offs.push_back(total);
				total += adac_block_stride(s.count, s.vec_width);
				on_device.push_back(i);
			} else {
				// slots at the type's own width (nothing to gain from packing): the image is built from the host
				// block — bytes only, no arithmetic
				adac_segment_desc d;
				std::memset(&d, 0, sizeof d);
				d.min = s.min_factor;
				d.count = (uint32_t)s.count;
				d.width = (uint8_t)(8 * s.type_size);
				d.flags = 0;
				std::vector<uint64_t> words(adac_packed_words(s.count, d.width) + 1, 0);
				std::memcpy(words.data(), s.raw.data(), s.count * s.type_size);
				images[i].resize(adac_block_bytes(s.count, d.width));
				if (adac_block_write(&d, ptype, words.data(), images[i].data(), images[i].size()) != images[i].size()) {
					throw InternalException("adac_block_write failed");
				}
			}
			s.segment_type = ColumnSegmentType::PERSISTENT;
		}
		if (on_device.empty()) continue;
		std::lock_guard<std::mutex> pg(pool.lock);
		void *d_blocks = pool.Staging(total);
		Check(adac_blocks_write(pool.ctx, ptype, descs.data(), offs.data(), descs.size(), pool.d_arena, d_blocks),
		      "adac_blocks_write");
		uint8_t *host = pool.PinnedStaging(total);
		Check(adac_memcpy_d2h(pool.ctx, host, d_blocks, total), "adac_memcpy_d2h(blocks)");
		for (size_t k = 0; k < on_device.size(); k++) {
			const size_t bytes = adac_block_bytes(descs[k].count, descs[k].width);
			images[on_device[k]].assign(host + offs[k], host + offs[k] + bytes);
		}
	}
}

std::vector<std::unique_ptr<ColumnSegment>>
ColumnSegment::CreatePersistentSegments(DatabaseInstance &db,
                                        const std::vector<std::pair<const uint8_t *, size_t>> &images,
                                        const std::vector<idx_t> &starts) {
	if (images.size() != starts.size()) throw InternalException("CreatePersistentSegments: one start row per image");
	std::vector<std::unique_ptr<ColumnSegment>> out(images.size());
	std::vector<adac_segment_desc> headers(images.size());
	std::vector<int> types(images.size());
	for (size_t i = 0; i < images.size(); i++) {
		Check(adac_block_peek(images[i].first, images[i].second, &headers[i], &types[i]), "adac_block_peek");
		const PhysicalType type = (PhysicalType)types[i];
		const idx_t ts = adac_type_size(types[i]);
		// a segment of the loaded size: Storage::BLOCK_SIZE unless the image holds more rows than that
		const idx_t seg_size = std::max<idx_t>(BLOCK_SIZE, (idx_t)headers[i].count * ts);
		auto fn = db.GetCompressionFunction(CompressionType::COMPRESSION_SUCCINCT, type);
		out[i].reset(new ColumnSegment(db, type, starts[i], seg_size, fn, true,
		                               db.config.adaptive_succinct_compression_enabled));
	}
	// per (pool, type): place the packed ones in the arena, one upload of their images, one device pass
	std::map<std::pair<int, uint8_t>, std::vector<size_t>> groups;
	for (size_t i = 0; i < images.size(); i++) groups[{out[i]->pool.index, (uint8_t)out[i]->type}].push_back(i);
	for (auto &g : groups) {
		SegmentPool &pool = *db.pools[g.first.first];
		const int ptype = g.first.second;
		std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
		std::vector<adac_segment_desc> descs;
		std::vector<uint64_t> offs;
		std::vector<size_t> idx;
		uint64_t total = 0;
		ArenaRollback rollback(pool);
		for (size_t i : g.second) {
			const adac_segment_desc &h = headers[i];
			if (!(h.flags & ADAC_SEG_PACKED)) continue;
			adac_segment_desc d = h;
			const uint64_t need = adac_arena_words(h.count, h.width);
			if (!pool.TryAllocate(need, d.word_off)) throw InternalException("adacodec: segment pool arena exhausted");
			rollback.blocks.emplace_back(d.word_off, need);
			descs.push_back(d);
			offs.push_back(total);
			total += adac_block_stride(h.count, h.width);
			idx.push_back(i);
		}
		if (!descs.empty()) {
			std::lock_guard<std::mutex> pg(pool.lock);
			uint8_t *host = pool.PinnedStaging(total);
			for (size_t k = 0; k < idx.size(); k++) {
				const size_t stride = adac_block_stride(descs[k].count, descs[k].width);
				const size_t have = std::min(images[idx[k]].second, stride);
				std::memcpy(host + offs[k], images[idx[k]].first, have);
				std::memset(host + offs[k] + have, 0, stride - have);
			}
			void *d_blocks = pool.Staging(total);
			Check(adac_memcpy_h2d(pool.ctx, d_blocks, host, total), "adac_memcpy_h2d(blocks)");
			Check(adac_blocks_read(pool.ctx, ptype, descs.data(), offs.data(), descs.size(), d_blocks, pool.d_arena),
			      "adac_blocks_read");
		}
		rollback.armed = false;
		size_t k = 0;
		for (size_t i : g.second) {
			ColumnSegment &s = *out[i];
			const adac_segment_desc &h = headers[i];
			std::lock_guard<std::mutex> sl(s.bit_compression_lock);
			const idx_t before = s.GetDataSize();
			s.count = h.count;
			s.num_elements = h.count;
			s.segment_type = ColumnSegmentType::PERSISTENT;
			s.appended_via_succinct = true;
			s.compacted = true;
			s.vec_width = h.width;
			if (h.flags & ADAC_SEG_PACKED) {
				s.vec_slots = h.count;
				s.packed_on_device = true;
				s.word_off = descs[k].word_off;
				s.arena_words = adac_arena_words(h.count, h.width);
				s.device_min = h.min;
				s.min_factor = h.min;
				// the image does not carry max_factor; the tightest bound the width allows stands in for it
				s.max_factor = h.width >= 64 ? UINT64_MAX : h.min + ((1ull << h.width) - 1ull);
				std::vector<uint8_t>().swap(s.raw);
				k++;
			} else {
				// unpacked slots: the bytes of the image's words are the values
				s.vec_slots = s.segment_size / s.type_size;
				s.min_factor = h.min;
				std::memcpy(s.raw.data(), images[i].first + 9, (size_t)h.count * s.type_size);
			}
			s.version++;
			db.data_size += (int64_t)s.GetDataSize() - (int64_t)before;
		}
	}
	for (size_t i = 0; i + 1 < out.size(); i++) {
		if (out[i]->type == out[i + 1]->type && out[i + 1]->start == out[i]->start + out[i]->count) {
			out[i]->SetNext(out[i + 1].get());
		}
	}
	return out;
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
	segments.insert(segment);
}

void ColumnSegmentCatalog::RemoveColumnSegment(ColumnSegment *segment) {
	std::lock_guard<std::mutex> g(lock);
	segments.erase(segment);
}

void ColumnSegmentCatalog::AddReadAccess(ColumnSegment *segment) {
	// column_segment_catalog.cpp:37-54
	if (segment == nullptr || !segment->is_data_segment) return;
	segment->num_reads.fetch_add(1, std::memory_order_relaxed);
	event_counter.fetch_add(1, std::memory_order_relaxed);
}

idx_t ColumnSegmentCatalog::NumSegments() {
	std::lock_guard<std::mutex> g(lock);
	return segments.size();
}

std::string ColumnSegmentCatalog::LastBackgroundError() {
	std::lock_guard<std::mutex> g(lock);
	return last_background_error;
}

// Runs f(pool index) for every pool that has work, one host thread per pool when there are several (each pool is
// its own GPU and stream); the first exception is rethrown after all of them have finished.
template <typename F>
static void ForEachPool(const std::vector<char> &has_work, F &&f) {
	std::vector<size_t> todo;
	for (size_t p = 0; p < has_work.size(); p++) {
		if (has_work[p]) todo.push_back(p);
	}
	if (todo.size() <= 1) {
		for (size_t p : todo) f(p);
		return;
	}
	std::vector<std::thread> th;
	std::vector<std::string> errors(todo.size());
	for (size_t k = 0; k < todo.size(); k++) {
		th.emplace_back([&, k]() {
			try {
				f(todo[k]);
			} catch (const std::exception &e) {
				errors[k] = e.what()[0] ? e.what() : "error";
			}
		});
	}
	for (auto &t : th) t.join();
	for (auto &e : errors) {
		if (!e.empty()) throw InternalException(e);
	}
}

void ColumnSegmentCatalog::CompactAllSegments() {
	// column_segment_catalog.cpp:56-62.  Pool by pool under the pool's flip lock, so that a segment destroyed
	// meanwhile (its destructor takes the same lock before it leaves the catalog) is never touched
	const size_t npools = db.pools.size();
	ForEachPool(std::vector<char>(npools, 1), [&](size_t p) {
		SegmentPool &pool = *db.pools[p];
		std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
		std::vector<ColumnSegment *> mine;
		{
			std::lock_guard<std::mutex> g(lock);
			for (auto *s : segments) {
				if (&s->pool == &pool) mine.push_back(s);
			}
		}
		ColumnSegment::CompactMany(db, mine);
	});
}

size_t ColumnSegmentCatalog::GetTotalDataSize() {
	std::lock_guard<std::mutex> g(lock);
	size_t data_size = 0;
	for (auto *s : segments) data_size += s->GetDataSize();
	return data_size;
}

void ColumnSegmentCatalog::CompressLowestKSegmentsOnce(double compression_rate) {
	// column_segment_catalog.cpp:79-112.  The reference sorts by num_reads only (ties in unordered_map
	// order); ties are broken here by the segment's start row, then id, to be deterministic.
	struct Snap {
		ColumnSegment *seg;
		idx_t num_reads, start;
		uint64_t id;
		int pool;
	};
	std::vector<Snap> v;
	{
		std::lock_guard<std::mutex> g(lock); // a registered segment is alive: everything needed is copied here
		v.reserve(segments.size());
		for (auto *s : segments) {
			v.push_back(Snap {s, s->num_reads.load(std::memory_order_relaxed), s->start, s->segment_id, s->pool.index});
		}
	}
	std::sort(v.begin(), v.end(), [](const Snap &l, const Snap &r) {
		if (l.num_reads != r.num_reads) return l.num_reads < r.num_reads;
		if (l.start != r.start) return l.start < r.start;
		return l.id < r.id;
	});
	const size_t npools = db.pools.size();
	std::vector<std::vector<Snap>> to_compact(npools), to_uncompact(npools);
	std::vector<char> has_work(npools, 0);
	float cum_sum = 0;
	idx_t curr_counter = v.size();
	for (auto &e : v) {
		cum_sum += 1;
		if (cum_sum / curr_counter < compression_rate) { // float / idx_t against a double, as the reference compares
			to_compact[e.pool].push_back(e);
		} else {
			to_uncompact[e.pool].push_back(e);
		}
		has_work[e.pool] = 1;
	}
	ForEachPool(has_work, [&](size_t p) {
		SegmentPool &pool = *db.pools[p];
		// a segment's destructor takes this lock before it leaves the catalog: what is still registered once the
		// lock is held stays alive until it is released.  The id is compared too: the address of a destroyed segment
		// may already belong to a new one, possibly of another pool, whose destructor this lock would not hold off
		std::lock_guard<std::recursive_mutex> flips(pool.flip_lock);
		std::vector<ColumnSegment *> compact, uncompact;
		{
			std::lock_guard<std::mutex> g(lock);
			for (auto &e : to_compact[p]) {
				if (segments.count(e.seg) && e.seg->segment_id == e.id) compact.push_back(e.seg);
			}
			for (auto &e : to_uncompact[p]) {
				if (segments.count(e.seg) && e.seg->segment_id == e.id) uncompact.push_back(e.seg);
			}
			for (auto *s : compact) s->num_reads.store(0, std::memory_order_relaxed);
			for (auto *s : uncompact) s->num_reads.store(0, std::memory_order_relaxed);
		}
		ColumnSegment::CompactMany(db, compact);
		for (auto *s : uncompact) s->Uncompact();
	});
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
			// nothing may escape the thread (std::terminate would take the host process down): a failed round —
			// a HIP error, an allocation failure — is recorded and the next round tries again
			try {
				CompressLowestKSegmentsOnce(0.90);
			} catch (const std::exception &e) {
				background_errors++;
				std::lock_guard<std::mutex> g(lock);
				last_background_error = e.what();
			} catch (...) {
				background_errors++;
				std::lock_guard<std::mutex> g(lock);
				last_background_error = "unknown exception";
			}
			background_rounds++;
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

extern "C" adach_db *adach_db_create_pools(const int *devices, int npools, int succinct_enabled, int adaptive, int padded,
                                           uint64_t arena_bytes, uint64_t decoded_cache_bytes, uint32_t scan_lanes,
                                           uint32_t prefetch_segments) {
	adach_db *h = nullptr;
	Guard([&]() {
		if (!devices || npools < 1) throw InternalException("adach_db_create_pools: at least one pool");
		DBConfig cfg;
		cfg.succinct_enabled = succinct_enabled != 0;
		cfg.adaptive_succinct_compression_enabled = adaptive != 0;
		cfg.succinct_padded_to_next_byte_enabled = padded != 0;
		cfg.decoded_cache_bytes = decoded_cache_bytes;
		if (scan_lanes) cfg.scan_lanes = scan_lanes;
		if (prefetch_segments) cfg.prefetch_segments = prefetch_segments;
		if (const char *z = std::getenv("ADACH_ZERO_COPY")) cfg.zero_copy_decode = z[0] != '0'; // A/B knob
		std::vector<int> devs(devices, devices + npools);
		auto db = std::unique_ptr<DatabaseInstance>(new DatabaseInstance(devs, cfg, arena_bytes));
		h = new adach_db {std::move(db)};
	});
	return h;
}

extern "C" adach_db *adach_db_create(int device, int succinct_enabled, int adaptive, int padded, uint64_t arena_bytes) {
	return adach_db_create_pools(&device, 1, succinct_enabled, adaptive, padded, arena_bytes, 0, 0, 0);
}

extern "C" adach_db *adach_db_create_cached(int device, int succinct_enabled, int adaptive, int padded,
                                            uint64_t arena_bytes, uint64_t decoded_cache_bytes) {
	return adach_db_create_pools(&device, 1, succinct_enabled, adaptive, padded, arena_bytes, decoded_cache_bytes, 0, 0);
}

extern "C" uint64_t adach_db_num_pools(adach_db *h) {
	return h->db->pools.size();
}

extern "C" void adach_db_cache_stats(adach_db *h, uint64_t *hits, uint64_t *misses, uint64_t *bytes) {
	uint64_t a = 0, b = 0, c = 0;
	for (auto &p : h->db->pools) {
		a += p->cache_hits.load();
		b += p->cache_misses.load();
		c += p->CacheUsedBytes();
	}
	if (hits) *hits = a;
	if (misses) *misses = b;
	if (bytes) *bytes = c;
}

extern "C" void adach_db_prefetch_stats(adach_db *h, uint64_t *batches, uint64_t *prefetched, uint64_t *exhausted) {
	uint64_t a = 0, b = 0, c = 0;
	for (auto &p : h->db->pools) {
		a += p->cache_batches.load();
		b += p->cache_prefetched.load();
		c += p->exhausted_events.load();
	}
	if (batches) *batches = a;
	if (prefetched) *prefetched = b;
	if (exhausted) *exhausted = c;
}

// Full scan of a chain of segments in the engine's call pattern — ColumnData::ScanVector (column_data.cpp:92-139):
// InitializeScan on a segment, ColumnSegment::Scan on vector_size-row vectors, on to `next` when a segment is
// exhausted — timed on the host; checksum = wrapping sum of all rows.
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

static void ScanRange(adach_segment **segs, uint64_t lo, uint64_t hi, uint64_t vector_size, uint64_t &sum,
                      uint64_t &rows) {
	std::vector<uint8_t> vec(vector_size * 8 + 64);
	ColumnScanState st;
	for (uint64_t i = lo; i < hi; i++) {
		ColumnSegment &s = *segs[i]->seg;
		st.current = &s;
		s.InitializeScan(st); // state.current->InitializeScan(state) when ScanVector moves to the next segment
		for (idx_t r = 0; r < s.count; r += vector_size) {
			idx_t c = std::min<idx_t>(vector_size, s.count - r);
			st.row_index = s.start + r;
			Vector v;
			v.data = vec.data();
			s.Scan(st, c, v, 0, true);
			sum += SumVector(vec.data(), c, s.type_size); // the consumer touches every value
			rows += c;
		}
	}
}

extern "C" int adach_full_scan_mt(adach_segment **segs, uint64_t nseg, uint64_t vector_size, uint32_t threads,
                                  uint64_t *checksum, double *seconds, uint64_t *rows_out) {
	return Guard([&]() {
		if (threads < 1) threads = 1;
		if (threads > nseg) threads = (uint32_t)std::max<uint64_t>(1, nseg);
		std::vector<uint64_t> sums(threads, 0), rows(threads, 0);
		std::vector<std::string> errors(threads);
		auto t0 = std::chrono::steady_clock::now();
		if (threads == 1) {
			ScanRange(segs, 0, nseg, vector_size, sums[0], rows[0]);
		} else {
			// one contiguous run of segments per thread: the row-group morsels of a parallel table scan
			// (src/storage/table/row_group_collection.cpp:119-155)
			std::vector<std::thread> th;
			for (uint32_t t = 0; t < threads; t++) {
				th.emplace_back([&, t]() {
					try {
						ScanRange(segs, nseg * t / threads, nseg * (t + 1) / threads, vector_size, sums[t], rows[t]);
					} catch (const std::exception &e) {
						errors[t] = e.what()[0] ? e.what() : "error";
					}
				});
			}
			for (auto &t : th) t.join();
			for (auto &e : errors) {
				if (!e.empty()) throw InternalException(e);
			}
		}
		auto t1 = std::chrono::steady_clock::now();
		uint64_t sum = 0, nrows = 0;
		for (uint32_t t = 0; t < threads; t++) {
			sum += sums[t];
			nrows += rows[t];
		}
		if (checksum) *checksum = sum;
		if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
		if (rows_out) *rows_out = nrows;
	});
}

extern "C" int adach_full_scan(adach_segment **segs, uint64_t nseg, uint64_t vector_size, uint64_t *checksum,
                               double *seconds, uint64_t *rows_out) {
	return adach_full_scan_mt(segs, nseg, vector_size, 1, checksum, seconds, rows_out);
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

extern "C" int adach_checkpoint_column(adach_db *h, int compression_type, int physical_type, uint64_t row_group_start,
                                       const void *values, const uint64_t *validity, uint64_t n,
                                       adach_segment **out_segments, uint64_t max_segments, uint64_t *out_nseg,
                                       uint64_t *out_sizes, uint64_t *out_score, void *out_blocks, uint64_t blocks_cap,
                                       uint64_t *out_block_offs) {
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
		if (out_blocks || out_block_offs) {
			// the block images ConvertToPersistent produced, back to back (8-byte aligned), offsets [nseg + 1]
			uint64_t off = 0;
			for (size_t i = 0; i < checkpointer.flushed_blocks.size(); i++) {
				const auto &b = checkpointer.flushed_blocks[i];
				if (out_block_offs) out_block_offs[i] = off;
				if (out_blocks) {
					if (off + b.size() > blocks_cap) throw InternalException("block buffer too small");
					std::memcpy(static_cast<uint8_t *>(out_blocks) + off, b.data(), b.size());
				}
				off += (b.size() + 7) & ~size_t(7);
			}
			if (out_block_offs) out_block_offs[checkpointer.flushed_blocks.size()] = off;
		}
		for (size_t i = 0; i < checkpointer.flushed_segments.size(); i++) {
			out_segments[i] = new adach_segment {std::move(checkpointer.flushed_segments[i])};
			if (out_sizes) out_sizes[i] = checkpointer.flushed_sizes[i];
		}
	});
}

extern "C" int adach_compress_column(adach_db *h, int compression_type, int physical_type, uint64_t row_group_start,
                                     const void *values, const uint64_t *validity, uint64_t n,
                                     adach_segment **out_segments, uint64_t max_segments, uint64_t *out_nseg,
                                     uint64_t *out_sizes, uint64_t *out_score) {
	return adach_checkpoint_column(h, compression_type, physical_type, row_group_start, values, validity, n, out_segments,
	                               max_segments, out_nseg, out_sizes, out_score, nullptr, 0, nullptr);
}

extern "C" uint64_t adach_segment_block_bytes(adach_segment *s) {
	ColumnSegment &seg = *s->seg;
	const uint8_t w = seg.IsBitCompressed() ? seg.Width() : (uint8_t)(8 * seg.type_size);
	return adac_block_bytes(seg.count, w); // an upper bound before the segment has been compacted
}

extern "C" int adach_segments_persist(adach_db *h, adach_segment **segs, uint64_t nseg, void *out, uint64_t cap,
                                      uint64_t *out_offs) {
	return Guard([&]() {
		std::vector<ColumnSegment *> v;
		for (uint64_t i = 0; i < nseg; i++) v.push_back(segs[i]->seg.get());
		std::vector<std::vector<uint8_t>> images;
		ColumnSegment::ConvertManyToPersistent(*h->db, v, images);
		uint64_t off = 0;
		for (uint64_t i = 0; i < nseg; i++) {
			out_offs[i] = off;
			if (off + images[i].size() > cap) throw InternalException("block buffer too small");
			std::memcpy(static_cast<uint8_t *>(out) + off, images[i].data(), images[i].size());
			off += (images[i].size() + 7) & ~size_t(7);
		}
		out_offs[nseg] = off;
	});
}

extern "C" int adach_segments_load(adach_db *h, const void *blocks, const uint64_t *offs, const uint64_t *lens,
                                   const uint64_t *starts, uint64_t nseg, adach_segment **out_segments) {
	return Guard([&]() {
		std::vector<std::pair<const uint8_t *, size_t>> images;
		std::vector<idx_t> st;
		for (uint64_t i = 0; i < nseg; i++) {
			images.emplace_back(static_cast<const uint8_t *>(blocks) + offs[i], (size_t)lens[i]);
			st.push_back(starts[i]);
		}
		auto segs = ColumnSegment::CreatePersistentSegments(*h->db, images, st);
		for (uint64_t i = 0; i < nseg; i++) out_segments[i] = new adach_segment {std::move(segs[i])};
	});
}

extern "C" int adach_db_reserve_staging(adach_db *h, uint64_t bytes) {
	return Guard([&]() {
		// page-locking the upload buffer is the one slow step of a first compaction round (hipHostMalloc: ~20 ms
		// per 100 MB): an engine sizes it once, here, instead of inside the first policy step
		for (auto &p : h->db->pools) {
			std::lock_guard<std::mutex> pg(p->lock);
			p->PinnedStaging(bytes);
			p->Staging(bytes);
		}
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
		st.current = s->seg.get();
		s->seg->InitializeScan(st);
		Vector v;
		v.data = static_cast<data_ptr_t>(result);
		s->seg->Scan(st, count, v, result_offset, entire_vector != 0);
	});
}

// The scan state of ONE scanning thread as the engine keeps it (ColumnScanState, scan_state.hpp): init_scan's result
// lives in it from one scan_vector call to the next, which is what lets it pin the decoded block of the segment.
struct adach_scan_state {
	ColumnScanState st;
};

extern "C" adach_scan_state *adach_scan_state_create(void) {
	return new (std::nothrow) adach_scan_state();
}

extern "C" void adach_scan_state_destroy(adach_scan_state *st) {
	delete st; // releases the pin
}

extern "C" int adach_segment_init_scan(adach_segment *s, adach_scan_state *st) {
	return Guard([&]() {
		st->st.current = s->seg.get();
		s->seg->InitializeScan(st->st); // replaces (and thereby unpins) what the state held for the previous segment
	});
}

extern "C" int adach_segment_scan_with(adach_segment *s, adach_scan_state *st, uint64_t row_index, uint64_t count,
                                       void *result, uint64_t result_offset, int entire_vector) {
	return Guard([&]() {
		if (st->st.current != s->seg.get() || !st->st.scan_state) throw InternalException("scan state of another segment");
		st->st.row_index = row_index;
		Vector v;
		v.data = static_cast<data_ptr_t>(result);
		s->seg->Scan(st->st, count, v, result_offset, entire_vector != 0);
	});
}

extern "C" int adach_segments_compact(adach_db *h, adach_segment **segs, uint64_t nseg) {
	return Guard([&]() {
		std::vector<ColumnSegment *> list;
		for (uint64_t i = 0; i < nseg; i++) list.push_back(segs[i]->seg.get());
		ColumnSegment::CompactMany(*h->db, list);
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
extern "C" int adach_segment_pool(adach_segment *s) {
	return s->seg->pool.index;
}
extern "C" int adach_segment_persistent(adach_segment *s) {
	return s->seg->segment_type == ColumnSegmentType::PERSISTENT;
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
extern "C" void adach_catalog_background_stats(adach_db *h, uint64_t *rounds, uint64_t *errors, char *last_error,
                                               uint64_t cap) {
	if (rounds) *rounds = h->db->catalog.BackgroundRounds();
	if (errors) *errors = h->db->catalog.BackgroundErrors();
	if (last_error && cap) {
		const std::string e = h->db->catalog.LastBackgroundError();
		std::snprintf(last_error, cap, "%s", e.c_str());
	}
}
extern "C" int64_t adach_db_data_size(adach_db *h) {
	return h->db->data_size.load();
}
extern "C" uint64_t adach_db_arena_used_bytes(adach_db *h) {
	uint64_t w = 0;
	for (auto &p : h->db->pools) w += p->UsedWords();
	return w * 8;
}
extern "C" uint64_t adach_db_pool_arena_used_bytes(adach_db *h, uint32_t pool) {
	return pool < h->db->pools.size() ? h->db->pools[pool]->UsedWords() * 8 : 0;
}
extern "C" int adach_type_is_supported(int physical_type) {
	return SuccinctFun::TypeIsSupported((PhysicalType)physical_type) ? 1 : 0;
}
