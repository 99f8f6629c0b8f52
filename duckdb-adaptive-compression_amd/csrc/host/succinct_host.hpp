// succinct_host.hpp — host-side mirror of the reference's succinct plumbing, in the reference's own language
// (C++) and with its names, built ONLY on the C ABI of include/adacodec.h (device work) and memcpy (host work:
// no value arithmetic happens on the CPU).  It is what the DuckDB adapter of INTEGRATION.md would be made of,
// kept self-contained (no DuckDB headers) so it builds and is tested on its own.
//
// Mirrors (paths relative to the reference checkout):
//   DBConfig flags                      src/include/duckdb/main/config.hpp:189-197
//   CompressionFunction (slot table)    src/include/duckdb/function/compression_function.hpp:65-177
//   SuccinctFun::GetFunction/TypeIs...  src/storage/compression/succinct.cpp:335-382
//   ColumnSegment succinct state        src/include/duckdb/storage/table/column_segment.hpp:60-64,139-214
//     CreateTransientSegment / ctor     src/storage/table/column_segment.cpp:45-110
//     InitializeScan / Scan / FetchRow  src/storage/table/column_segment.cpp:133-196
//     Append / Compact / Uncompact      src/storage/table/column_segment.cpp:247-346
//     GetDataSize / SuccinctSize        src/storage/table/column_segment.cpp:204-222
//     ConvertToPersistent (empty there) src/storage/table/column_segment.cpp:529-533
//   ColumnData::ScanVector (caller)     src/storage/table/column_data.cpp:92-139
//   ColumnSegmentCatalog                src/catalog/catalog_entry/column_segment_catalog.cpp:24-135
//
// MI355X shape: a database owns one SegmentPool per GPU (north star: "per-GPU segment pools"); a segment lives in
// the pool its id selects (id mod pools) for its whole life.  A pool is a packed arena in HBM, a main stream for
// representation flips (compaction, expansion, persistence), and a few SCAN LANES — streams with their own device
// staging — on which whole decoded segments are produced a batch at a time, ahead of the consumer, into a cache of
// page-locked host blocks; the engine's 2048-row scan_vector calls are then memcpys out of a block the scan state
// has pinned (what init_scan's buffer pin is in the reference, fixed_size_uncompressed.cpp:125-130).
#pragma once

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "adacodec.h"

#ifndef ADACH_ZERO_COPY_DEFAULT
#define ADACH_ZERO_COPY_DEFAULT true
#endif

namespace adacodec {

using idx_t = uint64_t;
using row_t = int64_t;
using data_ptr_t = uint8_t *;

constexpr idx_t STANDARD_VECTOR_SIZE = 2048; // src/include/duckdb/common/vector_size.hpp:17
constexpr idx_t BLOCK_SIZE = 262144 - 8;     // Storage::BLOCK_SIZE, src/include/duckdb/common/constants.hpp:104-106

// duckdb::PhysicalType codes of the supported types (src/include/duckdb/common/types.hpp:117-138)
enum class PhysicalType : uint8_t { UINT8 = 2, INT8 = 3, UINT16 = 4, INT16 = 5, UINT32 = 6, INT32 = 7, UINT64 = 8, INT64 = 9 };

// src/include/duckdb/common/enums/compression_type.hpp:17-27
enum class CompressionType : uint8_t { COMPRESSION_UNCOMPRESSED = 1, COMPRESSION_SUCCINCT = 10 };

// ColumnSegmentType (column_segment.hpp): a persistent segment came from / has been written as a block image
enum class ColumnSegmentType : uint8_t { TRANSIENT, PERSISTENT };

class InternalException : public std::runtime_error {
public:
	explicit InternalException(const std::string &msg) : std::runtime_error(msg) {
	}
};

// The four public booleans of DBConfig (config.hpp:189-197).
struct DBConfig {
	bool succinct_enabled = true;
	bool succinct_extract_prefix_enabled = true; // read only by the reference's dead SuccinctScanAndCompact
	bool succinct_padded_to_next_byte_enabled = false;
	bool adaptive_succinct_compression_enabled = false;
	// Not in the reference: bytes of page-locked host memory PER POOL used to keep whole DECODED segments, so the
	// engine's 2048-row scan_vector calls (ColumnData::ScanVector, src/storage/table/column_data.cpp:92-139)
	// cost one device decode + one PCIe copy per SEGMENT instead of per vector. 0 = off.
	uint64_t decoded_cache_bytes = 0;
	uint32_t scan_lanes = 4;        // decode + copy-down streams per pool
	uint32_t prefetch_segments = 8; // segments per decode batch = how far a scan decodes ahead of its consumer
	// true: the decode kernels of the scan lanes store straight into the page-locked cache blocks (zero copy);
	// false: they decode into device staging and a copy engine brings the blocks down
	bool zero_copy_decode = ADACH_ZERO_COPY_DEFAULT;
};

// The slice of duckdb::UnifiedVectorFormat the append slot reads (data, selection vector, validity mask).
struct UnifiedVectorFormat {
	const uint8_t *data = nullptr;
	const uint32_t *sel = nullptr;      // nullptr = identity
	const uint64_t *validity = nullptr; // nullptr = all valid; bit i of word i/64 set = row i valid
};

// The slice of duckdb::Vector a scan writes: flat data of the segment's type.
struct Vector {
	data_ptr_t data = nullptr;
	bool flat = true;
	const uint64_t *validity = nullptr; // read by the compress slot (Vector::ToUnifiedFormat); nullptr = all valid
};

class ColumnSegment;
class ColumnSegmentCatalog;
class DatabaseInstance;
class SegmentPool;

// States the slots hand back to the engine (compression_function.hpp:29-63): owned by the caller, virtual dtors.
struct AnalyzeState {
	virtual ~AnalyzeState() = default;
};
struct CompressionState {
	virtual ~CompressionState() = default;
};
struct SegmentScanState {
	virtual ~SegmentScanState() = default;
};
struct CompressionAppendState {
	virtual ~CompressionAppendState() = default;
};

// The slice of duckdb::ColumnScanState the path uses (src/include/duckdb/storage/table/scan_state.hpp):
// scan_state is what init_scan returned for `current`; ColumnSegment::InitializeScan replaces it.
struct ColumnScanState {
	idx_t row_index = 0; // absolute row; the segment subtracts its start
	ColumnSegment *current = nullptr;
	bool initialized = false;
	std::unique_ptr<SegmentScanState> scan_state;
};
struct ColumnFetchState {};

// The slice of duckdb::ColumnDataCheckpointer the compress slots use (GetDatabase, GetType, GetRowGroup().start,
// GetCheckpointState().FlushSegment): flushed segments are collected here in order.  FlushSegment is also where
// the engine calls ColumnSegment::ConvertToPersistent; the block images it yields are collected in flushed_blocks
// (filled by compress_finalize for the whole column in one device pass).
struct ColumnDataCheckpointer {
	ColumnDataCheckpointer(DatabaseInstance &db, PhysicalType type, idx_t row_group_start);
	~ColumnDataCheckpointer();
	DatabaseInstance &db;
	PhysicalType type;
	idx_t row_group_start;
	std::vector<std::unique_ptr<ColumnSegment>> flushed_segments;
	std::vector<idx_t> flushed_sizes; // the segment_size argument of FlushSegment (bytes used)
	std::vector<std::vector<uint8_t>> flushed_blocks; // block image of flushed_segments[i] (SUCCINCT only)
};

// The function-pointer table (compression_function.hpp:105-177), all sixteen slots in the reference's order as
// SuccinctGetFunction fills them (succinct.cpp:335-343): analyze borrowed from FixedSize*, the checkpoint-side
// compress slots append into transient segments (the reference prints "SHOULD NOT HAPPEN" there but wires them),
// init_segment and revert_append are nullptr.
struct CompressionFunction {
	CompressionType type;
	PhysicalType data_type;
	std::unique_ptr<AnalyzeState> (*init_analyze)(PhysicalType type);
	bool (*analyze)(AnalyzeState &state, Vector &input, idx_t count);
	idx_t (*final_analyze)(AnalyzeState &state);
	std::unique_ptr<CompressionState> (*init_compression)(ColumnDataCheckpointer &checkpointer,
	                                                     std::unique_ptr<AnalyzeState> state);
	void (*compress)(CompressionState &state, Vector &scan_vector, idx_t count);
	void (*compress_finalize)(CompressionState &state);
	std::unique_ptr<SegmentScanState> (*init_scan)(ColumnSegment &segment);
	void (*scan_vector)(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result);
	void (*scan_partial)(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result,
	                     idx_t result_offset);
	void (*fetch_row)(ColumnSegment &segment, ColumnFetchState &state, row_t row_id, Vector &result, idx_t result_idx);
	void (*skip)(ColumnSegment &segment, ColumnScanState &state, idx_t skip_count);
	void *(*init_segment)(ColumnSegment &segment, int64_t block_id); // nullptr for both codecs
	std::unique_ptr<CompressionAppendState> (*init_append)(ColumnSegment &segment);
	idx_t (*append)(ColumnSegment &segment, UnifiedVectorFormat &data, idx_t offset, idx_t count);
	idx_t (*finalize_append)(ColumnSegment &segment);
	void (*revert_append)(ColumnSegment &segment, idx_t start_row); // nullptr for both codecs
};

struct SuccinctFun {
	static CompressionFunction GetFunction(PhysicalType data_type); // throws InternalException if unsupported
	static bool TypeIsSupported(PhysicalType type);
};
struct UncompressedFun {
	static CompressionFunction GetFunction(PhysicalType data_type);
};

// One decode + copy-down of a few segments on a scan lane.  Consumers wait on it without holding any pool lock.
struct DecodeBatch {
	~DecodeBatch();
	void Publish(adac_event *ev, bool ok); // called once by the enqueuer
	bool Wait();                           // true when the decoded bytes are in host memory
	bool Finished();                       // no transfer can still be writing the blocks (true also after a failure)

private:
	std::mutex m;
	std::condition_variable cv;
	int state = 0; // 0 being enqueued, 1 enqueued, 2 finished, -1 failed
	adac_event *ev = nullptr;
};

// One decoded segment in the page-locked cache of a pool.
struct CacheEntry {
	uint64_t key = 0;     // segment id
	uint64_t version = 0; // representation version of the segment the bytes belong to
	uint8_t *data = nullptr;
	size_t bytes = 0;
	int32_t slot = -1;
	int pins = 0;         // scan states reading the block
	bool dropped = false; // no longer in the map: the slot is reclaimed when pins == 0 and the batch has finished
	bool trigger = false; // first touch schedules the batch that follows
	std::shared_ptr<DecodeBatch> batch;
	std::list<CacheEntry *>::iterator lru;
};

struct ScanLane {
	adac_ctx *ctx = nullptr;
	std::mutex lock;         // one enqueuer at a time
	void *d_stage = nullptr; // prefetch_segments x slot bytes of device memory the lane decodes into
};

// One GPU's segment pool.
class SegmentPool {
public:
	SegmentPool(int index, int device, size_t arena_bytes, const DBConfig &config);
	~SegmentPool();
	SegmentPool(const SegmentPool &) = delete;
	const int index, device;

	// ---- main stream: representation flips, persistence, uncached reads.  `lock` serialises its users.
	adac_ctx *ctx = nullptr;
	std::mutex lock;
	void *Staging(size_t bytes);          // device scratch, grown on demand (under lock)
	void *Staging2(size_t bytes);         // second device scratch (validity)
	uint8_t *PinnedStaging(size_t bytes); // page-locked host staging, grown on demand (under lock)

	// Serialises representation flips (Compact / CompactMany / Uncompact / Append / destruction) of THIS pool's
	// segments.  The reference has a single policy thread plus scan-triggered compaction and locks only the
	// function-pointer swap; here a flip stages the unpacked rows outside the segment lock, so two flips of one
	// segment must not overlap, an Append must not see its segment flip between its Uncompact and its write, and
	// a policy round must not meet a segment that is being destroyed.
	// Lock order: flip_lock -> catalog.lock -> bit_compression_lock -> chain_lock -> cache_lock; pool.lock after
	// bit_compression_lock; arena_lock is a leaf.
	std::recursive_mutex flip_lock;

	// ---- packed arena (first fit over 128-byte units)
	uint64_t *d_arena = nullptr;
	uint64_t arena_words = 0;
	bool TryAllocate(uint64_t words, uint64_t &word_off); // false when no block is large enough
	void Free(uint64_t word_off, uint64_t words);
	uint64_t UsedWords();
	std::atomic<uint64_t> exhausted_events {0}; // segments a compaction left unpacked for lack of arena space
	std::atomic<uint64_t> free_generation {0};  // bumped by every Free: "has space come back since I was refused?"

	// ---- decoded-segment cache: one page-locked slab cut into block-sized slots, LRU, entries pinned by scan states
	static constexpr size_t kCacheSlotBytes = 262144;
	const uint32_t prefetch_segments;
	uint64_t cache_capacity = 0;
	std::mutex cache_lock;
	std::unordered_map<uint64_t, CacheEntry *> cache; // segment id -> entry
	std::list<CacheEntry *> lru;                      // front = most recently used
	std::vector<CacheEntry *> zombies;                // dropped entries whose slot is still pinned or in flight
	std::set<int32_t> free_slots;
	uint8_t *cache_slab = nullptr;
	std::atomic<uint64_t> cache_hits {0}, cache_misses {0}, cache_batches {0}, cache_prefetched {0};
	uint64_t CacheUsedBytes();
	// the three *Locked need cache_lock held
	CacheEntry *CacheInsertLocked(uint64_t key, uint64_t version, size_t bytes); // nullptr when no slot can be had
	void CacheDropLocked(uint64_t key);
	void CacheReclaimLocked();
	void CacheUnpin(CacheEntry *e);
	void CacheDrop(uint64_t key);

	// ---- scan lanes
	std::vector<std::unique_ptr<ScanLane>> lanes;
	ScanLane &AcquireLane(std::unique_lock<std::mutex> &held);

private:
	void Release();
	std::mutex arena_lock;
	std::map<uint64_t, uint64_t> free_list; // offset -> length
	uint64_t used_words = 0;
	std::atomic<uint32_t> next_lane {0};
	void *d_staging = nullptr;
	size_t staging_bytes = 0;
	void *d_staging2 = nullptr;
	size_t staging2_bytes = 0;
	void *h_pinned = nullptr;
	size_t pinned_bytes = 0;
};

// column_segment_catalog.hpp:23-49.  The read counters live in the segments (atomics): the reference's
// AddReadAccess mutates an unordered_map from every scanning thread without a lock (its TSan suppressions cover it);
// here the map is only the registry and a read access is one relaxed increment.
class ColumnSegmentCatalog {
public:
	explicit ColumnSegmentCatalog(DatabaseInstance &db);
	~ColumnSegmentCatalog();
	void AddColumnSegment(ColumnSegment *segment);
	void AddReadAccess(ColumnSegment *segment);
	void RemoveColumnSegment(ColumnSegment *segment);
	void CompactAllSegments();
	size_t GetTotalDataSize();
	// One iteration of CompressLowestKSegments (column_segment_catalog.cpp:64-116) without the sleep: sort by
	// num_reads, Compact the first `compression_rate` share, Uncompact the rest, reset the counters.  The flips run
	// pool by pool (one host thread per pool when there are several), each under its pool's flip_lock.
	void CompressLowestKSegmentsOnce(double compression_rate = 0.90);
	void EnableBackgroundThreadCompaction(unsigned period_ms = 10000);
	void DisableBackgroundThreadCompaction();
	bool BackgroundCompactionEnabled() const {
		return background_compaction_enabled;
	}
	idx_t NumSegments();
	idx_t EventCounter() const {
		return event_counter;
	}
	// what the background thread met: rounds run, rounds that threw, and the last message (the thread survives)
	idx_t BackgroundRounds() const {
		return background_rounds;
	}
	idx_t BackgroundErrors() const {
		return background_errors;
	}
	std::string LastBackgroundError();

private:
	DatabaseInstance &db;
	std::mutex lock;
	std::unordered_set<ColumnSegment *> segments;
	std::atomic<idx_t> event_counter {0};
	std::atomic<bool> background_compaction_enabled {false};
	std::atomic<bool> stop {false};
	std::atomic<idx_t> background_rounds {0}, background_errors {0};
	std::string last_background_error;
	std::thread worker;
};

class DatabaseInstance {
public:
	// devices[i] = HIP device of pool i (the same device may appear more than once: several pools on one GPU)
	DatabaseInstance(const std::vector<int> &devices, const DBConfig &config, size_t arena_bytes_per_pool);
	~DatabaseInstance();
	DBConfig config;
	std::vector<std::unique_ptr<SegmentPool>> pools;
	std::atomic<int64_t> data_size {0}; // BufferManager::data_size accounting (buffer_manager.hpp:71-82)
	std::atomic<uint64_t> next_segment_id {0};
	std::mutex chain_lock; // SegmentBase::next / prev hints of all segments
	ColumnSegmentCatalog catalog;
	SegmentPool &PoolFor(uint64_t segment_id) {
		return *pools[segment_id % pools.size()];
	}
	const CompressionFunction *GetCompressionFunction(CompressionType type, PhysicalType data_type);

private:
	std::mutex fn_lock;
	std::map<std::pair<uint8_t, uint8_t>, CompressionFunction> functions; // lazy registry, compression_config.cpp:85-94
};

class ColumnSegment {
public:
	static std::unique_ptr<ColumnSegment> CreateTransientSegment(DatabaseInstance &db, PhysicalType type, idx_t start,
	                                                             idx_t segment_size = BLOCK_SIZE);
	// Load: segments from block images (ColumnSegment::CreatePersistentSegment, column_segment.cpp:25-43, for a codec
	// whose images live in HBM): one upload + one device pass per pool for the whole batch.
	static std::vector<std::unique_ptr<ColumnSegment>>
	CreatePersistentSegments(DatabaseInstance &db, const std::vector<std::pair<const uint8_t *, size_t>> &images,
	                         const std::vector<idx_t> &starts);
	~ColumnSegment();

	DatabaseInstance &db;
	const uint64_t segment_id;
	SegmentPool &pool;
	PhysicalType type;
	idx_t type_size;
	idx_t start;
	idx_t count = 0; // SegmentBase::count
	const CompressionFunction *function;
	ColumnSegmentType segment_type = ColumnSegmentType::TRANSIENT;
	bool succinct_possible;
	bool is_data_segment = true;
	std::atomic<idx_t> num_reads {0}; // AccessStatistics::num_reads
	// SegmentBase::next (src/include/duckdb/storage/table/segment_base.hpp): the following segment of the column.
	// Not owned; followed by ColumnData::ScanVector's mirror and by the decode-ahead of a scan (db.chain_lock).
	ColumnSegment *next_hint = nullptr;
	ColumnSegment *prev_hint = nullptr; // so that a destroyed segment can unlink itself
	void SetNext(ColumnSegment *next);
	ColumnSegment *Next();

	void InitializeScan(ColumnScanState &state);
	void Scan(ColumnScanState &state, idx_t scan_count, Vector &result, idx_t result_offset, bool entire_vector);
	void FetchRow(ColumnFetchState &state, row_t row_id, Vector &result, idx_t result_idx);
	void Skip(ColumnScanState &state);
	idx_t SegmentSize() const {
		return segment_size;
	}
	idx_t GetDataSize() const;
	idx_t SuccinctSize() const;
	idx_t Append(UnifiedVectorFormat &data, idx_t offset, idx_t count);
	idx_t FinalizeAppend();
	idx_t GetRelativeIndex(idx_t row_index) const {
		return row_index - start;
	}
	uint64_t GetMinFactor() const {
		return min_factor;
	}
	uint64_t GetMax() const {
		return max_factor;
	}
	bool IsBitCompressed() const {
		return compacted;
	}
	void Compact();
	void Uncompact();
	uint8_t Width() const {
		return vec_width;
	}
	idx_t NumElements() const {
		return num_elements;
	}

	// batched forms: one upload, one analyze, one pack per (pool, type, rule) for many segments
	static void CompactMany(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments);
	// The ConvertToPersistent the reference leaves empty for SUCCINCT (column_segment.cpp:529-533): the block image
	// (include/adacodec.h: sdsl::int_vector<0>::serialize + trailer) of every segment, built in HBM per pool and
	// brought down with one copy.  Segments that still need compaction are compacted first.
	static void ConvertManyToPersistent(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments,
	                                    std::vector<std::vector<uint8_t>> &images);
	void ConvertToPersistent(std::vector<uint8_t> &image);

	// codec internals reached by the CompressionFunction callbacks (state: the scan's, or nullptr for a one-off read)
	void ScanRows(ColumnScanState *state, idx_t start_row, idx_t scan_count, data_ptr_t target);
	idx_t AppendRows(UnifiedVectorFormat &data, idx_t offset, idx_t count);

	ColumnSegment(DatabaseInstance &db, PhysicalType type, idx_t start, idx_t segment_size, const CompressionFunction *fn,
	              bool succinct_possible, bool background_compaction_enabled);

private:
	friend class ColumnSegmentCatalog;
	friend struct SuccinctScanState;
	bool NeedsCompaction() const;
	static void CompactGroup(DatabaseInstance &db, SegmentPool &pool, int ptype, int rule,
	                         const std::vector<ColumnSegment *> &segs);
	void FinishCompaction(bool packed, uint8_t width, uint64_t mn, uint64_t mx, int rule, uint64_t word_off,
	                      uint64_t stored_min, std::vector<std::vector<uint8_t>> *graveyard = nullptr);
	adac_segment_desc DeviceDesc() const; // word_off, count, width, stored min, flags of the packed form
	// decoded image of this segment, pinned; schedules the decode of this and the following segments when needed
	CacheEntry *PinDecoded(bool may_schedule);
	void DecodeRowsDirect(idx_t start_row, idx_t scan_count, data_ptr_t target);

	idx_t num_elements = 0;
	idx_t segment_size;
	uint64_t min_factor = UINT64_MAX;
	uint64_t max_factor = 0;
	bool compacted = false;
	bool background_compaction_enabled;
	std::mutex bit_compression_lock;
	uint64_t version = 0; // bumped by every representation flip: scan states and cache entries carry the one they saw

	// succinct_vec, as {slots, width}; its bits live in `raw` (unpacked) or in the pool arena (packed)
	idx_t vec_slots = 0;
	uint8_t vec_width = 64;
	bool packed_on_device = false;
	uint64_t word_off = 0, arena_words = 0;
	uint64_t device_min = UINT64_MAX; // the descriptor's min (adac_stored_min of min_factor / max_factor)
	std::vector<uint8_t> raw;          // slots at 8*sizeof(T) bits / the uncompressed block
	std::vector<uint64_t> validity;    // NULL rows of the append phase (consumed by the first compaction)
	bool any_null = false;
	bool appended_via_succinct = false; // true: min/max follow the append rule; false: the recompaction rule
	bool arena_refused = false;         // the last compaction found no room in the arena ...
	uint64_t refused_generation = 0;    // ... at this free_generation of the pool
};

} // namespace adacodec
