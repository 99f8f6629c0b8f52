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
//     Scan / ScanPartial / FetchRow     src/storage/table/column_segment.cpp:137-196
//     Append / Compact / Uncompact      src/storage/table/column_segment.cpp:247-346
//     GetDataSize / SuccinctSize        src/storage/table/column_segment.cpp:204-222
//   ColumnSegmentCatalog                src/catalog/catalog_entry/column_segment_catalog.cpp:24-135
#pragma once

#include <atomic>
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <unordered_map>
#include <vector>

#include "adacodec.h"

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
	// Not in the reference: bytes of page-locked host memory used to keep whole DECODED segments, so the
	// engine's 2048-row scan_vector calls (ColumnData::ScanVector, src/storage/table/column_data.cpp:92-139)
	// cost one device decode + one PCIe copy per SEGMENT instead of per vector. 0 = off.
	uint64_t decoded_cache_bytes = 0;
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

struct ColumnScanState {
	idx_t row_index = 0; // absolute row; the segment subtracts its start
};
struct ColumnFetchState {};

class ColumnSegment;
class ColumnSegmentCatalog;
class DatabaseInstance;

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

// The slice of duckdb::ColumnDataCheckpointer the compress slots use (GetDatabase, GetType, GetRowGroup().start,
// GetCheckpointState().FlushSegment): flushed segments are collected here in order.
struct ColumnDataCheckpointer {
	ColumnDataCheckpointer(DatabaseInstance &db, PhysicalType type, idx_t row_group_start);
	~ColumnDataCheckpointer();
	DatabaseInstance &db;
	PhysicalType type;
	idx_t row_group_start;
	std::vector<std::unique_ptr<ColumnSegment>> flushed_segments;
	std::vector<idx_t> flushed_sizes; // the segment_size argument of FlushSegment (bytes used)
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

// One GPU's segment pool: context, packed arena (first-fit over 128-byte units) and staging buffers.
class SegmentPool {
public:
	SegmentPool(int device, size_t arena_bytes);
	~SegmentPool();
	adac_ctx *ctx = nullptr;
	uint64_t *d_arena = nullptr;
	uint64_t arena_words = 0;

	uint64_t Allocate(uint64_t words); // returns word offset (multiple of 16); throws when exhausted
	void Free(uint64_t word_off, uint64_t words);
	uint64_t UsedWords() const {
		return used_words;
	}
	void *Staging(size_t bytes);   // device scratch, grown on demand
	void *Staging2(size_t bytes);  // second device scratch (validity)
	uint8_t *PinnedStaging(size_t bytes); // page-locked host staging for uploads, grown on demand
	std::mutex lock;               // serialises device work of this pool (one stream)

	// Decoded-segment cache (page-locked host blocks, LRU by bytes).  Guarded by `lock`.
	struct CacheEntry {
		uint8_t *data = nullptr;
		size_t bytes = 0;
		uint64_t stamp = 0;
		int32_t slot = -1; // >= 0: a slot of the slab; -1: its own page-locked allocation (oversized segment)
		bool pending = false; // a prefetch (decode + async copy on the pool's stream) is still in flight
	};
	uint64_t cache_capacity = 0, cache_used = 0, cache_clock = 0, cache_hits = 0, cache_misses = 0;
	std::unordered_map<const void *, CacheEntry> cache;
	// The cache's page-locked memory is ONE slab cut into block-sized slots: page-locking costs ~50 us per 256 KiB
	// block, which made every cold segment pay more for its buffer than for its decode and copy.
	static constexpr size_t kCacheSlotBytes = 262144;
	uint8_t *cache_slab = nullptr;
	std::vector<int32_t> cache_free_slots;
	void CacheReserve(); // allocates the slab (idempotent); called when a cache capacity is configured
	void *PrefetchStaging(size_t bytes); // device scratch of the in-flight prefetch (separate from Staging)
	void CacheSettle(CacheEntry &e);     // waits for the entry's prefetch, if any
	uint64_t cache_prefetches = 0;
	const uint8_t *CacheLookup(const void *key);
	uint8_t *CacheInsert(const void *key, size_t bytes); // evicts least-recently-used entries; nullptr if too big
	void CacheDrop(const void *key);

private:
	std::map<uint64_t, uint64_t> free_list; // offset -> length
	uint64_t used_words = 0;
	void *d_prefetch = nullptr;
	size_t prefetch_bytes = 0;
	void *d_staging = nullptr;
	size_t staging_bytes = 0;
	void *d_staging2 = nullptr;
	size_t staging2_bytes = 0;
	void *h_pinned = nullptr;
	size_t pinned_bytes = 0;
};

struct AccessStatistics {
	idx_t num_reads = 0;
};

// column_segment_catalog.hpp:23-49
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
	// num_reads, Compact the first `compression_rate` share, Uncompact the rest, reset the counters.
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

private:
	DatabaseInstance &db;
	std::mutex lock; // the reference mutates the map unlocked (TSan suppression `race:~ColumnSegment`)
	std::unordered_map<ColumnSegment *, AccessStatistics> statistics;
	std::atomic<idx_t> event_counter {0};
	std::atomic<bool> background_compaction_enabled {false};
	std::atomic<bool> stop {false};
	std::thread worker;
};

class DatabaseInstance {
public:
	DatabaseInstance(int device, const DBConfig &config, size_t arena_bytes);
	DBConfig config;
	SegmentPool pool;
	ColumnSegmentCatalog catalog;
	std::atomic<int64_t> data_size {0}; // BufferManager::data_size accounting (buffer_manager.hpp:71-82)
	// Serialises representation flips (Compact / CompactMany / Uncompact) of this database.  The reference has a
	// single policy thread plus scan-triggered compaction and locks only the function-pointer swap; here a flip
	// stages the unpacked rows outside the segment lock, so two concurrent flips of one segment (two policy
	// threads, or two first scans) must not overlap, and an Append must not see its segment flip between its
	// Uncompact and its write.  Order: flip_lock -> bit_compression_lock -> pool.lock.
	std::recursive_mutex flip_lock; // recursive: Append holds it across its own Uncompact / Compact
	const CompressionFunction *GetCompressionFunction(CompressionType type, PhysicalType data_type);

private:
	std::mutex fn_lock;
	std::map<std::pair<uint8_t, uint8_t>, CompressionFunction> functions; // lazy registry, compression_config.cpp:85-94
};

class ColumnSegment {
public:
	static std::unique_ptr<ColumnSegment> CreateTransientSegment(DatabaseInstance &db, PhysicalType type, idx_t start,
	                                                             idx_t segment_size = BLOCK_SIZE);
	~ColumnSegment();

	DatabaseInstance &db;
	PhysicalType type;
	idx_t type_size;
	idx_t start;
	idx_t count = 0; // SegmentBase::count
	const CompressionFunction *function;
	bool succinct_possible;
	bool is_data_segment = true;
	// SegmentBase::next (src/include/duckdb/storage/table/segment_base.hpp): the following segment of the column.
	// Not owned; used only as a hint to start decoding it while the consumer is still reading this one.
	ColumnSegment *next_hint = nullptr;
	ColumnSegment *prev_hint = nullptr; // so that a destroyed segment can unlink itself
	void SetNext(ColumnSegment *next);

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

	// batched forms used by the catalog: one upload, one analyze, one pack for many segments
	static void CompactMany(DatabaseInstance &db, const std::vector<ColumnSegment *> &segments);

	// codec internals reached by the CompressionFunction callbacks
	void ScanRows(idx_t start_row, idx_t scan_count, data_ptr_t target);
	// decode + async copy of THIS segment into the pool's cache, on behalf of a scan of `reader` (pool.lock held)
	void PrefetchIntoCache(const ColumnSegment *reader);
	idx_t AppendRows(UnifiedVectorFormat &data, idx_t offset, idx_t count);

	ColumnSegment(DatabaseInstance &db, PhysicalType type, idx_t start, idx_t segment_size, const CompressionFunction *fn,
	              bool succinct_possible, bool background_compaction_enabled);

private:
	friend class ColumnSegmentCatalog;
	bool NeedsCompaction() const;
	void FinishCompaction(bool packed, uint8_t width, uint64_t mn, uint64_t mx, int rule, uint64_t word_off,
	                      std::vector<std::vector<uint8_t>> *graveyard = nullptr);

	idx_t num_elements = 0;
	idx_t segment_size;
	uint64_t min_factor = UINT64_MAX;
	uint64_t max_factor = 0;
	bool compacted = false;
	bool background_compaction_enabled;
	bool force_reinitializing_scan_state = false;
	std::mutex bit_compression_lock;

	// succinct_vec, as {slots, width}; its bits live in `raw` (unpacked) or in the pool arena (packed)
	idx_t vec_slots = 0;
	uint8_t vec_width = 64;
	bool packed_on_device = false;
	uint64_t word_off = 0, arena_words = 0;
	std::shared_ptr<void> device_layout; // the batch adac_layout this segment was packed with (shared by the batch)
	uint64_t layout_index = 0;           // this segment's index inside it
	std::vector<uint8_t> raw;          // slots at 8*sizeof(T) bits / the uncompressed block
	std::vector<uint64_t> validity;    // NULL rows of the append phase (consumed by the first compaction)
	bool any_null = false;
	bool appended_via_succinct = false; // true: min/max follow the append rule; false: the recompaction rule
};

} // namespace adacodec
