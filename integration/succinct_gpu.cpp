// succinct_gpu.cpp — the DuckDB-side adapter of INTEGRATION.md as real code: the replacement of the reference's
// src/storage/compression/succinct.cpp, written against the reference's OWN headers, as a THIN SHIM over the host
// mirror's C interface (include/adacodec_host.h).  Every callback below forwards to the adach_* call that
// tests/test_gpu_host_mirror.py and tests/test_gpu_adapter_sequence.py drive on the GPU, so the DuckDB-typed slots
// inherit exactly the tested code: pools per GPU, the arena free list, batched compaction, the scan lanes, the
// page-locked decoded-segment cache, representation versions and locks.  Nothing here touches a device, stages a
// row or decides a width itself.
//
// It is not built into libadacodec (that would need the DuckDB tree): tests/test_integration_adapter.py compiles it
// to an object (g++ -c) with the reference's include directories whenever the reference checkout is present, so every
// slot is checked against duckdb::CompressionFunction's typedefs (src/include/duckdb/function/
// compression_function.hpp:65-103) and every member used against ColumnSegment (src/include/duckdb/storage/table/
// column_segment.hpp:40-214).
//
// What changes inside the reference when this file replaces succinct.cpp (INTEGRATION.md §2 has the patch lines):
//   * ColumnSegment::succinct_vec is no longer used: a segment's rows live in its mirror segment (adach_segment),
//     held by the CompressedSegmentState the init_segment slot returns;
//   * ColumnSegment::BitCompressFromSuccinct / BitCompressFromUncompressed / UncompressSuccinct
//     (column_segment.cpp:348-506) become calls of SuccinctCompactOnDevice / SuccinctAdoptUncompressed /
//     SuccinctRestoreUncompressed below; GetDataSize / SuccinctSize (column_segment.cpp:204-222) ask SuccinctDataSize;
//   * ColumnSegmentCatalog::CompactAllSegments and a policy round (column_segment_catalog.cpp:56-116) hand their lists
//     to SuccinctCompactManyOnDevice: one upload, one analyze and one pack launch per pool and type.
#include "duckdb/common/types/null_value.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/function/compression/compression.hpp"
#include "duckdb/function/compression_function.hpp"
#include "duckdb/main/config.hpp"
#include "duckdb/main/database.hpp"
#include "duckdb/storage/buffer_manager.hpp"
#include "duckdb/storage/segment/uncompressed.hpp"
#include "duckdb/storage/statistics/numeric_statistics.hpp"
#include "duckdb/storage/table/append_state.hpp"
#include "duckdb/storage/table/column_segment.hpp"
#include "duckdb/storage/table/scan_state.hpp"

#include "adacodec.h"      // adac_device_count only (this repository's include/)
#include "adacodec_host.h" // everything else

#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>

namespace duckdb {

static void AdachCheck(int rc, const char *what) {
	if (rc != 0) {
		throw InternalException(string("adacodec: ") + what + ": " + adach_last_error());
	}
}

//===--------------------------------------------------------------------===//
// One mirror database (adach_db: one segment pool per visible gfx950 device, adacodec_host.h) per DuckDB instance,
// created on first use from the instance's own DBConfig flags (config.hpp:189-197).  Sizes come from the environment:
// ADAC_ARENA_BYTES (packed arena per GPU, default 8 GiB of the 288 GB) and ADAC_DECODED_CACHE_BYTES (page-locked
// host memory per GPU for decoded segments, default 1 GiB).  There is no CPU fallback: without a device the
// function table cannot be used and the first segment throws.
//===--------------------------------------------------------------------===//
class SuccinctMirror {
public:
	static adach_db *Get(DatabaseInstance &db) {
		static std::mutex lock;
		static std::map<DatabaseInstance *, adach_db *> mirrors; // live as long as the process (the pools own HBM)
		std::lock_guard<std::mutex> guard(lock);
		auto entry = mirrors.find(&db);
		if (entry != mirrors.end()) {
			return entry->second;
		}
		int n = adac_device_count();
		if (n < 1) {
			throw InternalException("adacodec: no gfx950 device (there is no CPU fallback)");
		}
		std::vector<int> devices;
		for (int d = 0; d < n; d++) {
			devices.push_back(d);
		}
		auto &config = DBConfig::GetConfig(db);
		adach_db *mirror = adach_db_create_pools(
		    devices.data(), n, config.succinct_enabled ? 1 : 0, config.adaptive_succinct_compression_enabled ? 1 : 0,
		    config.succinct_padded_to_next_byte_enabled ? 1 : 0, EnvBytes("ADAC_ARENA_BYTES", uint64_t(8) << 30),
		    EnvBytes("ADAC_DECODED_CACHE_BYTES", uint64_t(1) << 30), 0, 0);
		if (!mirror) {
			throw InternalException(string("adacodec: adach_db_create_pools: ") + adach_last_error());
		}
		mirrors[&db] = mirror;
		return mirror;
	}

private:
	static uint64_t EnvBytes(const char *name, uint64_t fallback) {
		const char *env = std::getenv(name);
		return env ? (uint64_t)std::strtoull(env, nullptr, 10) : fallback;
	}
};

//===--------------------------------------------------------------------===//
// Per-segment state (slot init_segment): the handle of the mirror segment, which replaces the succinct_vec member.
//===--------------------------------------------------------------------===//
struct SuccinctSegmentState : public CompressedSegmentState {
	explicit SuccinctSegmentState(adach_segment *handle_p) : handle(handle_p) {
	}
	~SuccinctSegmentState() override {
		adach_segment_destroy(handle); // arena block and cache entry go back to the pool
	}
	adach_segment *handle;
};

static adach_segment *Handle(ColumnSegment &segment) {
	auto state = (SuccinctSegmentState *)segment.GetSegmentState();
	if (!state) {
		throw InternalException("adacodec: succinct segment without its device state");
	}
	return state->handle;
}

static unique_ptr<CompressedSegmentState> SuccinctCreateState(ColumnSegment &segment) {
	adach_segment *handle = adach_segment_create(SuccinctMirror::Get(segment.db), (int)segment.type.InternalType(),
	                                             segment.start, segment.SegmentSize());
	if (!handle) {
		throw InternalException(string("adacodec: adach_segment_create: ") + adach_last_error());
	}
	return unique_ptr<CompressedSegmentState>(new SuccinctSegmentState(handle));
}

static unique_ptr<CompressedSegmentState> SuccinctInitSegment(ColumnSegment &segment, block_id_t block_id) {
	return SuccinctCreateState(segment);
}

//! what the mirror knows after a flip goes back into the members the engine reads (column_segment.hpp:139-171)
static void SuccinctSyncMembers(ColumnSegment &segment) {
	adach_segment *handle = Handle(segment);
	segment.UpdateMinFactor(adach_segment_min(handle));
	segment.UpdateMaxFactor(adach_segment_max(handle));
	if (adach_segment_compacted(handle)) {
		segment.SetBitCompressed();
	} else {
		segment.SetBitUncompressed();
	}
}

//===--------------------------------------------------------------------===//
// Analyze (FixedSize semantics, succinct.cpp:18-40) and compress (UncompressedFunctions: the transient segments
// it creates come back through the append slots below)
//===--------------------------------------------------------------------===//
struct SuccinctAnalyzeState : public AnalyzeState {
	idx_t count = 0;
};
static unique_ptr<AnalyzeState> SuccinctInitAnalyze(ColumnData &col_data, PhysicalType type) {
	return make_unique<SuccinctAnalyzeState>();
}
static bool SuccinctAnalyze(AnalyzeState &state_p, Vector &input, idx_t count) {
	((SuccinctAnalyzeState &)state_p).count += count;
	return true;
}
template <class T>
static idx_t SuccinctFinalAnalyze(AnalyzeState &state_p) {
	return sizeof(T) * ((SuccinctAnalyzeState &)state_p).count;
}

//===--------------------------------------------------------------------===//
// Append (succinct.cpp:264-330).  The rows go to the mirror segment (ColumnSegment::Append of the mirror: staging
// through the selection vector and the validity mask, NullValue<T> in the NULL slots, no arithmetic — min / max come
// from the device when the segment compacts); the engine's zonemap statistics are kept here, as the reference's
// append loop keeps them (succinct.cpp:284,299).
//===--------------------------------------------------------------------===//
static unique_ptr<CompressionAppendState> SuccinctInitAppend(ColumnSegment &segment) {
	auto &buffer_manager = BufferManager::GetBufferManager(segment.db);
	auto handle = buffer_manager.Pin(segment.block);
	return make_unique<CompressionAppendState>(move(handle));
}

template <class T>
static idx_t SuccinctAppend(CompressionAppendState &append_state, ColumnSegment &segment, SegmentStatistics &stats,
                            UnifiedVectorFormat &data, idx_t offset, idx_t count) {
	const validity_t *validity = data.validity.AllValid() ? nullptr : data.validity.GetData();
	int64_t copied = adach_segment_append(Handle(segment), data.data, validity, data.sel->data(), offset, count);
	if (copied < 0) {
		throw InternalException(string("adacodec: adach_segment_append: ") + adach_last_error());
	}
	auto sdata = (T *)data.data;
	for (idx_t i = 0; i < (idx_t)copied; i++) {
		auto source_idx = data.sel->get_index(offset + i);
		if (data.validity.RowIsValid(source_idx)) {
			NumericStatistics::Update<T>(stats, sdata[source_idx]);
		}
	}
	segment.count += copied;
	SuccinctSyncMembers(segment); // the mirror compacts a segment that filled up (column_segment.cpp:266-268)
	return (idx_t)copied;
}

template <class T>
static idx_t SuccinctFinalizeAppend(ColumnSegment &segment, SegmentStatistics &stats) {
	return segment.count * sizeof(T); // succinct.cpp:324-330
}

//===--------------------------------------------------------------------===//
// Bodies for ColumnSegment's three private conversions (column_segment.cpp:348-506) and its size accounting
//===--------------------------------------------------------------------===//

//! BitCompressFromSuccinct for many segments: per (pool, type, rule) one upload, one analyze, one pack
void SuccinctCompactManyOnDevice(const std::vector<ColumnSegment *> &segments) {
	std::map<DatabaseInstance *, std::vector<ColumnSegment *>> by_db;
	for (auto segment : segments) {
		if (segment->function->type == CompressionType::COMPRESSION_SUCCINCT && segment->GetSegmentState()) {
			by_db[&segment->db].push_back(segment);
		}
	}
	for (auto &entry : by_db) {
		std::vector<adach_segment *> handles;
		for (auto segment : entry.second) {
			handles.push_back(Handle(*segment));
		}
		AdachCheck(adach_segments_compact(SuccinctMirror::Get(*entry.first), handles.data(), handles.size()),
		           "adach_segments_compact");
		for (auto segment : entry.second) {
			SuccinctSyncMembers(*segment); // a segment the arena had no room for stays unpacked and is tried again
		}
	}
}

//! column_segment.cpp:348 — void ColumnSegment::BitCompressFromSuccinct() { SuccinctCompactOnDevice(*this); }
void SuccinctCompactOnDevice(ColumnSegment &segment) {
	SuccinctCompactManyOnDevice(std::vector<ColumnSegment *> {&segment});
}

//! column_segment.cpp:385 — BitCompressFromUncompressed (adaptive mode): the rows of an UNCOMPRESSED transient segment
//! move out of its block into a mirror segment, which packs them under the zero-extended rule
//! (column_segment.cpp:390-420: the mirror database was created with the adaptive flag, so its segments start
//! uncompressed and Compact() is that rule).  The caller stores the returned state in segment_state and swaps
//! `function` to SUCCINCT under bit_compression_lock, as column_segment.cpp:451-455 does.
unique_ptr<CompressedSegmentState> SuccinctAdoptUncompressed(ColumnSegment &segment) {
	auto state = SuccinctCreateState(segment);
	adach_segment *handle = ((SuccinctSegmentState &)*state).handle;
	auto &buffer_manager = BufferManager::GetBufferManager(segment.db);
	auto pinned = buffer_manager.Pin(segment.block);
	idx_t count = segment.count;
	int64_t copied = adach_segment_append(handle, pinned.Ptr(), nullptr, nullptr, 0, count);
	if (copied != (int64_t)count) {
		throw InternalException(string("adacodec: adopting an uncompressed segment: ") + adach_last_error());
	}
	AdachCheck(adach_segment_compact(handle), "adach_segment_compact");
	return state;
}

//! column_segment.cpp:458 — UncompressSuccinct: expand on the device, bring the rows back into `target` (the freshly
//! allocated block of column_segment.cpp:461-472).  The caller swaps `function` to UNCOMPRESSED and sets
//! force_reinitializing_scan_state under bit_compression_lock (column_segment.cpp:494-503); segments that were
//! appended through the succinct slots keep their state (the mirror serves them unpacked).
void SuccinctRestoreUncompressed(ColumnSegment &segment, data_ptr_t target) {
	adach_segment *handle = Handle(segment);
	AdachCheck(adach_segment_uncompact(handle), "adach_segment_uncompact");
	if (target && segment.count > 0) {
		AdachCheck(adach_segment_scan(handle, segment.start, segment.count, target, 0, 1), "adach_segment_scan");
	}
	SuccinctSyncMembers(segment);
}

//! column_segment.cpp:204-222 — GetDataSize / SuccinctSize: sdsl::size_in_bytes(succinct_vec) of the current form
idx_t SuccinctDataSize(ColumnSegment &segment) {
	return adach_segment_data_size(Handle(segment));
}

//===--------------------------------------------------------------------===//
// Scan (succinct.cpp:123-144, 232-240) and fetch (succinct.cpp:244-260, intended semantics).
// init_scan returns the state that lives in ColumnScanState::scan_state from one scan_vector call to the next
// (ColumnData::ScanVector, column_data.cpp:92-139): the mirror pins the segment's decoded block in it — decoded once
// per segment, together with the next segments of the column, ahead of the consumer — so a 2048-row call is a memcpy
// out of page-locked memory and no call goes to the device.
//===--------------------------------------------------------------------===//
struct SuccinctScanState : public SegmentScanState {
	SuccinctScanState() : state(adach_scan_state_create()) {
		if (!state) {
			throw InternalException("adacodec: adach_scan_state_create");
		}
	}
	~SuccinctScanState() override {
		adach_scan_state_destroy(state); // releases the pin
	}
	adach_scan_state *state;
};

static unique_ptr<SegmentScanState> SuccinctInitScan(ColumnSegment &segment) {
	adach_segment *handle = Handle(segment);
	// SegmentBase::next as the mirror's decode-ahead hint: the following segment of the column, once it exists
	auto next = (ColumnSegment *)segment.Next();
	if (next && next->function->type == CompressionType::COMPRESSION_SUCCINCT && next->GetSegmentState()) {
		AdachCheck(adach_segment_set_next(handle, Handle(*next)), "adach_segment_set_next");
	}
	auto result = make_unique<SuccinctScanState>();
	AdachCheck(adach_segment_init_scan(handle, result->state), "adach_segment_init_scan");
	return move(result);
}

template <class T>
static void SuccinctScanPartial(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result,
                                idx_t result_offset) {
	auto &scan_state = (SuccinctScanState &)*state.scan_state;
	result.SetVectorType(VectorType::FLAT_VECTOR);
	AdachCheck(adach_segment_scan_with(Handle(segment), scan_state.state, state.row_index, scan_count,
	                                   FlatVector::GetData(result), result_offset, 0),
	           "adach_segment_scan_with");
}

template <class T>
static void SuccinctScan(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result) {
	SuccinctScanPartial<T>(segment, state, scan_count, result, 0);
}

//! the mirror reads under the segment's lock, from the decoded block when the segment has one (no device call),
//! else with one range decode; ColumnSegment::FetchRow hands over row_id - start (column_segment.cpp:193-195)
template <class T>
static void SuccinctFetchRow(ColumnSegment &segment, ColumnFetchState &state, row_t row_id, Vector &result,
                             idx_t result_idx) {
	AdachCheck(adach_segment_fetch_row(Handle(segment), row_id + (row_t)segment.start, FlatVector::GetData(result),
	                                   result_idx),
	           "adach_segment_fetch_row");
}

//===--------------------------------------------------------------------===//
// Get Function (succinct.cpp:335-382): same table shape; init_segment now carries the device handle
//===--------------------------------------------------------------------===//
template <class T>
static CompressionFunction SuccinctGetFunction(PhysicalType type) {
	return CompressionFunction(CompressionType::COMPRESSION_SUCCINCT, type, SuccinctInitAnalyze, SuccinctAnalyze,
	                           SuccinctFinalAnalyze<T>, UncompressedFunctions::InitCompression,
	                           UncompressedFunctions::Compress, UncompressedFunctions::FinalizeCompress,
	                           SuccinctInitScan, SuccinctScan<T>, SuccinctScanPartial<T>, SuccinctFetchRow<T>,
	                           UncompressedFunctions::EmptySkip, SuccinctInitSegment, SuccinctInitAppend,
	                           SuccinctAppend<T>, SuccinctFinalizeAppend<T>, nullptr);
}

CompressionFunction SuccinctFun::GetFunction(PhysicalType data_type) {
	switch (data_type) {
	case PhysicalType::INT8:
		return SuccinctGetFunction<int8_t>(data_type);
	case PhysicalType::UINT8:
		return SuccinctGetFunction<uint8_t>(data_type);
	case PhysicalType::INT16:
		return SuccinctGetFunction<int16_t>(data_type);
	case PhysicalType::UINT16:
		return SuccinctGetFunction<uint16_t>(data_type);
	case PhysicalType::INT32:
		return SuccinctGetFunction<int32_t>(data_type);
	case PhysicalType::UINT32:
		return SuccinctGetFunction<uint32_t>(data_type);
	case PhysicalType::INT64:
		return SuccinctGetFunction<int64_t>(data_type);
	case PhysicalType::UINT64:
		return SuccinctGetFunction<uint64_t>(data_type);
	default:
		throw InternalException("Unsupported type for FixedSizeSuccinct::GetFunction");
	}
}

bool SuccinctFun::TypeIsSupported(PhysicalType type) {
	return adach_type_is_supported((int)type) != 0; // duckdb::PhysicalType codes are the mirror's type codes
}

} // namespace duckdb
