// succinct_gpu.cpp — the DuckDB-side adapter of INTEGRATION.md as real code: the callbacks of
// src/storage/compression/succinct.cpp (reference) re-implemented on libadacodec's C ABI, written against the
// reference's OWN headers.  It is not built into libadacodec (that would need the DuckDB tree); the test
// tests/test_integration_adapter.py syntax-checks it with the reference's include directories when the reference
// checkout is present, so every signature below is checked against duckdb::CompressionFunction's slot typedefs
// (src/include/duckdb/function/compression_function.hpp:65-103) and ColumnSegment's members
// (src/include/duckdb/storage/table/column_segment.hpp:40-214).
//
// What changes inside the reference when this file replaces succinct.cpp:
//   * ColumnSegment::succinct_vec (sdsl::int_vector<>) is no longer used: the bits live in HBM, the per-segment
//     handle is the CompressedSegmentState returned by the init_segment slot (SuccinctInitSegment);
//   * ColumnSegment::Compact() / Uncompact() (column_segment.cpp:273-346) call SuccinctCompactOnDevice /
//     SuccinctUncompactFromDevice instead of BitCompressFromSuccinct / UncompressSuccinct.
#include "duckdb/common/types/null_value.hpp"
#include "duckdb/common/types/vector.hpp"
#include "duckdb/function/compression/compression.hpp"
#include "duckdb/function/compression_function.hpp"
#include "duckdb/main/config.hpp"
#include "duckdb/main/database.hpp"
#include "duckdb/storage/buffer_manager.hpp"
#include "duckdb/storage/segment/uncompressed.hpp"
#include "duckdb/storage/table/append_state.hpp"
#include "duckdb/storage/table/column_segment.hpp"
#include "duckdb/storage/table/scan_state.hpp"

#include "adacodec.h" // this repository's include/

#include <cstring>
#include <mutex>
#include <vector>

namespace duckdb {

static void AdacCheck(adac_status st, const char *what) {
	if (st != ADAC_OK) {
		throw InternalException(string("adacodec: ") + what + ": " + adac_status_string(st) + " (" + adac_last_error() + ")");
	}
}

//===--------------------------------------------------------------------===//
// One GPU's pool: context, packed arena (bump allocation here; the host mirror has the first-fit free list),
// device and page-locked staging.  One instance per process is enough for the sketch; a real build keys it by
// DatabaseInstance and device.
//===--------------------------------------------------------------------===//
class SuccinctDevicePool {
public:
	static SuccinctDevicePool &Get() {
		static SuccinctDevicePool pool;
		return pool;
	}
	adac_ctx *ctx = nullptr;
	uint64_t *d_arena = nullptr;
	uint64_t arena_words = 0, arena_used = 0;
	void *d_staging = nullptr;
	idx_t staging_bytes = 0;
	std::mutex lock; // serialises the device work of this pool (one stream)

	uint64_t *AllocateArena(uint64_t words) {
		words = (words + 15) & ~uint64_t(15);
		if (arena_used + words > arena_words) {
			throw OutOfMemoryException("succinct device arena exhausted");
		}
		auto p = d_arena + arena_used;
		arena_used += words;
		return p;
	}
	void *Staging(idx_t bytes) {
		if (bytes > staging_bytes) {
			if (d_staging) {
				adac_dev_free(ctx, d_staging);
			}
			AdacCheck(adac_dev_alloc(ctx, bytes + 64, &d_staging), "adac_dev_alloc(staging)");
			staging_bytes = bytes;
		}
		return d_staging;
	}

private:
	SuccinctDevicePool() {
		AdacCheck(adac_ctx_create(0, nullptr, &ctx), "adac_ctx_create"); // ADAC_ERR_NO_DEVICE: no CPU fallback
		arena_words = (idx_t(1) << 30) / 8;
		void *p = nullptr;
		AdacCheck(adac_dev_alloc(ctx, arena_words * 8, &p), "adac_dev_alloc(arena)");
		d_arena = (uint64_t *)p;
	}
};

//===--------------------------------------------------------------------===//
// Per-segment state (slot init_segment): replaces the succinct_vec member
//===--------------------------------------------------------------------===//
struct SuccinctSegmentState : public CompressedSegmentState {
	~SuccinctSegmentState() override {
		if (layout) {
			adac_layout_destroy(layout);
		}
	}
	adac_layout *layout = nullptr; // single-segment layout (the host mirror batches many segments into one)
	uint64_t *d_words = nullptr;   // packed words in the pool arena
	adac_segment_desc desc;        // host copy: width, min, count, flags
	bool packed_on_device = false;
	std::vector<data_t> staged;    // appended rows, raw, at 8 * sizeof(T) bits per slot
	std::vector<uint64_t> validity; // one bit per staged row (all ones until a NULL arrives)
	bool any_null = false;
};

static unique_ptr<CompressedSegmentState> SuccinctInitSegment(ColumnSegment &segment, block_id_t block_id) {
	auto state = make_unique<SuccinctSegmentState>();
	state->staged.resize(segment.SegmentSize());
	return move(state);
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
// Append (succinct.cpp:264-330): rows are staged raw, no arithmetic on the host — min/max come from adac_analyze
// when the segment compacts
//===--------------------------------------------------------------------===//
static unique_ptr<CompressionAppendState> SuccinctInitAppend(ColumnSegment &segment) {
	auto &buffer_manager = BufferManager::GetBufferManager(segment.db);
	auto handle = buffer_manager.Pin(segment.block);
	return make_unique<CompressionAppendState>(move(handle));
}

template <class T>
static idx_t SuccinctAppend(CompressionAppendState &append_state, ColumnSegment &segment, SegmentStatistics &stats,
                            UnifiedVectorFormat &data, idx_t offset, idx_t count) {
	auto &state = (SuccinctSegmentState &)*segment.GetSegmentState();
	idx_t max_tuple_count = segment.SegmentSize() / sizeof(T);
	idx_t copy_count = MinValue<idx_t>(count, max_tuple_count - segment.count);
	if (state.validity.empty()) {
		state.validity.assign((max_tuple_count + 63) / 64 + 1, ~uint64_t(0));
	}
	auto sdata = (T *)data.data;
	auto tdata = (T *)state.staged.data();
	for (idx_t i = 0; i < copy_count; i++) {
		auto source_idx = data.sel->get_index(offset + i);
		auto target_idx = segment.count + i;
		if (data.validity.RowIsValid(source_idx)) {
			tdata[target_idx] = sdata[source_idx];
		} else {
			tdata[target_idx] = NullValue<T>(); // succinct.cpp:288-291
			state.validity[target_idx >> 6] &= ~(uint64_t(1) << (target_idx & 63));
			state.any_null = true;
		}
	}
	segment.count += copy_count;
	return copy_count;
}

template <class T>
static idx_t SuccinctFinalizeAppend(ColumnSegment &segment, SegmentStatistics &stats) {
	return segment.count * sizeof(T); // succinct.cpp:324-330
}

//===--------------------------------------------------------------------===//
// Compact / Uncompact bodies for ColumnSegment (were BitCompressFromSuccinct / UncompressSuccinct)
//===--------------------------------------------------------------------===//
void SuccinctCompactOnDevice(ColumnSegment &segment) {
	auto &state = (SuccinctSegmentState &)*segment.GetSegmentState();
	auto &pool = SuccinctDevicePool::Get();
	auto &config = DBConfig::GetConfig(segment.db);
	std::lock_guard<std::mutex> guard(pool.lock);
	const uint32_t count = (uint32_t)segment.count;
	const idx_t bytes = idx_t(count) * segment.type_size;
	const idx_t vbytes = (idx_t(count) + 63) / 64 * 8 + 8;
	auto d_vals = (data_ptr_t)pool.Staging(bytes + 64 + vbytes);
	auto d_valid = (uint64_t *)(d_vals + ((bytes + 63) & ~idx_t(63)));
	AdacCheck(adac_memcpy_h2d(pool.ctx, d_vals, state.staged.data(), bytes), "upload rows");
	if (state.any_null) {
		AdacCheck(adac_memcpy_h2d(pool.ctx, d_valid, state.validity.data(), vbytes - 8), "upload validity");
	}
	if (!state.layout) {
		AdacCheck(adac_layout_create(pool.ctx, (int)segment.type.InternalType(), &count, nullptr, 1, &state.layout),
		          "adac_layout_create");
	}
	state.d_words = pool.AllocateArena(adac_layout_max_arena_words(state.layout));
	AdacCheck(adac_encode(state.layout, d_vals, state.any_null ? d_valid : nullptr, ADAC_RULE_APPEND,
	                      config.succinct_padded_to_next_byte_enabled ? 1 : 0, state.d_words),
	          "adac_encode");
	AdacCheck(adac_layout_get_descs(state.layout, &state.desc), "adac_layout_get_descs");
	uint64_t minmax[2];
	AdacCheck(adac_layout_get_minmax(state.layout, minmax), "adac_layout_get_minmax");
	segment.UpdateMinFactor(minmax[0]); // keeps GetMinFactor() / GetMax() callers working
	segment.UpdateMaxFactor(minmax[1]);
	state.packed_on_device = true;
	std::vector<data_t>().swap(state.staged); // the unpacked image is gone, as after SDSL's realloc shrink
	segment.SetBitCompressed();
}

void SuccinctUncompactFromDevice(ColumnSegment &segment) {
	auto &state = (SuccinctSegmentState &)*segment.GetSegmentState();
	auto &pool = SuccinctDevicePool::Get();
	std::lock_guard<std::mutex> guard(pool.lock);
	state.staged.resize(segment.SegmentSize());
	if (segment.count) {
		const idx_t bytes = idx_t(segment.count) * segment.type_size;
		auto d_out = pool.Staging(bytes);
		AdacCheck(adac_unpack_range(state.layout, state.d_words, 0, 0, segment.count, d_out, 0), "adac_unpack_range");
		AdacCheck(adac_memcpy_d2h(pool.ctx, state.staged.data(), d_out, bytes), "adac_memcpy_d2h");
	}
	state.packed_on_device = false; // (the host mirror returns the arena block to its free list here)
	segment.SetBitUncompressed();
}

//===--------------------------------------------------------------------===//
// Scan (succinct.cpp:123-144, 232-240) and fetch (succinct.cpp:244-260, intended semantics)
//===--------------------------------------------------------------------===//
struct SuccinctScanState : public SegmentScanState {};

static unique_ptr<SegmentScanState> SuccinctInitScan(ColumnSegment &segment) {
	return make_unique<SuccinctScanState>();
}

static void SuccinctReadRows(ColumnSegment &segment, idx_t start, idx_t count, data_ptr_t target) {
	auto &state = (SuccinctSegmentState &)*segment.GetSegmentState();
	const idx_t bytes = count * segment.type_size;
	if (!state.packed_on_device) {
		memcpy(target, state.staged.data() + start * segment.type_size, bytes);
		return;
	}
	auto &pool = SuccinctDevicePool::Get();
	std::lock_guard<std::mutex> guard(pool.lock);
	auto d_out = pool.Staging(bytes);
	AdacCheck(adac_unpack_range(state.layout, state.d_words, 0, start, count, d_out, 0), "adac_unpack_range");
	AdacCheck(adac_memcpy_d2h(pool.ctx, target, d_out, bytes), "adac_memcpy_d2h");
}

template <class T>
static void SuccinctScanPartial(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result,
                                idx_t result_offset) {
	auto start = segment.GetRelativeIndex(state.row_index);
	result.SetVectorType(VectorType::FLAT_VECTOR);
	SuccinctReadRows(segment, start, scan_count, FlatVector::GetData(result) + result_offset * sizeof(T));
}

template <class T>
static void SuccinctScan(ColumnSegment &segment, ColumnScanState &state, idx_t scan_count, Vector &result) {
	SuccinctScanPartial<T>(segment, state, scan_count, result, 0);
}

template <class T>
static void SuccinctFetchRow(ColumnSegment &segment, ColumnFetchState &state, row_t row_id, Vector &result,
                             idx_t result_idx) {
	SuccinctReadRows(segment, (idx_t)row_id, 1, FlatVector::GetData(result) + result_idx * sizeof(T));
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
	return adac_type_is_supported((int)type) != 0; // duckdb::PhysicalType codes are adac_type codes
}

} // namespace duckdb
