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
//     SuccinctUncompactFromDevice instead of BitCompressFromSuccinct / UncompressSuccinct, and
//     ColumnSegmentCatalog::CompactAllSegments / CompressLowestKSegments (column_segment_catalog.cpp:56-116) hand
//     their whole list to SuccinctCompactManyOnDevice (one upload + one pack launch per pool and type).
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

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

namespace duckdb {

static void AdacCheck(adac_status st, const char *what) {
	if (st != ADAC_OK) {
		throw InternalException(string("adacodec: ") + what + ": " + adac_status_string(st) + " (" + adac_last_error() + ")");
	}
}

//===--------------------------------------------------------------------===//
// Per-GPU segment pools: context, packed arena with a first-fit free list (blocks return to it when a segment is
// expanded or destroyed), device staging.  One pool per visible gfx950 device (adac_device_count()), created on first
// use; a segment is bound to pool (creation counter mod pools) for its whole life — segments are independent units
// (src/storage/table/row_group_collection.cpp:119-155), so nothing ever crosses between pools.
// csrc/host/succinct_host.cpp is the tested implementation of the same design (plus the decoded-segment cache and
// the scan lanes); this file is its shape against DuckDB's real types.
//===--------------------------------------------------------------------===//
class SuccinctDevicePool {
public:
	explicit SuccinctDevicePool(int device_p, idx_t arena_bytes) : device(device_p) {
		AdacCheck(adac_ctx_create(device, nullptr, &ctx), "adac_ctx_create"); // ADAC_ERR_NO_DEVICE: no CPU fallback
		arena_words = (arena_bytes / 8 + 15) & ~uint64_t(15);
		void *p = nullptr;
		AdacCheck(adac_dev_alloc(ctx, arena_words * 8, &p), "adac_dev_alloc(arena)");
		d_arena = (uint64_t *)p;
		AdacCheck(adac_dev_memset(ctx, d_arena, 0, arena_words * 8), "adac_dev_memset(arena)");
		free_list[0] = arena_words;
	}
	~SuccinctDevicePool() {
		if (d_staging) {
			adac_dev_free(ctx, d_staging);
		}
		if (d_arena) {
			adac_dev_free(ctx, d_arena);
		}
		adac_ctx_destroy(ctx);
	}
	const int device;
	adac_ctx *ctx = nullptr;
	uint64_t *d_arena = nullptr;
	uint64_t arena_words = 0;
	std::mutex lock; // serialises the device work of this pool (one stream)

	//! First fit over 128-byte units; false when no block is large enough (the segment then stays unpacked)
	bool TryAllocate(uint64_t words, uint64_t &word_off) {
		words = (words + 15) & ~uint64_t(15);
		std::lock_guard<std::mutex> guard(arena_lock);
		for (auto it = free_list.begin(); it != free_list.end(); ++it) {
			if (it->second >= words) {
				uint64_t off = it->first, len = it->second;
				free_list.erase(it);
				if (len > words) {
					free_list[off + words] = len - words;
				}
				word_off = off;
				return true;
			}
		}
		return false;
	}
	void Free(uint64_t off, uint64_t words) {
		words = (words + 15) & ~uint64_t(15);
		std::lock_guard<std::mutex> guard(arena_lock);
		auto next = free_list.lower_bound(off);
		if (next != free_list.begin()) {
			auto prev = std::prev(next);
			if (prev->first + prev->second == off) {
				off = prev->first;
				words += prev->second;
				free_list.erase(prev);
			}
		}
		if (next != free_list.end() && off + words == next->first) {
			words += next->second;
			free_list.erase(next);
		}
		free_list[off] = words;
	}
	//! device scratch, grown on demand (call with `lock` held)
	void *Staging(idx_t bytes) {
		if (bytes > staging_bytes) {
			if (d_staging) {
				adac_dev_free(ctx, d_staging);
				d_staging = nullptr;
				staging_bytes = 0;
			}
			AdacCheck(adac_dev_alloc(ctx, bytes + 64, &d_staging), "adac_dev_alloc(staging)");
			staging_bytes = bytes;
		}
		return d_staging;
	}

private:
	std::mutex arena_lock;
	std::map<uint64_t, uint64_t> free_list; // word offset -> length
	void *d_staging = nullptr;
	idx_t staging_bytes = 0;
};

class SuccinctDevicePools {
public:
	//! the pools of this process: one per device, arena size from ADAC_ARENA_BYTES (default 8 GiB of the 288 GB)
	static SuccinctDevicePools &Get() {
		static SuccinctDevicePools pools;
		return pools;
	}
	SuccinctDevicePool &Next() {
		return *pools[counter.fetch_add(1) % pools.size()];
	}
	std::vector<unique_ptr<SuccinctDevicePool>> pools;

private:
	SuccinctDevicePools() {
		int n = adac_device_count();
		if (n < 1) {
			throw InternalException("adacodec: no gfx950 device (there is no CPU fallback)");
		}
		idx_t arena_bytes = idx_t(8) << 30;
		if (const char *env = std::getenv("ADAC_ARENA_BYTES")) {
			arena_bytes = (idx_t)std::strtoull(env, nullptr, 10);
		}
		for (int d = 0; d < n; d++) {
			pools.push_back(unique_ptr<SuccinctDevicePool>(new SuccinctDevicePool(d, arena_bytes)));
		}
	}
	std::atomic<uint64_t> counter {0};
};

//===--------------------------------------------------------------------===//
// Per-segment state (slot init_segment): replaces the succinct_vec member.  No device object per segment: the packed
// form is a block of the pool's arena + the 32-byte descriptor; reads are layout-free jobs (adac_unpack_jobs).
//===--------------------------------------------------------------------===//
struct SuccinctSegmentState : public CompressedSegmentState {
	SuccinctSegmentState() : pool(SuccinctDevicePools::Get().Next()) {
	}
	~SuccinctSegmentState() override {
		if (packed_on_device) {
			pool.Free(desc.word_off, arena_words); // the block goes back to the pool with the segment
		}
	}
	SuccinctDevicePool &pool;
	adac_segment_desc desc;        // word_off into the pool arena, count, width, min, flags
	uint64_t arena_words = 0;
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
	// the running min / max of SuccinctAppendLoop (succinct.cpp:286-299): uint64_t(T x), NULL rows excluded.  They
	// ride along with the copy the slot has to make anyway, so Compact() needs no analyze pass over the rows
	uint64_t min = segment.GetMinFactor(), max = segment.GetMax();
	for (idx_t i = 0; i < copy_count; i++) {
		auto source_idx = data.sel->get_index(offset + i);
		auto target_idx = segment.count + i;
		if (data.validity.RowIsValid(source_idx)) {
			tdata[target_idx] = sdata[source_idx];
			uint64_t v = uint64_t(sdata[source_idx]);
			min = MinValue<uint64_t>(min, v);
			max = MaxValue<uint64_t>(max, v);
		} else {
			tdata[target_idx] = NullValue<T>(); // succinct.cpp:288-291
			state.validity[target_idx >> 6] &= ~(uint64_t(1) << (target_idx & 63));
			state.any_null = true;
		}
	}
	segment.UpdateMinFactor(min); // succinct.cpp:317-318
	segment.UpdateMaxFactor(max);
	segment.count += copy_count;
	return copy_count;
}

template <class T>
static idx_t SuccinctFinalizeAppend(ColumnSegment &segment, SegmentStatistics &stats) {
	return segment.count * sizeof(T); // succinct.cpp:324-330
}

//===--------------------------------------------------------------------===//
// Compact / Uncompact bodies for ColumnSegment (were BitCompressFromSuccinct / UncompressSuccinct).
// A2: the append slot carried min / max, so compaction is the width decision on the host (adac_width,
// column_segment.cpp:351-363) + ONE pack pass on the device; many segments of a pool go through one upload and one
// launch (SuccinctCompactManyOnDevice: what CompactAllSegments and a policy round call).
//===--------------------------------------------------------------------===//
void SuccinctCompactManyOnDevice(const std::vector<ColumnSegment *> &segments) {
	std::map<std::pair<SuccinctDevicePool *, int>, std::vector<ColumnSegment *>> groups; // (pool, physical type)
	for (auto segment : segments) {
		auto &state = (SuccinctSegmentState &)*segment->GetSegmentState();
		if (!state.packed_on_device && segment->count > 0) {
			groups[std::make_pair(&state.pool, (int)segment->type.InternalType())].push_back(segment);
		}
	}
	for (auto &group : groups) {
		auto &pool = *group.first.first;
		const int ptype = group.first.second;
		const idx_t ts = adac_type_size(ptype);
		const idx_t per16 = 16 / ts;
		auto &config = DBConfig::GetConfig(group.second[0]->db);
		std::vector<uint32_t> counts;
		std::vector<uint64_t> offs;
		std::vector<adac_segment_desc> descs;
		std::vector<ColumnSegment *> packed;
		uint64_t span = 0;
		bool any_null = false;
		for (auto segment : group.second) {
			auto &state = (SuccinctSegmentState &)*segment->GetSegmentState();
			const uint64_t mn = segment->GetMinFactor(), mx = segment->GetMax();
			const uint8_t w = adac_width(mn, mx, ADAC_RULE_APPEND, config.succinct_padded_to_next_byte_enabled ? 1 : 0);
			if (8 * ts <= w) { // `if (old_width > min_width)` fails (column_segment.cpp:363): the slots stay as they are
				segment->SetBitCompressed();
				continue;
			}
			adac_segment_desc d;
			const uint64_t need = adac_arena_words(segment->count, w);
			if (!pool.TryAllocate(need, d.word_off)) {
				continue; // arena full: the segment keeps its unpacked form (tried again by the next round)
			}
			d.val_off = span;
			d.min = adac_stored_min(mn, mx, w);
			d.count = (uint32_t)segment->count;
			d.width = w;
			d.flags = ADAC_SEG_PACKED;
			d.reserved = 0;
			state.arena_words = need;
			descs.push_back(d);
			counts.push_back(d.count);
			offs.push_back(span);
			packed.push_back(segment);
			span += (segment->count + per16 - 1) / per16 * per16; // every segment 16-byte aligned in the staging
			any_null |= state.any_null;
		}
		if (packed.empty()) {
			continue;
		}
		std::lock_guard<std::mutex> guard(pool.lock);
		const idx_t bytes = span * ts + 16;
		const idx_t vbytes = (span / 64 + 2) * 8;
		auto d_vals = (data_ptr_t)pool.Staging(bytes + 64 + vbytes);
		auto d_valid = (uint64_t *)(d_vals + ((bytes + 63) & ~idx_t(63)));
		std::vector<uint64_t> vmask;
		if (any_null) {
			vmask.assign(vbytes / 8, ~uint64_t(0));
		}
		adac_layout *layout = nullptr;
		try {
			for (idx_t i = 0; i < packed.size(); i++) {
				auto &state = (SuccinctSegmentState &)*packed[i]->GetSegmentState();
				AdacCheck(adac_memcpy_h2d(pool.ctx, d_vals + offs[i] * ts, state.staged.data(), idx_t(counts[i]) * ts),
				          "upload rows");
				for (idx_t r = 0; state.any_null && r < counts[i]; r++) {
					if (!((state.validity[r >> 6] >> (r & 63)) & 1)) {
						vmask[(offs[i] + r) >> 6] &= ~(uint64_t(1) << ((offs[i] + r) & 63));
					}
				}
			}
			if (any_null) {
				AdacCheck(adac_memcpy_h2d(pool.ctx, d_valid, vmask.data(), vbytes), "upload validity");
			}
			AdacCheck(adac_layout_create(pool.ctx, ptype, counts.data(), offs.data(), packed.size(), &layout),
			          "adac_layout_create");
			AdacCheck(adac_layout_set_descs(layout, descs.data()), "adac_layout_set_descs");
			AdacCheck(adac_pack(layout, d_vals, any_null ? d_valid : nullptr, pool.d_arena), "adac_pack");
			AdacCheck(adac_ctx_sync(pool.ctx), "adac_ctx_sync");
		} catch (...) {
			if (layout) {
				adac_layout_destroy(layout);
			}
			for (auto &d : descs) {
				pool.Free(d.word_off, adac_arena_words(d.count, d.width)); // nothing of the batch keeps its block
			}
			throw;
		}
		adac_layout_destroy(layout);
		for (idx_t i = 0; i < packed.size(); i++) {
			auto &state = (SuccinctSegmentState &)*packed[i]->GetSegmentState();
			state.desc = descs[i];
			state.desc.val_off = 0;
			state.packed_on_device = true;
			std::vector<data_t>().swap(state.staged); // the unpacked image is gone, as after SDSL's realloc shrink
			packed[i]->SetBitCompressed();
		}
	}
}

void SuccinctCompactOnDevice(ColumnSegment &segment) {
	SuccinctCompactManyOnDevice(std::vector<ColumnSegment *> {&segment});
}

//! rows [start, start + count) of a packed segment into `target` (host memory)
static void SuccinctDecodeRows(SuccinctSegmentState &state, PhysicalType type, idx_t type_size, idx_t start, idx_t count,
                               data_ptr_t target) {
	auto &pool = state.pool;
	std::lock_guard<std::mutex> guard(pool.lock);
	const idx_t bytes = count * type_size;
	auto d_out = pool.Staging(bytes);
	adac_unpack_job job;
	memset(&job, 0, sizeof(job));
	job.word_off = state.desc.word_off;
	job.min = state.desc.min;
	job.out_off = 0;
	job.start = (uint32_t)start;
	job.count = (uint32_t)count;
	job.width = state.desc.width;
	job.flags = state.desc.flags;
	AdacCheck(adac_unpack_jobs(pool.ctx, (int)type, &job, 1, pool.d_arena, d_out), "adac_unpack_jobs");
	AdacCheck(adac_memcpy_d2h(pool.ctx, target, d_out, bytes), "adac_memcpy_d2h");
}

void SuccinctUncompactFromDevice(ColumnSegment &segment) {
	auto &state = (SuccinctSegmentState &)*segment.GetSegmentState();
	if (state.packed_on_device) {
		state.staged.resize(segment.SegmentSize());
		if (segment.count) {
			SuccinctDecodeRows(state, segment.type.InternalType(), segment.type_size, 0, segment.count, state.staged.data());
		}
		state.pool.Free(state.desc.word_off, state.arena_words); // the block returns to the pool's free list
		state.packed_on_device = false;
	}
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
	if (!state.packed_on_device) {
		memcpy(target, state.staged.data() + start * segment.type_size, count * segment.type_size);
		return;
	}
	// (the host mirror serves this call from its decoded-segment cache: see csrc/host/succinct_host.cpp, PinDecoded)
	SuccinctDecodeRows(state, segment.type.InternalType(), segment.type_size, start, count, target);
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
