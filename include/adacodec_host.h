/*
 * adacodec_host.h — C wrappers around the C++ host mirror (csrc/host/succinct_host.hpp) of the reference's
 * ColumnSegment / CompressionFunction / ColumnSegmentCatalog plumbing.  They exist so hosts without a C++
 * toolchain — and the Python tests — can drive the same state machine the DuckDB adapter would
 * (INTEGRATION.md).  All value arithmetic happens on the device through include/adacodec.h; functions
 * returning int return 0 on success and 1 on failure (text in adach_last_error()).
 *
 * Reference interfaces mirrored (paths relative to the reference checkout):
 *   adach_db_create               DBConfig flags (src/include/duckdb/main/config.hpp:189-197) + one GPU segment pool
 *   adach_db_create_pools         the same with one segment pool per listed device (per-GPU segment pools; a segment
 *                                 lives in pool id mod pools — the unit of independence is the segment,
 *                                 src/storage/table/row_group_collection.cpp:119-155)
 *   adach_segment_create          ColumnSegment::CreateTransientSegment (src/storage/table/column_segment.cpp:45-82)
 *   adach_segment_append          ColumnSegment::Append (column_segment.cpp:247-271) -> append slot (succinct.cpp:308-322)
 *   adach_segment_scan            ColumnSegment::Scan / ScanPartial (column_segment.cpp:137-188) -> scan_vector / scan_partial
 *   adach_segment_fetch_row       ColumnSegment::FetchRow (column_segment.cpp:193-195) -> fetch_row slot
 *   adach_segment_compact/uncompact   ColumnSegment::Compact / Uncompact (column_segment.cpp:273-346)
 *   adach_segment_data_size       ColumnSegment::GetDataSize (column_segment.cpp:204-214)
 *   adach_catalog_*               ColumnSegmentCatalog (src/catalog/catalog_entry/column_segment_catalog.cpp:24-135)
 */
#ifndef ADACODEC_HOST_H
#define ADACODEC_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct adach_db adach_db;
typedef struct adach_segment adach_segment;

const char *adach_last_error(void);
int adach_type_is_supported(int physical_type);

/* arena_bytes: capacity of this GPU's packed-segment arena. NULL on failure (e.g. no device). */
adach_db *adach_db_create(int device, int succinct_enabled, int adaptive, int padded, uint64_t arena_bytes);
/* same, with `decoded_cache_bytes` of page-locked host memory holding whole decoded segments so that the
 * engine's 2048-row scan_vector calls cost one device decode + one PCIe copy per segment (0 = off) */
adach_db *adach_db_create_cached(int device, int succinct_enabled, int adaptive, int padded, uint64_t arena_bytes,
                                 uint64_t decoded_cache_bytes);
/* One segment pool per entry of devices[] (the same device may be listed more than once).  arena_bytes and
 * decoded_cache_bytes are PER POOL.  scan_lanes: decode + copy-down streams per pool (0 = 4); prefetch_segments:
 * segments per decode batch, i.e. how far a sequential scan decodes ahead of its consumer (0 = 8, at most 48). */
adach_db *adach_db_create_pools(const int *devices, int npools, int succinct_enabled, int adaptive, int padded,
                                uint64_t arena_bytes, uint64_t decoded_cache_bytes, uint32_t scan_lanes,
                                uint32_t prefetch_segments);
uint64_t adach_db_num_pools(adach_db *db);
void adach_db_cache_stats(adach_db *db, uint64_t *hits, uint64_t *misses, uint64_t *bytes);
/* decode batches enqueued, segments decoded ahead of their first touch, segments a compaction left unpacked because
 * the arena was full (summed over the pools) */
void adach_db_prefetch_stats(adach_db *db, uint64_t *batches, uint64_t *prefetched, uint64_t *exhausted);
/* adach_full_scan with `threads` consumers, each scanning one contiguous run of the segments (the row-group morsels
 * of a parallel table scan, src/storage/table/row_group_collection.cpp:119-155) */
int adach_full_scan_mt(adach_segment **segs, uint64_t nseg, uint64_t vector_size, uint32_t threads, uint64_t *checksum,
                       double *seconds, uint64_t *rows);
/* Full scan of segs[0..nseg) the way ColumnData::ScanVector drives the codec (vector_size rows per
 * ColumnSegment::Scan call), timed on the host; checksum = wrapping sum of every scanned row. */
int adach_full_scan(adach_segment **segs, uint64_t nseg, uint64_t vector_size, uint64_t *checksum, double *seconds,
                    uint64_t *rows);
/* Pre-size the page-locked upload buffer and its device twin (both otherwise grow on demand inside the first
 * batched Compact: page-locking ~100 MB costs ~20 ms once). */
int adach_db_reserve_staging(adach_db *db, uint64_t bytes);
void adach_db_destroy(adach_db *db);
int64_t adach_db_data_size(adach_db *db);          /* BufferManager::GetDataSize analogue */
uint64_t adach_db_arena_used_bytes(adach_db *db);  /* HBM actually held by packed segments */

/* The checkpoint-side half of the plugin table (compression_function.hpp:65-103), driven the way
 * ColumnDataCheckpointer does: init_analyze / analyze (every 2048-row vector) / final_analyze -> *out_score, then
 * init_compression / compress (every vector) / compress_finalize (succinct.cpp:52-119: CreateEmptySegment,
 * Append until full, FlushSegment).  compression_type: 10 = SUCCINCT, 1 = UNCOMPRESSED.  The flushed segments are
 * returned in order (caller destroys them); out_sizes[i] = bytes FinalizeAppend reported for segment i. */
int adach_compress_column(adach_db *db, int compression_type, int physical_type, uint64_t row_group_start,
                          const void *values, const uint64_t *validity, uint64_t n, adach_segment **out_segments,
                          uint64_t max_segments, uint64_t *out_nseg, uint64_t *out_sizes, uint64_t *out_score);
/* The same, also returning what ColumnCheckpointState::FlushSegment -> ColumnSegment::ConvertToPersistent yields for
 * every flushed segment — the conversion the reference leaves empty for SUCCINCT (src/storage/table/
 * column_segment.cpp:529-533): block images (include/adacodec.h) back to back at 8-byte aligned offsets,
 * out_block_offs[0 .. nseg] (image i = [offs[i], offs[i] + adac_block_bytes)).  All images of the column are built
 * in ONE device pass per pool.  out_blocks / out_block_offs may be NULL. */
int adach_checkpoint_column(adach_db *db, int compression_type, int physical_type, uint64_t row_group_start,
                            const void *values, const uint64_t *validity, uint64_t n, adach_segment **out_segments,
                            uint64_t max_segments, uint64_t *out_nseg, uint64_t *out_sizes, uint64_t *out_score,
                            void *out_blocks, uint64_t blocks_cap, uint64_t *out_block_offs);
/* ConvertToPersistent for any list of segments (compacting those that still need it): images as above. */
uint64_t adach_segment_block_bytes(adach_segment *seg); /* upper bound of the segment's image size */
int adach_segments_persist(adach_db *db, adach_segment **segs, uint64_t nseg, void *out, uint64_t cap,
                           uint64_t *out_offs /* nseg + 1 */);
/* ColumnSegment::CreatePersistentSegment for a batch (src/storage/table/column_segment.cpp:25-43): segments whose
 * packed words are moved into the pools' arenas with one upload and one device pass per pool.  lens[i]:
 * adac_block_bytes of image i; starts[i]: its first row. */
int adach_segments_load(adach_db *db, const void *blocks, const uint64_t *offs, const uint64_t *lens,
                        const uint64_t *starts, uint64_t nseg, adach_segment **out_segments);

/* present[0..15] = which of the sixteen slots the function table fills, in the reference's order (init_analyze,
 * analyze, final_analyze, init_compression, compress, compress_finalize, init_scan, scan_vector, scan_partial,
 * fetch_row, skip, init_segment, init_append, append, finalize_append, revert_append): succinct.cpp:335-343 leaves
 * init_segment and revert_append null. */
int adach_function_slots(adach_db *db, int compression_type, int physical_type, int *present);

adach_segment *adach_segment_create(adach_db *db, int physical_type, uint64_t start, uint64_t segment_size);
/* SegmentBase::next as a hint: with the decoded-segment cache on, the first scan of `seg` also starts decoding and
 * copying `next` on the pool's stream, so a sequential scan overlaps PCIe with the consumer.  NULL unlinks. */
int adach_segment_set_next(adach_segment *seg, adach_segment *next);
void adach_segment_destroy(adach_segment *seg);
/* returns rows consumed (the caller opens a new segment for the rest), -1 on error */
int64_t adach_segment_append(adach_segment *seg, const void *vals, const uint64_t *validity, const uint32_t *sel,
                             uint64_t offset, uint64_t count);
int adach_segment_scan(adach_segment *seg, uint64_t row_index, uint64_t count, void *result, uint64_t result_offset,
                       int entire_vector);
int adach_segment_fetch_row(adach_segment *seg, int64_t row_id, void *result, uint64_t result_idx);

/* The engine's scan state (duckdb::ColumnScanState, src/include/duckdb/storage/table/scan_state.hpp) as an object that
 * lives across scan_vector calls: adach_segment_init_scan is ColumnSegment::InitializeScan (column_segment.cpp:133-135
 * -> init_scan slot) — ColumnData::ScanVector calls it whenever it moves on to the next segment
 * (src/storage/table/column_data.cpp:92-139) — and the state then PINS the segment's decoded block in the page-locked
 * cache on its first scan, so every following 2048-row call is a memcpy.  adach_segment_scan (above) builds a fresh
 * state per call and is the uncached shape; integration/succinct_gpu.cpp uses the calls below.  One state per
 * scanning thread; destroying it (or initialising it for another segment) releases the pin. */
typedef struct adach_scan_state adach_scan_state;
adach_scan_state *adach_scan_state_create(void);
void adach_scan_state_destroy(adach_scan_state *state);
int adach_segment_init_scan(adach_segment *seg, adach_scan_state *state);
int adach_segment_scan_with(adach_segment *seg, adach_scan_state *state, uint64_t row_index, uint64_t count,
                            void *result, uint64_t result_offset, int entire_vector);
/* ColumnSegment::Compact for a list of segments of any pools: one upload, one analyze, one pack per (pool, type,
 * rule) — what CompactAllSegments and a policy round of the engine's own catalog hand over
 * (src/catalog/catalog_entry/column_segment_catalog.cpp:56-116) */
int adach_segments_compact(adach_db *db, adach_segment **segs, uint64_t nseg);
int adach_segment_compact(adach_segment *seg);
int adach_segment_uncompact(adach_segment *seg);
uint64_t adach_segment_count(adach_segment *seg);
uint64_t adach_segment_min(adach_segment *seg);
uint64_t adach_segment_max(adach_segment *seg);
uint32_t adach_segment_width(adach_segment *seg);
int adach_segment_compacted(adach_segment *seg);
int adach_segment_function(adach_segment *seg); /* CompressionType: 1 uncompressed, 10 succinct */
uint64_t adach_segment_data_size(adach_segment *seg);
int adach_segment_pool(adach_segment *seg);       /* index of the pool (GPU) the segment lives in */
int adach_segment_persistent(adach_segment *seg); /* ColumnSegmentType::PERSISTENT */

int adach_catalog_compact_all(adach_db *db);
uint64_t adach_catalog_total_data_size(adach_db *db);
uint64_t adach_catalog_num_segments(adach_db *db);
/* one round of CompressLowestKSegments without the sleep */
int adach_catalog_policy_step(adach_db *db, double compression_rate);
int adach_catalog_enable_background(adach_db *db, unsigned period_ms);
int adach_catalog_disable_background(adach_db *db);
/* rounds the background thread has run, rounds that failed (the thread records the error and carries on), and the
 * last error text */
void adach_catalog_background_stats(adach_db *db, uint64_t *rounds, uint64_t *errors, char *last_error, uint64_t cap);
uint64_t adach_db_pool_arena_used_bytes(adach_db *db, uint32_t pool);

#ifdef __cplusplus
}
#endif
#endif /* ADACODEC_HOST_H */
