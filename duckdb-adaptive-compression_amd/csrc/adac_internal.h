// adac_internal.h — launch wrappers shared between the C ABI (adac_capi.cpp) and the gfx950 kernels
// (adac_kernels.hip).  Not installed; the public surface is include/adacodec.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "adacodec.h"

namespace adac {

constexpr int kWorkgroup = 256;       // 4 wavefronts of 64 lanes
constexpr int kTileBytes = 16384;     // decoded bytes per tile: tile = 16 KiB / sizeof(T) values
constexpr uint32_t kNoTile = 0xffffffffu;

// One tile of one segment: `first` is a multiple of the type's tile size.
struct TileRef {
	uint32_t seg;
	uint32_t first;
};

// A tile with its segment's CURRENT descriptor folded in (rebuilt whenever descriptors change, like the scan groups): the
// gather (k_gather_c) reads ONE 32-byte record before its first data load instead of the dependent chain tile entry ->
// descriptor — a workgroup's lifetime is a handful of memory round trips, and the bytes a CU keeps in flight per lifetime
// bound that kernel (+4 - 7 %).  The decode (k_unpack) does NOT take it: measured 4 - 17 % slower with the record, same
// box, interleaved (profiles/r03_tile_records.json) — as with the scalar descriptor load of round 3's first session.
struct alignas(32) TileRec {
	uint64_t word_off; // the segment's first word in the arena
	uint64_t elem0;    // element index of the tile's first row
	uint64_t add;      // frame of reference to add back (0 if none)
	uint32_t first;    // first row of the tile inside the segment
	uint16_t n;        // rows in the tile (<= 16384)
	uint8_t width, flags;
};
static_assert(sizeof(TileRec) == 32, "device record");

// Work item of the fused scans: `ntiles` consecutive tiles of ONE segment starting at row `first` (static per
// layout and grouping), and its expansion with the segment's current descriptor (rebuilt whenever descriptors
// change): a scan workgroup then needs ONE 64-byte load before its first data load instead of the dependent
// chain tile entry -> descriptor.
// The ARRIVAL fields (round 3) let a scan finish its results inside the one kernel — no clearing pass before it and no
// merge kernel after it: `seg_groups` groups add into the segment's result cell and the last one to arrive stores the
// total; a bitmap word that several groups cover in part (dense value spaces only) is ORed into an edge cell by
// `n_first` / `n_last` groups and stored by the last of them.  Cells are zero between calls (the last arriver resets).
struct ScanGroupRef {
	uint32_t seg;
	uint32_t first;
	uint32_t rows; // a segment's tiles are dealt evenly to its groups, so group sizes differ between segments
	uint32_t seg_groups; // scan groups of this segment
	uint32_t cell_first; // edge cell of the group's first bitmap word when that word is covered in part
	uint32_t cell_last;  // ... of its last word (groups of more than one word)
	uint16_t n_first;    // groups that cover the first word in part (0: the word is whole, or the value space has gaps)
	uint16_t n_last;
	uint32_t pad;
};
static_assert(sizeof(ScanGroupRef) == 32, "device record");
struct alignas(64) ScanGroup {
	adac_segment_desc d;
	uint32_t seg;
	uint32_t first;
	uint32_t n; // rows
	uint32_t seg_groups;
	uint32_t cell_first, cell_last;
	uint16_t n_first, n_last;
	uint32_t pad;
};
static_assert(sizeof(ScanGroup) == 64, "one cache-line record per scan group");

// Single-segment range decode (scan_vector / scan_partial): tiles are implicit.
struct RangeArgs {
	uint32_t seg;
	uint32_t start;
	uint32_t count;
	uint64_t out_off;
};

// Layout-free range decode of several segments in ONE launch (adac_unpack_jobs): the jobs travel by value in the
// kernel-argument segment, so a batch needs no device table, no upload and no allocation — what a host that serves an
// engine vector by vector (or prefetches the next few segments of a scan) can afford per call.
constexpr int kMaxUnpackJobs = 48;
struct UnpackJob {
	uint64_t word_off; // first word of the segment in the arena
	uint64_t add;      // frame of reference to add back (0 if none: unpacked slots, or no min)
	uint64_t out_off;  // element offset of the job's first decoded row in the output
	uint32_t start;    // first row of the segment to decode
	uint32_t count;    // rows to decode
	uint32_t width;
	uint32_t tile0;    // first workgroup of this job in the launch
};
struct UnpackJobTable {
	uint32_t njobs, ntiles;
	UnpackJob jobs[kMaxUnpackJobs];
};
hipError_t launch_unpack_jobs(hipStream_t s, uint32_t type_size, const UnpackJobTable &table, const uint64_t *d_words,
                              void *d_out);

// Launch-shape knobs (adac_set_tuning): which kernel form the scan entry points use and how many persistent
// workgroups are launched.  Defaults are the measured-best settings on MI355X.
struct Tuning {
	int persistent_unpack = 0;  // measured 5-15 % slower than one tile per workgroup (profiles/r01_ab_*.json)
	int single_pass_encode = 1; // A/B: 0 = analyze + plan + pack as three kernels (the raw column is read twice)
	int group_sum_wide = 0;     // A/B: adac_scan_group_sum always in 64-bit arithmetic
	int group_sum_rw = 1;       // A/B: 0 = adac_scan_group_sum without the register-walk kernel (k_group_sum only)
	int encode_placement = 0;   // single-pass encode: 0 = arena order is segment order (look-back), 1 = order of completion
	int encode_big_image = 1;   // single-pass encode, ordered placement: a segment whose packed words fit the LDS pool is packed there whole and publishes the NEXT footprint before it waits (A/B: 0)
	int encode_publish_ahead = 1; // single-pass encode, ordered placement: the parked flow publishes the NEXT footprint before it waits (A/B: 0)
	int encode_stamps = 0;      // diagnostic: phase time stamps of the single-pass encode (adac_debug_encode_stamps)
	int grouped_repack = 1;     // A/B: 0 = one tile per workgroup with the 16 KiB row image (the first version)
	int tile_records = 1;       // expanded 32-byte tile records (TileRec): bit 0 = the gather (k_gather_c: +4 - 7 %), bit 1 = the decode (k_unpack: 4 - 17 % SLOWER, off)
	int scan_cells = 1;         // fused scans: 1 = results (and shared bitmap words) finished inside the scan kernel through arrival cells, 0 = clearing pass + atomics (+ merge kernel)
	int gather_compact = 3;     // adac_unpack_selected: 0 = a store per selected row (k_gather), 1 = wave-level compaction + dense stores (k_gather_c), 3 = the same with non-temporal stores
	int sel_debug = 0;          // diagnostic: selection scan without its flush (1) / without any bitmap emit (2)
	int scan_probe = 0;         // diagnostic: fused-scan loop + loads only (no field walk)
	int templated_scan = 1;     // width-templated register path of the fused scans for 4 <= w <= 32
	int scan_tiles_per_wg = 0; // tiles per fused-scan workgroup; 0 = by type (12 tiles of u64, 6 of u32, 8 of u16, 4 of u8)
	int num_cus = 0;        // 0 = the device's own count (256 on a whole MI355X: 8 XCDs x 32 CUs); > 0 overrides
	int blocks_per_cu = 8;  // 256-thread workgroups resident per CU (2048 threads, <= 16.5 KiB LDS each)
};
extern Tuning g_tuning;

inline uint32_t tile_values(uint32_t type_size) {
	return kTileBytes / type_size;
}

// type_size in {1,2,4,8}; every launcher returns the hipError_t of the launch.
hipError_t launch_analyze(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                          const adac_segment_desc *d_descs, const TileRef *d_tiles, uint64_t ntiles,
                          const void *d_vals, const uint64_t *d_validity, uint64_t *d_minmax);
hipError_t launch_minmax_init(hipStream_t s, uint64_t *d_minmax, uint64_t nseg);
hipError_t launch_plan(hipStream_t s, uint32_t type_size, int rule, int pad_to_byte, adac_segment_desc *d_descs,
                       const uint64_t *d_minmax, uint64_t nseg);
hipError_t launch_pack(hipStream_t s, uint32_t type_size, uint64_t null_bits, const adac_segment_desc *d_descs,
                       const TileRef *d_tiles, uint64_t ntiles, const void *d_vals, const uint64_t *d_validity,
                       uint64_t *d_words);
hipError_t launch_unpack(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const TileRef *d_tiles,
                         uint64_t ntiles, const uint64_t *d_words, void *d_out, const TileRec *d_recs);
hipError_t launch_unpack_range(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, RangeArgs range,
                               const uint64_t *d_words, void *d_out);
hipError_t launch_fetch(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const uint64_t *d_words,
                        const uint32_t *d_segs, const uint32_t *d_rows, uint64_t n, void *d_out);
hipError_t launch_analyze_packed(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                                 const adac_segment_desc *d_src_descs, const TileRef *d_tiles, uint64_t ntiles,
                                 const uint64_t *d_src_words, const uint64_t *d_validity, uint64_t *d_minmax);
hipError_t launch_repack(hipStream_t s, uint32_t type_size, uint64_t null_bits, const adac_segment_desc *d_src_descs,
                         const adac_segment_desc *d_dst_descs, const TileRef *d_tiles, uint64_t ntiles,
                         const uint64_t *d_src_words, const uint64_t *d_validity, uint64_t *d_dst_words);
// single-pass encode (adac_encode_1p.inl): every segment must fit sixteen 16-byte chunks per thread of a 1024-thread
// workgroup, counted from the 16-byte boundary at or before its first element
constexpr uint64_t kEncodeOnePassBytes = 16ull * 1024 * 16;
// 64-bit words of device scratch the single-pass encode needs for nseg segments (look-back words, ticket, cursor)
inline uint64_t encode_1p_state_words(uint64_t nseg) { return nseg + 2; }
hipError_t read_encode_stamps(void *host, uint64_t bytes);
// grouped SUM / COUNT over two packed columns of one table (adac_group_sum.inl)
uint64_t group_sum_partial_bytes();
uint32_t group_sum_max_groups();
hipError_t launch_group_sum(hipStream_t s, uint32_t v_type_size, bool v_signed, uint32_t k_type_size,
                            const adac_segment_desc *d_vdescs, const TileRef *d_vtiles, uint64_t ntiles,
                            const ScanGroup *d_vgroups, uint64_t nvgroups, const uint64_t *d_vwords,
                            const adac_segment_desc *d_kdescs, const uint64_t *d_kwords, uint32_t ngroups, void *d_partial,
                            uint32_t call_parity, uint64_t *d_sums, uint64_t *d_counts);
hipError_t launch_encode_1p(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                            int pad_to_byte, adac_segment_desc *d_descs, uint64_t nseg, const void *d_vals,
                            const uint64_t *d_validity, uint64_t *d_minmax, void *d_scan_state, uint64_t *d_words);
hipError_t launch_analyze_packed_g(hipStream_t s, uint32_t type_size, bool sign_extend, uint64_t null_bits, int rule,
                                   const ScanGroup *d_src_groups, uint64_t ngroups, const uint64_t *d_src_words,
                                   const uint64_t *d_validity, uint64_t *d_minmax);
hipError_t launch_repack_g(hipStream_t s, uint32_t type_size, uint64_t null_bits, const ScanGroup *d_src_groups,
                           uint64_t ngroups, const adac_segment_desc *d_dst_descs, const uint64_t *d_src_words,
                           const uint64_t *d_validity, uint64_t *d_dst_words);
// The fused scans' work items as the launchers see them: every group, and the indices of those at widths 2 and 3,
// which a kernel of their own scans (n_narrow is read back once per expansion).
struct ScanGroupList {
	const ScanGroup *d_groups;
	uint64_t ngroups;
	const uint32_t *d_narrow_idx;
	uint32_t n_narrow;
	// arrival cells (see ScanGroupRef), zero between calls; nullptr: the clearing pass + atomics form
	unsigned long long *d_res_cells; // per segment: four words (arrive_sum / arrive_count)
	uint32_t *d_edge_cells;          // two per group: one 64-bit word each (arrive_or)
};
inline uint64_t scan_res_cell_bytes(uint64_t nseg) { return (nseg ? nseg : 1) * 4 * sizeof(unsigned long long); }
inline uint64_t scan_edge_cell_bytes(uint64_t ngroups) { return (ngroups ? ngroups : 1) * 4 * sizeof(uint32_t); }
hipError_t launch_expand_tiles(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs, const TileRef *d_tiles,
                               uint64_t ntiles, TileRec *d_recs);
hipError_t launch_expand_groups(hipStream_t s, const adac_segment_desc *d_descs, const ScanGroupRef *d_refs,
                                uint64_t ngroups, ScanGroup *d_groups, uint32_t *d_narrow_idx, uint32_t *d_narrow_count);
hipError_t launch_gather_selected(hipStream_t s, uint32_t type_size, const adac_segment_desc *d_descs,
                                  const TileRef *d_tiles, uint64_t ntiles, const uint64_t *d_words,
                                  const TileRec *d_recs, const uint64_t *d_bitmap, uint64_t bitmap_words, uint32_t *d_tile_cnt,
                                  uint64_t *d_tile_off, uint64_t *d_block_tot, void *d_out, uint64_t *d_out_ids,
                                  uint64_t *d_total);
hipError_t launch_scan_sum(hipStream_t s, uint32_t type_size, const ScanGroupList &gl, const uint64_t *d_words,
                           const uint64_t *d_validity, uint64_t sbit, uint64_t *d_sums);
uint64_t sel_edge_bytes(uint64_t ngroups);
hipError_t launch_sel_merge_edges(hipStream_t s, const void *d_edges, uint64_t ngroups, uint64_t *d_bitmap,
                                  uint64_t tail_word);
hipError_t launch_scan_count_range(hipStream_t s, uint32_t type_size, const ScanGroupList &gl, const uint64_t *d_words,
                                   const uint64_t *d_validity, uint64_t blo, uint64_t bspan, uint64_t sbit,
                                   uint64_t *d_counts, uint64_t *d_bitmap, void *d_edges, bool edge_cells,
                                   uint64_t tail_word);

// Persistent block images (adac_block_image.inl): one segment's packed words <-> its image in a block buffer.
struct BlockJob {
	uint64_t word_off;   // first word of the segment in the packed arena
	uint64_t block_off;  // byte offset of its image in the block buffer, multiple of 8
	uint64_t min;
	uint64_t bit_size;   // count * width: the int_vector's m_size
	uint32_t nwords;     // ceil(bit_size / 64)
	uint32_t arena_words; // words the segment owns in the arena (parse: the tail past nwords is zeroed)
	uint8_t width, flags, type, pad[5];
};
static_assert(sizeof(BlockJob) == 48, "device record");
hipError_t launch_blocks_write(hipStream_t s, const BlockJob *d_jobs, uint64_t njobs, uint32_t max_units,
                               const uint64_t *d_words, void *d_blocks);
hipError_t launch_blocks_read(hipStream_t s, const BlockJob *d_jobs, uint64_t njobs, uint32_t max_units,
                              const void *d_blocks, uint64_t *d_words, uint32_t *d_bad);

// DuckDB BITPACKING segments (adac_bitpacking.inl).  Host view of one metadata group; must match BpGroup.
struct BpGroupHost {
	uint64_t block_off;
	uint64_t out_off;
	uint32_t group;
	uint32_t rows;
	uint64_t payload_off, frame, extra; // parsed header, filled on the device by launch_bp_prepare
	uint32_t mode, width;
};
// compress side: per-group statistics (device -> host) and write records (host -> device); must match
// BpStats / BpWrite in adac_bitpacking.inl
struct BpStatsHost {
	uint64_t bmin, bmax, bdmin, bdmax, v0;
	uint32_t rows, nvalid, delta_overflow, pad;
};
struct BpWriteHost {
	uint64_t frame, extra, first;
	uint32_t seg, data_off, meta_off, mode, width, rows, first_of_segment, pad;
};
hipError_t launch_bp_stats(hipStream_t s, uint32_t type_size, bool is_signed, const void *d_vals,
                           const uint64_t *d_validity, uint64_t n, void *d_stats);
hipError_t launch_bp_write(hipStream_t s, uint32_t type_size, const void *d_recs, uint64_t ngroups, const void *d_vals,
                           const uint64_t *d_validity, uint64_t block_stride, void *d_blocks);
hipError_t launch_bp_prepare(hipStream_t s, uint32_t type_size, void *d_groups, uint64_t ngroups, const void *d_blocks);
hipError_t launch_bp_unpack(hipStream_t s, uint32_t type_size, const void *d_groups, uint64_t ngroups,
                            const void *d_blocks, void *d_out);
hipError_t launch_bp_unpack_range(hipStream_t s, uint32_t type_size, const void *d_groups, uint32_t group0,
                                  uint32_t skip_first, uint64_t count, uint64_t out_off, const void *d_blocks,
                                  void *d_out);
hipError_t launch_bp_fetch(hipStream_t s, uint32_t type_size, const uint64_t *d_block_offs, const void *d_blocks,
                           const uint32_t *d_segs, const uint32_t *d_rows, uint64_t n, void *d_out);

} // namespace adac
