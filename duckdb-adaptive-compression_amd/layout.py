"""Host-side segment layout of the reference's bulk-load path (no device code).

How the reference cuts an appended column into ColumnSegments (paths relative to the reference checkout):
  * Appender flushes every FLUSH_COUNT = STANDARD_VECTOR_SIZE * 100 = 204 800 rows
    (src/include/duckdb/main/appender.hpp:32, src/main/appender.cpp:322-338); every flush is its own local
    append, i.e. a fresh row-group collection whose first row group starts at MAX_ROW_ID;
  * a row group holds 122 880 rows (STANDARD_VECTOR_SIZE * 60, src/include/duckdb/storage/table/row_group.hpp);
  * ColumnData::AppendTransientSegment (src/storage/table/column_data.cpp:343-355) creates a segment of
    STANDARD_VECTOR_SIZE * sizeof(T) bytes when start_row == MAX_ROW_ID (the first segment of a flush) and
    of Storage::BLOCK_SIZE = 262 136 bytes otherwise (src/include/duckdb/common/constants.hpp:104-106);
  * a SUCCINCT segment has segment_size / sizeof(T) slots (src/storage/table/column_segment.cpp:101-105).
Checked against the reference's own pragma_storage_info / GetTotalDataSize figures recorded in
SURVEY.md §3.1/§8c and BASELINE.md §2 (tests/test_host_layout.py).
"""
import numpy as np

STANDARD_VECTOR_SIZE = 2048      # src/include/duckdb/common/vector_size.hpp:17
FLUSH_COUNT = STANDARD_VECTOR_SIZE * 100
ROW_GROUP_SIZE = STANDARD_VECTOR_SIZE * 60
BLOCK_SIZE = 262144 - 8          # Storage::BLOCK_SIZE


def appender_segments(n_rows, type_size):
    """[(count, capacity_slots)] for every data segment of a column of n_rows appended through the Appender."""
    segs = []
    big = BLOCK_SIZE // type_size
    done = 0
    while done < n_rows:
        flush = min(FLUSH_COUNT, n_rows - done)
        in_flush = 0
        first_of_flush = True
        while in_flush < flush:
            rg = min(ROW_GROUP_SIZE, flush - in_flush)
            in_rg = 0
            while in_rg < rg:
                cap = STANDARD_VECTOR_SIZE if first_of_flush else big
                first_of_flush = False
                c = min(cap, rg - in_rg)
                segs.append((c, cap))
                in_rg += c
            in_flush += rg
        done += flush
    return segs


def appender_segment_counts(n_rows, type_size):
    return np.array([c for c, _ in appender_segments(n_rows, type_size)], dtype=np.uint32)


def aligned_value_offsets(counts, type_size, align_bytes=128):
    """Element offsets that give every segment its own align_bytes-aligned decoded block (the analogue of the
    per-segment buffer block UncompressSuccinct allocates, column_segment.cpp:461-471)."""
    per = align_bytes // type_size
    counts = np.asarray(counts, dtype=np.uint64)
    padded = (counts + np.uint64(per - 1)) // np.uint64(per) * np.uint64(per)
    offs = np.zeros(len(counts), dtype=np.uint64)
    if len(counts) > 1:
        offs[1:] = np.cumsum(padded)[:-1]
    return offs
