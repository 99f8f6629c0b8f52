"""The BITPACKING oracle (oracle/bitpacking_oracle.c) against the reference's OWN fixtures for this codec: the data
shapes and expected results of /root/reference/test/sql/storage/compression/bitpacking/*.test, committed as data in
tests/golden/bitpacking_sql_cases.json.  Every case runs under each forced mode the file loops over: the column must
be encodable ("compression = BitPacking") and scan back to the rows / aggregates the file expects.

What this pins with reference-held material: the codec's round-trip semantics under every forced mode, NULL handling,
point fetches, every bit width of every type size, the numerical limits of the types and the refusal of a column whose
range needs the type's full width (bitpacking_bitwidths / bitpacking_types.test_coverage).  What stays a restatement: the byte layout of the block images above the 32-value pack routine
(pinned separately against the reference's real fastpforlib, tests/golden/fastpfor_vectors.json)."""
import numpy as np
import pytest

import bp_sql_cases as sc
from oracle import bitpacking as bp


@pytest.mark.parametrize("case_id", sc.case_ids())
def test_oracle_reproduces_the_reference_sql_expectations(case_id):
    case = next(c for c in sc.load_cases() if c["id"] == case_id)
    vals, valid = sc.build_column(case)
    for mode in case["forced_modes"]:
        if sc.refused(case):  # the file expects 'Uncompressed': BitpackingAnalyze must decline the column
            with pytest.raises(ValueError):
                bp.Compressed(vals, valid, force_mode=sc.MODE_CODE[mode])
            continue
        if case.get("may_be_refused"):  # the file does not assert the compression of this table
            try:
                bp.Compressed(vals, valid, force_mode=sc.MODE_CODE[mode])
            except ValueError:
                continue
        comp = bp.Compressed(vals, valid, force_mode=sc.MODE_CODE[mode])  # raises when the codec cannot encode
        got = np.concatenate([comp.scan(i) for i in range(comp.nseg)])
        assert len(got) == len(vals)
        starts = [comp.start(i) for i in range(comp.nseg)]

        def fetch(row):
            i = max(k for k, s in enumerate(starts) if s <= row)
            return comp.scan(i, row - starts[i], 1)[0]

        sc.check_expectations(case, got, valid, fetch=fetch)
        # the forced mode is taken wherever it applies: the file's comment promises every mode finds a range to use
        if "comment_expectation" in case and valid is None:
            assert comp.groups_by_mode()[mode] >= 1, (mode, comp.groups_by_mode())
