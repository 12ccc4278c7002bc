"""Pins oracle/ (the CPU restatement) against dumps produced by the reference itself.

The reference has no tests for this path; the pins are the reference's own
`--threads=1 --serialize=1` outputs committed by tools/make_golden.py.  Byte identity is required
for the graph dump and equality for every per-iteration edge count the reference printed
(src/GraphCreators/GraphCreatorPrefSuf.cpp:96-99).
"""
import numpy as np
import pytest

import oracle_lib as O


@pytest.mark.parametrize("name", O.FIXTURES)
def test_oracle_matches_reference_dump(golden_dir, name):
    fx = O.Fixture(golden_dir, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        nd = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
        assert nd["n"] == fx.meta["nodes"]
        assert nd["min_overlap"] == fx.meta["min_overlap"]
        assert nd["rsoemo"] == fx.meta["rsoemo"]
        assert nd["removed_prefix"] == fx.meta["removed_prefix_reads"]
        assert int((nd["len"] == 0).sum()) == fx.meta["removed_short_reads"]
        edges, after_iter, _ = O.prefsuf(nd["words"], nd["len"], nd["min_overlap"], nd["rsoemo"])
        ref = fx.ref_graph()
        assert len(edges) == fx.meta["edges_before_simplifier"]
        assert O.graph_bytes(nd["n"], edges) == ref
        want = fx.meta["edges_after_iter"]
        assert [int(x) for x in after_iter] == [e for _, e in want]
        assert want[0][0] == nd["min_overlap"]
    finally:
        fx.cleanup()


def test_graph_bytes_roundtrip():
    e = np.array([[0, 3, 5], [0, 4, 1], [2, 0, 7], [5, 1, 0]], dtype=np.int32)
    b = O.graph_bytes(6, e)
    n, e2 = O.parse_graph(b)
    assert n == 6 and (e2 == e).all()


def test_min_period_and_pack():
    L = O.lib()
    assert L.oracle_min_period(b"ACACACAC", 8) == 2
    assert L.oracle_min_period(b"ACGTACGTAC", 10) == 4
    assert L.oracle_min_period(b"AAAAAAA", 7) == 1
    assert L.oracle_min_period(b"ACGGT", 5) == 5
    w = np.zeros(2, dtype=np.uint32)
    s = b"ACGTTGCANACGTACGTA"
    L.oracle_pack(s, len(s), w.ctypes.data, 2)
    codes = {65: 0, 67: 1, 71: 2, 84: 3}
    for i, ch in enumerate(s):
        assert (int(w[i // 16]) >> (2 * (i % 16))) & 3 == codes.get(ch, 0)
