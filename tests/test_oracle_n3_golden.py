"""Pins the oracle's restatement of the first simplifier step (oracle_cut_triangles: Graph::sortEdgesByIncreasingOffset +
GraphSimplifier::cutNonAndWeaklyMetricTriangles) against dumps the reference's own code produced (tools/make_golden_n3.py through
oracle/ref_driver.cpp): byte identity, in-list order included."""
import gzip
import json
import os

import pytest

import oracle_lib as O


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "n3_aftercut.json")))


@pytest.mark.parametrize("name", ["f1_cfg1", "f2_err2", "f4_varlen", "f5_messy", "f7_pkb"])
def test_cut_triangles_matches_reference_dump(golden_dir, name):
    c = _cases(golden_dir)[name]
    with gzip.open(os.path.join(golden_dir, c["graph_in"]), "rb") as f:
        n, e = O.parse_graph(f.read())
    assert len(e) == c["edges_before"]
    got = O.cut_triangles(n, e, c["max_offset_parallel_paths"])
    assert len(got) == c["edges_after"] and c["edges_after"] < c["edges_before"]
    with gzip.open(os.path.join(golden_dir, name + ".aftercut.graph.gz"), "rb") as f:
        assert O.graph_bytes(n, got) == f.read()
