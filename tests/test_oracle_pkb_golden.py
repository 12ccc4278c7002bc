"""Pins the oracle's restatement of the approximate supplement (oracle/alga_oracle_pkb.cpp)
against vectors produced by the reference's own code (tools/make_golden_pkb.py through oracle/ref_driver.cpp)."""
import gzip
import json
import os

import numpy as np

import oracle_lib as O


def _meta(golden_dir):
    return json.load(open(os.path.join(golden_dir, "f7_pkb.json")))


def test_can_align_matches_reference_vectors(golden_dir):
    meta = _meta(golden_dir)
    words, lens = O.load_nodes_bin(os.path.join(golden_dir, "f7_pkb.nodes.bin.gz"))
    z = np.load(os.path.join(golden_dir, "f7_pkb.canalign.npz"))
    p = O.pkb_params(meta["avg_len"], error_rate_percent=meta["error_rate_percent"])
    assert (p.min_overlap_area, p.max_offset_pct, p.min_identity_pct) == (meta["min_overlap_area"], meta["max_offset_pct"], meta["min_identity_pct"])
    got = O.can_align(words, lens, z["triples"], p)
    assert int(got.sum()) == meta["accepted"] and 500 < meta["accepted"] < len(got) - 500
    assert (got == z["verdict"]).all()


def test_li_kmers_match_reference_vectors(golden_dir):
    meta = _meta(golden_dir)
    words, lens = O.load_nodes_bin(os.path.join(golden_dir, "f7_pkb.nodes.bin.gz"))
    with gzip.open(os.path.join(golden_dir, "f7_pkb.likmers.bin.gz"), "rb") as f:
        buf = f.read()
    pos = 0
    prio = [0, 1, 2, 3]
    checked = 0
    for rot in range(4):
        for i in range(meta["likmer_nodes"]):
            if lens[i] < meta["li_k"]:
                continue
            c = int(np.frombuffer(buf[pos:pos + 4], dtype=np.int32)[0]); pos += 4
            want_h, want_i = [], []
            for _ in range(c):
                want_h.append(int(np.frombuffer(buf[pos:pos + 8], dtype=np.uint64)[0])); pos += 8
                want_i.append(int(np.frombuffer(buf[pos:pos + 4], dtype=np.int32)[0])); pos += 4
            h, ind = O.li_kmers(words[i], lens[i], meta["li_k"], meta["li_intervals"], prio)
            assert h.tolist() == want_h and ind.tolist() == want_i
            checked += 1
        prio = prio[1:] + prio[:1]
    assert pos == len(buf) and checked > 1000


def test_supplement_graph_matches_reference(golden_dir):
    meta = _meta(golden_dir)
    words, lens = O.load_nodes_bin(os.path.join(golden_dir, "f7_pkb.nodes.bin.gz"))
    with gzip.open(os.path.join(golden_dir, meta["pre_graph"]), "rb") as f:
        n, pre = O.parse_graph(f.read())
    assert n == len(lens) and len(pre) == meta["edges_before"]
    p = O.pkb_params(meta["avg_len"], error_rate_percent=meta["error_rate_percent"])
    got, calls = O.supplement(words, lens, pre, p, meta["kmer_length_bucket"])
    assert len(got) == meta["edges_after"] and calls > 0
    with gzip.open(os.path.join(golden_dir, "f7_pkb.supplement.graph.gz"), "rb") as f:
        ref = f.read()
    assert O.graph_bytes(n, got) == ref
