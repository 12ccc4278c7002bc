"""GPU input stage N1 (duplicate / prefix-read removal + id compaction, alga_preprocess_nodes) against the oracle's literal
ingest: the node set that reaches the graph creator must be identical -- rows, lengths, pairedReadOffset, counters."""
import os

import numpy as np
import pytest

import alga_amd
import gen_reads
import oracle_lib as O
from alga_amd.engine import device_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = alga_amd.Engine(0)
    yield e
    e.close()


def _device_nodes(eng, pr, mode=None):
    ds = eng.preprocess_nodes(pr["rows"], pr["len"], pr["remove_pref_reads"] if mode is None else mode, 3 + pr["li_kmer_length"])
    n, st = ds.n, ds.stride_words
    words = device_view(ds.d_words, (n, st)).cpu().numpy().view(np.uint32) if n else np.zeros((0, st), np.uint32)
    lens = device_view(ds.d_len, (n,)).cpu().numpy() if n else np.zeros(0, np.int32)
    pair = device_view(ds.d_pair_off, ((n + 3) // 4,)).cpu().numpy().view(np.uint8)[:n] if n else np.zeros(0, np.uint8)
    return ds, words, lens, pair


def _same_nodes(want, words, lens, pair):
    assert want["n"] == len(lens)
    assert (want["len"] == lens).all()
    W = want["W"]
    assert (words[:, :W] == want["words"]).all() and not words[:, W:].any()
    assert (want["pair_off"] == pair).all()


@pytest.mark.parametrize("name", O.FIXTURES)
def test_fixture_node_set_identical(eng, golden_dir, name):
    fx = O.Fixture(golden_dir, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        want = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
        pr = alga_amd.parse_files(f1, f2, threads=4, min_overlap=lo, rsoemo=rs)
    finally:
        fx.cleanup()
    assert (pr["min_overlap"], pr["rsoemo"], pr["li_kmer_length"], pr["LEN"]) == (want["min_overlap"], want["rsoemo"], want["li_kmer_length"], want["LEN"])
    ds, words, lens, pair = _device_nodes(eng, pr)
    _same_nodes(want, words, lens, pair)
    assert ds.removed_prefix == want["removed_prefix"]
    # and the graph built from the device-resident node set is the reference's graph
    import torch
    ptr, m = eng.prefsuf_device(device_view(ds.d_words, (ds.n, ds.stride_words)), device_view(ds.d_len, (ds.n,)), want["min_overlap"], want["rsoemo"])
    from alga_amd.engine import device_edges_to_numpy
    assert O.graph_bytes(ds.n, device_edges_to_numpy(ptr, m)) == fx.ref_graph()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_removal_modes_and_messy_reads(eng, tmp_path, mode):
    """duplicates, reads that prefix other reads (either strand), palindromes, variable lengths, removed reads in between"""
    rng = np.random.default_rng(5 + mode)
    g = rng.integers(0, 4, 3000, dtype=np.uint8)
    recs = []
    for _ in range(4000):
        L = int(rng.integers(40, 90))
        p = int(rng.integers(0, len(g) - L))
        r = g[p: p + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        s = "".join("ACGT"[c] for c in r)
        if rng.random() < 0.03:
            s = s[:10] + "N" + s[11:]
        recs.append(s)
    recs += ["ACGT" * 15, "ACGTACGTTACGTACGT" + "ACGTTGCA" * 6 + "TGCAACGT" * 6 + "ACGTACGTAACGTACGT"]       # STR, long palindrome-ish
    half = "".join("ACGT"[c] for c in g[100:130])
    recs.append(half + "".join("ACGT"[3 - "ACGT".index(c)] for c in reversed(half)))                      # a true palindrome
    rng.shuffle(recs)
    path = str(tmp_path / "m.fasta")
    with open(path, "w") as f:
        for i, s in enumerate(recs):
            f.write(">r%d\n%s\n" % (i, s))
    want = O.ingest(path, None, remove_pref_reads=mode)
    pr = alga_amd.parse_files(path, None, threads=3, remove_pref_reads=mode)
    ds, words, lens, pair = _device_nodes(eng, pr)
    _same_nodes(want, words, lens, pair)
    assert ds.removed_prefix == want["removed_prefix"]
    host = alga_amd.ingest_files(path, None, threads=3, remove_pref_reads=mode)                            # the host statement of the stage agrees too
    assert host["n"] == ds.n and (host["len"] == lens).all() and host["removed_short"] == ds.removed_short


def test_degenerate_inputs(eng):
    ds = eng.preprocess_nodes(np.zeros((0, 4), np.uint32), np.zeros(0, np.int32))
    assert ds.n == 0
    ds = eng.preprocess_nodes(np.zeros((6, 4), np.uint32), np.full(6, -1, np.int32))
    assert ds.n == 0
    with pytest.raises(alga_amd.AlgaError):
        eng.preprocess_nodes(np.zeros((3, 4), np.uint32), np.zeros(3, np.int32))                           # odd node count
    with pytest.raises(alga_amd.AlgaError):
        eng.preprocess_nodes(np.zeros((2, 1), np.uint32), np.array([40, 40], np.int32))                    # rows too short for the lengths


def test_read_without_its_twin(eng):
    # the reference's compaction looks at the even node of a pair (src/main.cpp:165-171): present without the odd one -> it asserts;
    # absent -> the pair is dropped silently (what happens to a read that equals its own reverse complement)
    rows = np.zeros((4, 4), np.uint32)
    rows[:, 0] = [0x1234567, 0x7654321, 0x2222222, 0x1111111]
    with pytest.raises(alga_amd.AlgaError) as ei:
        eng.preprocess_nodes(rows, np.array([40, -1, 40, 40], np.int32), remove_pref_reads=3)
    assert ei.value.code == -1
    ds = eng.preprocess_nodes(rows, np.array([-1, 40, 40, 40], np.int32), remove_pref_reads=3)
    assert ds.n == 2
