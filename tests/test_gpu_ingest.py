"""GPU input stages against the oracle's literal ingest: the node set that reaches the graph creator must be identical -- rows,
lengths, pairedReadOffset, counters -- both for N1 alone (duplicate / prefix-read removal + id compaction on host-parsed rows,
alga_preprocess_nodes) and for the whole input stage on the GPU (N2 + N1: files in, alga_ingest_device)."""
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


def _read_back(ds):
    n, st = ds.n, ds.stride_words
    words = device_view(ds.d_words, (n, st)).cpu().numpy().view(np.uint32) if n else np.zeros((0, st), np.uint32)
    lens = device_view(ds.d_len, (n,)).cpu().numpy() if n else np.zeros(0, np.int32)
    pair = device_view(ds.d_pair_off, ((n + 3) // 4,)).cpu().numpy().view(np.uint8)[:n] if n else np.zeros(0, np.uint8)
    return words, lens, pair


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
    # the whole input stage on the GPU: files in, the same node set out
    fx2 = O.Fixture(golden_dir, name)
    try:
        g1, g2 = fx2.inputs()
        ds2, info = eng.ingest_device(g1, g2, min_overlap=lo, rsoemo=rs)
    finally:
        fx2.cleanup()
    _same_nodes(want, *_read_back(ds2))
    assert (info["min_overlap"], info["rsoemo"], info["li_kmer_length"], info["LEN"]) == (want["min_overlap"], want["rsoemo"], want["li_kmer_length"], want["LEN"])
    assert (info["removed_n"], info["removed_str"], ds2.removed_prefix) == (want["removed_n"], want["removed_str"], want["removed_prefix"])
    ds = ds2
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
    ds2, info = eng.ingest_device(path, None, remove_pref_reads=mode)                                      # files in: N2 + N1 on the GPU
    _same_nodes(want, *_read_back(ds2))
    assert (info["removed_n"], info["removed_str"]) == (want["removed_n"], want["removed_str"])
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


def test_device_ingest_messy_files(eng, tmp_path):
    """record shapes the reference's reader handles: blanks around the sequence, text after a blank, short reads that escape the
    trimming, U with and without --rna, a last line without newline, input that ends at an empty sequence line, FASTQ; and the
    inputs the device stage hands back (other file types, random N replacement) or rejects (a letter outside ACGTNU)."""
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, 4000, dtype=np.uint8)

    def read(L):
        p = int(rng.integers(0, len(g) - L))
        return "".join("ACGT"[c] for c in g[p: p + L])
    recs = [read(int(rng.integers(30, 120))) for _ in range(1500)]
    recs[3] = "   " + recs[3]
    recs[4] = recs[4] + " trailing words"
    recs[5] = read(14)                                     # shorter than trim + 10: not trimmed, then STR (<= 20 nt)
    recs[6] = read(25)
    recs[7] = recs[7][:20] + "U" + recs[7][21:]
    recs[8] = recs[8][:5] + "N" + recs[8][6:]
    for ext, lpr in (("fasta", 2), ("fastq", 4)):
        path = str(tmp_path / ("m." + ext))
        with open(path, "w") as f:
            for i, s in enumerate(recs):
                if lpr == 2:
                    f.write(">r%d\n%s" % (i, s))
                else:
                    f.write("@r%d\n%s\n+\n%s" % (i, s, "I" * len(s)))
                f.write("\n" if i + 1 < len(recs) else "")                       # no newline after the last record
        for kw in ({}, {"rna": 1}, {"trim_left": 0, "trim_right": 7}):
            want = O.ingest(path, None, **kw)
            ds, info = eng.ingest_device(path, None, **kw)
            _same_nodes(want, *_read_back(ds))
            assert info["records"] == len(recs)
    # the input ends at the first empty sequence line
    path = str(tmp_path / "cut.fasta")
    with open(path, "w") as f:
        for i, s in enumerate(recs[:100]):
            f.write(">r%d\n%s\n" % (i, s if i != 60 else ""))
    want = O.ingest(path, None)
    ds, info = eng.ingest_device(path, None)
    _same_nodes(want, *_read_back(ds))
    assert info["records"] == 60
    # handed back / rejected
    other = str(tmp_path / "reads.txt")
    open(other, "w").write("\n".join(recs[:10]) + "\n")
    for bad_call in (lambda: eng.ingest_device(other, None), lambda: eng.ingest_device(path, None, remove_reads_with_n=0)):
        with pytest.raises(alga_amd.AlgaError) as ei:
            bad_call()
        assert ei.value.code == -7
    badf = str(tmp_path / "bad.fasta")
    open(badf, "w").write(">a\n%s\n>b\n%sX%s\n" % (recs[0], recs[1][:30], recs[1][30:]))
    with pytest.raises(alga_amd.AlgaError) as ei:
        eng.ingest_device(badf, None)
    assert ei.value.code == -6 and "s[i] = X" in str(ei.value)
