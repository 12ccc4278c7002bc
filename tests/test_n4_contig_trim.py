"""Contig trimming (reference src/main.cpp:633-725): the oracle's restatement (oracle creator on
contigs + reverse complements, longest in-overlap between forward contigs) pinned to the values the reference's own code produced
(tools/make_golden_n4.py through oracle/ref_driver.cpp), and the GPU form (alga_contig_trim_host) against both."""
import gzip
import os

import numpy as np
import pytest

import oracle_lib as O


def _golden(golden_dir):
    words, lens = O.load_nodes_bin(os.path.join(golden_dir, "n4_contigs.nodes.bin.gz"))
    want, seqs = [], []
    for line in gzip.open(os.path.join(golden_dir, "n4_contigs.trim.txt.gz"), "rt"):
        t, s = line.split()
        want.append(int(t)); seqs.append(s)
    return words, lens, np.array(want, dtype=np.int32), seqs


def _cut(words, lens, trim):
    out = []
    for i in range(len(lens)):
        s = "".join("ACGT"[(int(words[i, k >> 4]) >> ((k & 15) << 1)) & 3] for k in range(int(lens[i])))
        out.append(s[trim[i]:] if trim[i] + 10 < len(s) else "CCCC")          # src/main.cpp:703-706 with trimRight = 0
    return out


def test_oracle_contig_trim_matches_reference_values(golden_dir):
    words, lens, want, seqs = _golden(golden_dir)
    got = O.contig_trim(words, lens)
    assert (got == want).all() and (want > 0).sum() > 20
    assert _cut(words, lens, got) == seqs


@pytest.mark.gpu
def test_gpu_contig_trim_matches_reference_values(golden_dir):
    import alga_amd
    words, lens, want, seqs = _golden(golden_dir)
    eng = alga_amd.Engine(0)
    try:
        got = eng.contig_trim(words, lens)
        assert (got == want).all()
        assert _cut(words, lens, got) == seqs
        # a second, longer set against the oracle only
        rng = np.random.default_rng(73)
        g = rng.integers(0, 4, 200000, dtype=np.uint8)
        sq, p = [], 0
        while p < len(g) - 12000:
            L = int(rng.integers(2000, 11000))
            sq.append(g[p: p + L].copy())
            p += L - int(rng.integers(25, 500))
        mx = max(len(x) for x in sq)
        codes = np.zeros((len(sq), mx), dtype=np.uint8)
        ln = np.array([len(x) for x in sq], dtype=np.int32)
        for i, x in enumerate(sq):
            codes[i, : len(x)] = x
        w = alga_amd.pack_reads(codes, ln)
        assert (eng.contig_trim(w, ln) == O.contig_trim(w, ln)).all()
        assert eng.contig_trim(np.zeros((0, 4), np.uint32), np.zeros(0, np.int32)).shape == (0,)
    finally:
        eng.close()
