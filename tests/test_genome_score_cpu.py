"""tools/genome_score.py (the referee of the approximate path: contigs placed on the known synthetic genome) on hand-made contigs:
exact, with substitutions inside and outside the bound, reverse complement, a chimera of two loci, a sequence from nowhere."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import genome_score  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def _seq(c):
    return ACGT[c].tobytes()


def test_placements():
    rng = np.random.default_rng(5)
    g = rng.integers(0, 4, 200_000, dtype=np.uint8)
    idx = genome_score.GenomeIndex(g)
    exact = g[1000:1800]
    rc = (3 - g[50_000:50_700])[::-1]
    noisy = g[90_000:91_000].copy()
    pos = rng.choice(1000, 15, replace=False)                # 1.5 % substitutions
    noisy[pos] = (noisy[pos] + 1) & 3
    too_noisy = g[120_000:121_000].copy()
    pos = rng.choice(1000, 40, replace=False)                # 4 %: beyond the bound, although its seeds hit
    too_noisy[pos] = (too_noisy[pos] + 2) & 3
    chimera = np.concatenate([g[10_000:10_400], g[150_000:150_500]])
    nowhere = rng.integers(0, 4, 600, dtype=np.uint8)
    dup = g[1200:1500]                                       # inside `exact`: duplicated bp
    rep = genome_score.genome_report([_seq(x) for x in (exact, rc, noisy, too_noisy, chimera, nowhere, dup)], idx)
    assert rep["contigs"] == 7 and rep["aligned"] == 4 and rep["misjoined"] == 2 and rep["chimeric"] >= 1 and rep["unplaced"] == 1
    assert rep["mismatches"] == 15 and rep["duplicated_bp"] == 300
    assert rep["aligned_bp"] == 800 + 700 + 1000 + 300
    assert abs(rep["genome_fraction"] - (800 + 700 + 1000) / 200_000) < 1e-9
    assert rep["misjoined_longest"] == 1000 and rep["aligned_n50"] == 800


def test_kmer_values_are_position_independent():
    c = np.array([0, 1, 2, 3] * 10, dtype=np.uint8)
    v = genome_score.kmer_values(c)
    assert len(v) == 40 - genome_score.K + 1 and v[0] == v[4] and v[0] != v[1]
