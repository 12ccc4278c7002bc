"""The source-side form of the transitive reduction (tests/source_side_rule.py) == the oracle's literal per-target replay."""
import os

import numpy as np
import pytest

import oracle_lib as O
from source_side_rule import decode_rows, preconditions, source_side_edges
from bucket_side_rule import bucket_side_edges

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(words, lens, lo, rs, af=None, at=None):
    assert preconditions(lens, lo, rs, af, at)
    want, _, _ = O.prefsuf(words, lens, lo, rs, af, at)
    got = source_side_edges(decode_rows(words, lens), lo, rs, af, at)
    assert got.shape == want.shape and np.array_equal(got, want), (got.shape, want.shape)
    # the same decision taken from the TARGET's candidate list (the seed-bucket-sharded N-GPU build: tests/bucket_side_rule.py)
    got2 = bucket_side_edges(decode_rows(words, lens), lo, rs, af, at)
    assert got2.shape == want.shape and np.array_equal(got2, want), (got2.shape, want.shape)
    return len(want)


@pytest.mark.parametrize("name", ["f1_cfg1", "f4_varlen", "f5_messy", "f6_l40"])
def test_fixture(name):
    fx = O.Fixture(GOLDEN, name)
    try:
        f1, f2 = fx.inputs()
        lo, rs = fx.explicit_params()
        nd = O.ingest(f1, f2, min_overlap=lo, rsoemo=rs)
        assert _check(nd["words"], nd["len"], nd["min_overlap"], nd["rsoemo"]) > 0
    finally:
        fx.cleanup()


def _random_nodes(rng, n_reads, genome_len, lo_len, hi_len, err, repeat_period=0, keep_prefix_reads=False):
    g = rng.integers(0, 4, genome_len, dtype=np.uint8)
    if repeat_period:                                   # tandem repeats: the same target overlaps a source at several offsets
        for s in range(0, genome_len - 400, 1500):
            unit = g[s: s + repeat_period].copy()
            for k in range(1, 300 // repeat_period):
                g[s + k * repeat_period: s + (k + 1) * repeat_period] = unit
    seqs = set()
    for _ in range(n_reads):
        L = int(rng.integers(lo_len, hi_len + 1))
        p = int(rng.integers(0, genome_len - L))
        r = g[p: p + L].copy()
        if err:
            m = rng.random(L) < err
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) & 3
        seqs.add(r.tobytes())
    seqs = sorted(seqs)
    if not keep_prefix_reads:                           # the default preprocessing removes reads that prefix another read
        seqs = [s for i, s in enumerate(seqs) if not (i + 1 < len(seqs) and seqs[i + 1].startswith(s))]
    nodes = []
    for s in seqs:
        a = np.frombuffer(s, dtype=np.uint8)
        nodes.append((3 - a[::-1]).astype(np.uint8).tobytes())
        nodes.append(s)
    W = (2 * hi_len + 31) // 32
    words = np.zeros((len(nodes), W), dtype=np.uint32)
    lens = np.zeros(len(nodes), dtype=np.int32)
    for i, s in enumerate(nodes):
        a = np.frombuffer(s, dtype=np.uint8).astype(np.uint64)
        lens[i] = len(a)
        for k in range(len(a)):
            words[i, k >> 4] |= np.uint32(int(a[k]) << (2 * (k & 15)))
    return words, lens


@pytest.mark.parametrize("seed,kw", [
    (1, dict(n_reads=1500, genome_len=6000, lo_len=60, hi_len=60, err=0.0)),
    (2, dict(n_reads=1500, genome_len=5000, lo_len=60, hi_len=60, err=0.01)),
    (3, dict(n_reads=1500, genome_len=5000, lo_len=40, hi_len=70, err=0.0)),
    (4, dict(n_reads=1500, genome_len=5000, lo_len=40, hi_len=70, err=0.0, keep_prefix_reads=True)),
    (5, dict(n_reads=2000, genome_len=6000, lo_len=60, hi_len=60, err=0.0, repeat_period=7)),
    (6, dict(n_reads=2000, genome_len=6000, lo_len=50, hi_len=64, err=0.003, repeat_period=11, keep_prefix_reads=True)),
    (7, dict(n_reads=800, genome_len=6000, lo_len=60, hi_len=60, err=0.0)),          # low coverage: long gaps, no big via
    (8, dict(n_reads=700, genome_len=6000, lo_len=60, hi_len=60, err=0.0, repeat_period=5)),     # same target at several offsets survives
    (9, dict(n_reads=900, genome_len=4000, lo_len=48, hi_len=64, err=0.0, repeat_period=9, keep_prefix_reads=True)),
])
def test_random(seed, kw):
    rng = np.random.default_rng(seed)
    words, lens = _random_nodes(rng, **kw)
    hi = int(lens.max())
    for lo, rs in [(int(hi * 0.55), int(hi * 0.775)), (int(hi * 0.4), int(hi * 0.6)), (int(hi * 0.5), int(hi * 0.5))]:
        _check(words, lens, lo, rs)


def test_masks_and_removed_nodes():
    rng = np.random.default_rng(11)
    words, lens = _random_nodes(rng, n_reads=1500, genome_len=5000, lo_len=60, hi_len=60, err=0.0)
    n = len(lens)
    dead = rng.random(n) < 0.1
    lens2 = np.where(dead, 0, lens).astype(np.int32)
    af = (rng.random(n) < 0.8).astype(np.uint8)
    at = np.maximum(af, (rng.random(n) < 0.5).astype(np.uint8))          # alignFrom implies alignTo
    _check(words, lens2, 33, 46, af, at)
    _check(words, lens2, 33, 46, None, None)
