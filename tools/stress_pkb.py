#!/usr/bin/env python3
"""Randomised cross-check of the approximate supplement on the GPU: this round's kernels (tip records, four groups per wave, 96-bit k-mer walk,
one-pass head list, own sorts, merge tail) against round 4's forms of the same steps (engine option pkb_legacy = 255: every piece switched
back), which the test suite holds to the oracle.  Same graph, same work counters, on random read sets: lengths 80 .. 250 (fixed or variable),
coverage 8 .. 300, 1 .. 4 % substitutions, tandem repeats in a third of the cases.
usage: tools/stress_pkb.py [n_cases=100] [first_seed=5000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alga_amd  # noqa: E402


def make_case(seed):
    rng = np.random.default_rng(seed)
    maxlen = int(rng.choice([80, 100, 125, 150, 150, 200, 250]))
    varlen = rng.random() < 0.35
    n_reads = int(rng.integers(400, 5000))
    cov = float(rng.choice([8, 20, 40, 80, 150, 300]))
    err = float(rng.choice([0.01, 0.02, 0.02, 0.03, 0.04]))
    G = max(maxlen * 3, int(n_reads * maxlen / cov))
    g = rng.integers(0, 4, G, dtype=np.uint8)
    if rng.random() < 0.33 and G > 600:
        unit = rng.integers(0, 4, int(rng.integers(20, 60)), dtype=np.uint8)
        at = int(rng.integers(0, G - 500))
        rep = np.tile(unit, 500 // len(unit) + 1)[:500]
        g[at:at + 500] = rep
    rows, lens = [], []
    for _ in range(n_reads):
        L = int(rng.integers(int(maxlen * 0.75), maxlen + 1)) if varlen else maxlen
        st = int(rng.integers(0, G - L + 1))
        r = g[st:st + L].copy()
        m = rng.random(L) < err
        r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) & 3
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        for x in ((3 - r)[::-1], r):
            row = np.zeros(maxlen, np.uint8); row[:len(x)] = x
            rows.append(row); lens.append(len(x))
    return np.stack(rows), np.array(lens, dtype=np.int32), err


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    eng = alga_amd.Engine(0)
    bad = 0
    tot_add = 0
    classes = np.zeros(8, dtype=np.int64)
    for c in range(n_cases):
        codes, lens, err = make_case(seed0 + c)
        words = alga_amd.pack_reads(codes, lens)
        avg = float(lens.mean())
        lo, rs = alga_amd.derive_params(avg)
        pre = eng.prefsuf_host(words, lens, lo, rs)
        p = eng.pkb_params(avg, max(err, 0.011), min(2 * lo // 3, 60))
        new = eng.pkb_supplement_host(words, lens, pre, p)
        s_new = eng.pkb_last_stats()
        eng.set_option("pkb_legacy", 255)
        try:
            old = eng.pkb_supplement_host(words, lens, pre, p)
            s_old = eng.pkb_last_stats()
        finally:
            eng.set_option("pkb_legacy", 0)
        same = new.shape == old.shape and bool((new == old).all()) and all(s_new[k] == s_old[k] for k in ("kmers", "groups", "can_align_calls", "edges_after", "group_hist", "max_group"))
        if not same:
            bad += 1
            print("MISMATCH seed", seed0 + c, new.shape, old.shape, flush=True)
        tot_add += len(new) - len(pre)
        classes += np.array(s_new["group_hist"], dtype=np.int64).sum(axis=0)
        if (c + 1) % 25 == 0:
            print("..", c + 1, "cases, mismatches", bad, flush=True)
    print("cases %d: edges added %d; groups by size (2, 3, 4, 5-7, 8-15, 16-31, 32-64, more) %s; mismatches %d" % (n_cases, tot_add, classes.tolist(), bad))
    eng.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
