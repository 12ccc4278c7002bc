#!/usr/bin/env python3
"""Randomised cross-check on the GPU: the two forms of the transitive reduction (per-target replay, source-side) must give
the same edges on every input the source-side form accepts -- the source-side form through the seed-table probe, through the
clustered probe with k_probe_stream first (sources in key order and in id order), and through the clustered probe's general kernel alone.  Random read lengths (fixed / variable), coverage, substitution
errors, tandem repeats, exact duplicates and prefix reads left in, masks, min_overlap / rsoemo choices; and, round 4, the probe through piles (a build without
work counters) wherever the input allows it.
usage: tools/stress_forms.py [--pile] [n_cases=100] [first_seed=1000]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alga_amd  # noqa: E402


PILE_DOMAIN = False          # --pile: only cases the pile path takes (one read length <= 144, no masks, at most 64 suffix windows, few or no errors)


def make_case(seed):
    rng = np.random.default_rng(seed)
    maxlen = int(rng.choice([48, 64, 80, 100, 144, 150, 200, 250]))
    if PILE_DOMAIN:
        maxlen = int(rng.choice([64, 80, 94, 100, 128, 144]))
    varlen = rng.random() < 0.4 and not PILE_DOMAIN
    minlen = int(maxlen * rng.uniform(0.6, 0.95)) if varlen else maxlen
    n_reads = int(rng.integers(300, 6000))
    cov = float(rng.choice([3, 8, 20, 50, 120]))
    G = max(maxlen * 3, int(n_reads * (maxlen + minlen) / 2 / cov))
    err = float(rng.choice([0, 0, 0.001, 0.005, 0.02]))
    if PILE_DOMAIN:
        err = float(rng.choice([0, 0, 0, 0.0005, 0.002]))
    g = rng.integers(0, 4, G, dtype=np.uint8)
    if rng.random() < 0.4:                                  # tandem repeats / multi-copy repeats
        period = int(rng.integers(3, 60))
        for s0 in range(0, max(1, G - 400), int(rng.integers(500, 3000))):
            unit = g[s0: s0 + period].copy()
            for k in range(1, 300 // period):
                if s0 + (k + 1) * period <= G:
                    g[s0 + k * period: s0 + (k + 1) * period] = unit
    if rng.random() < 0.3 and G > 2000:                     # a dispersed repeat with several copies
        rep = g[100:100 + int(rng.integers(100, 400))].copy()
        for _ in range(int(rng.integers(2, 12))):
            p = int(rng.integers(0, G - len(rep)))
            g[p: p + len(rep)] = rep
    reads = []
    for _ in range(n_reads):
        L = int(rng.integers(minlen, maxlen + 1))
        p = int(rng.integers(0, G - L + 1))
        r = g[p: p + L].copy()
        if err:
            m = rng.random(L) < err
            r[m] = (r[m] + rng.integers(1, 4, int(m.sum()))) & 3
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    if rng.random() < 0.5:                                  # remove exact duplicates (as the reference's preprocessing would)
        seen, uniq = set(), []
        for r in reads:
            k = min(r.tobytes(), (3 - r)[::-1].tobytes())
            if k not in seen:
                seen.add(k); uniq.append(r)
        reads = uniq
    n = len(reads)
    codes = np.zeros((2 * n, maxlen), dtype=np.uint8)
    lens = np.zeros(2 * n, dtype=np.int32)
    for i, r in enumerate(reads):
        codes[2 * i, : len(r)] = (3 - r)[::-1]
        codes[2 * i + 1, : len(r)] = r
        lens[2 * i] = lens[2 * i + 1] = len(r)
    words = alga_amd.pack_reads(codes, lens)
    lo = int(maxlen * rng.uniform(0.35, 0.7))
    if maxlen - lo > 127:
        lo = maxlen - int(rng.integers(20, 127))
    if PILE_DOMAIN:
        lo = maxlen - int(rng.integers(8, 64))
    rs = int(rng.integers(lo, maxlen + 2))
    af = at = None
    if rng.random() < 0.25 and not PILE_DOMAIN:
        dead = rng.random(2 * n) < 0.05
        lens = np.where(dead, 0, lens).astype(np.int32)
        words[dead] = 0
    if rng.random() < 0.2 and not PILE_DOMAIN:
        af = (rng.random(2 * n) < 0.85).astype(np.uint8)
        at = np.maximum(af, (rng.random(2 * n) < 0.5).astype(np.uint8))
    desc = dict(seed=seed, maxlen=maxlen, minlen=minlen, reads=n, genome=G, err=err, lo=lo, rs=rs, masks=af is not None)
    return words, lens, lo, rs, af, at, desc


def main():
    global PILE_DOMAIN
    if "--pile" in sys.argv:
        PILE_DOMAIN = True
        sys.argv.remove("--pile")
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    eng = alga_amd.Engine(0)
    used, declined, bad = 0, 0, 0
    big = generic = clustered = piled = pile_kept = pile_forced = ranged_builds = 0
    for seed in range(first, first + n_cases):
        words, lens, lo, rs, af, at, desc = make_case(seed)
        a = eng.prefsuf_host(words, lens, lo, rs, af, at, reduction="per_target")
        ok = True
        for probe, pairs, order in (("table", 1, 1), ("cluster", 1, 1), ("cluster", 1, 0), ("cluster", 0, 1)):
            eng.set_option("probe", probe)
            eng.set_option("cluster_pairs", pairs)
            eng.set_option("cluster_order", order)
            try:
                b = eng.prefsuf_host(words, lens, lo, rs, af, at, reduction="source_side", collect_stats=True)
            except alga_amd.AlgaError as e:
                if e.code != -7:
                    raise
                ok = False
                break
            finally:
                eng.set_option("probe", "auto")
                eng.set_option("cluster_pairs", 1)
                eng.set_option("cluster_order", 1)
            st = eng.last_stats()
            if probe == "table":
                big += st["big_sources"] > 0
                generic += st["generic_sources"] > 0
            elif pairs:
                clustered += st["probe_used"] == 2
            if a.shape != b.shape or not (a == b).all():
                bad += 1
                print("MISMATCH", probe, pairs, desc, a.shape, b.shape, flush=True)
        if not ok:
            declined += 1
            continue
        used += 1
        # the probe through piles (prefsuf_pile.hip): a build WITHOUT work counters through the clustered probe -- reads of one length without masks take it
        for mode in (1, 2):                                # 2: without the sample that declines irregular data -- every irregular bucket's sources are handed on one by one
            eng.set_option("probe", "cluster")
            eng.set_option("pile", mode)
            try:
                c = eng.prefsuf_host(words, lens, lo, rs, af, at, reduction="source_side")
            finally:
                eng.set_option("probe", "auto")
                eng.set_option("pile", 1)
            st = eng.last_stats()
            if mode == 1 and st["pile_buckets"] > 0:
                piled += 1
                pile_kept += st["pile_irregular"] * alga_amd.engine.PILE_IRREGULAR_ONE_IN <= st["pile_buckets"]
            if mode == 2:
                pile_forced += st["ms_pile"] > 0
            if a.shape != c.shape or not (a == c).all():
                bad += 1
                print("MISMATCH pile", mode, desc, a.shape, c.shape, flush=True)
        # ... and for a rank's share (round 5): the sources dealt out by id range -- random cuts, any parity -- through the piles, pure (2) and mixed (3)
        # form; a range is a build of its own (keys_shared 0) or a further piece of the build before it (keys_shared 2), at random
        if PILE_DOMAIN and st["ms_pile"] > 0:
            import torch
            from alga_amd.engine import device_view
            rng = np.random.default_rng(seed + 7)
            n = len(lens)
            stride = words.shape[1]
            dw = torch.from_numpy(np.ascontiguousarray(words).view(np.int32).reshape(n, stride)).cuda()
            dl = torch.from_numpy(lens.astype(np.int32)).cuda()
            for mode in (2, 3):
                cuts = sorted(set([0, n] + [int(x) for x in rng.integers(1, n, size=int(rng.integers(1, 5)))]))
                eng.set_option("probe", "cluster")
                eng.set_option("pile", mode)
                try:
                    parts = []
                    for k, (lo_id, hi_id) in enumerate(zip(cuts[:-1], cuts[1:])):
                        ks = 2 if (k > 0 and rng.random() < 0.5) else 0
                        r = eng.build_range_device(dw, dl, lo, rs, lo_id, hi_id, keys_shared=ks)
                        ranged_builds += 1
                        parts.append(device_view(r[0], (r[1], 3), dw.device).cpu().numpy().astype(a.dtype))
                finally:
                    eng.set_option("probe", "auto")
                    eng.set_option("pile", 1)
                c = np.concatenate(parts) if parts else a[:0]
                if a.shape != c.shape or not (a == c).all():
                    bad += 1
                    print("MISMATCH pile ranges", mode, cuts, desc, a.shape, c.shape, flush=True)
    print("cases %d: source-side used %d (all-pairs branch in %d, second pass in %d, clustered probe in %d; pile path sampled in %d, kept the build in %d, forced through it in %d, %d builds of id ranges through it), declined %d, mismatches %d" %
          (n_cases, used, generic, big, clustered, piled, pile_kept, pile_forced, ranged_builds, declined, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
