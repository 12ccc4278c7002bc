#!/usr/bin/env python3
"""Randomised cross-check of the input stage: GPU duplicate / prefix-read removal + compaction (alga_preprocess_nodes) against
the host statement of the same stage (alga_ingest_files, itself checked against the oracle in tests/) on random messy FASTA /
FASTQ files: duplicates on either strand, reads that prefix other reads, palindromes, N, short-tandem-repeat reads, variable
lengths, paired files, all three removal modes.   usage: tools/stress_ingest.py [n_cases=60] [first_seed=1]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alga_amd  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402


def write_reads(path, recs, fastq):
    with open(path, "w") as f:
        for i, s in enumerate(recs):
            if fastq:
                f.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
            else:
                f.write(">r%d\n%s\n" % (i, s))


def make_case(seed, wd):
    rng = np.random.default_rng(seed)
    G = int(rng.integers(800, 6000))
    g = rng.integers(0, 4, G, dtype=np.uint8)
    lo_len, hi_len = sorted(int(x) for x in rng.integers(36, 160, 2))
    n = int(rng.integers(50, 4000))
    paired = rng.random() < 0.3
    fastq = rng.random() < 0.3

    def one():
        L = int(rng.integers(lo_len, hi_len + 1))
        p = int(rng.integers(0, G - L + 1))
        r = g[p: p + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        s = "".join("ACGT"[c] for c in r)
        u = rng.random()
        if u < 0.03:
            k = int(rng.integers(0, L)); s = s[:k] + "N" + s[k + 1:]
        elif u < 0.05:
            unit = s[: int(rng.integers(1, 21))]; s = (unit * (L // len(unit) + 1))[:L]      # STR read
        elif u < 0.08:
            h = s[: L // 2]; s = h + "".join("ACGT"[3 - "ACGT".index(c)] for c in reversed(h))  # palindrome
        return s
    recs = [one() for _ in range(n)]
    for _ in range(int(n * rng.uniform(0, 0.2))):                       # exact duplicates and prefixes of existing reads
        s = recs[int(rng.integers(0, len(recs)))]
        recs.append(s if rng.random() < 0.5 else s[: max(30, int(len(s) * rng.uniform(0.5, 1.0)))])
    rng.shuffle(recs)
    ext = ".fastq" if fastq else ".fasta"
    f1 = os.path.join(wd, "a%d%s" % (seed, ext))
    f2 = None
    if paired:
        half = len(recs) // 2
        write_reads(f1, recs[:half], fastq)
        f2 = os.path.join(wd, "b%d%s" % (seed, ext))
        write_reads(f2, recs[half: 2 * half], fastq)
    else:
        write_reads(f1, recs, fastq)
    return f1, f2, int(rng.integers(1, 4))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    eng = alga_amd.Engine(0)
    bad = 0
    with tempfile.TemporaryDirectory() as wd:
        for seed in range(first, first + n_cases):
            f1, f2, mode = make_case(seed, wd)
            try:
                host = alga_amd.ingest_files(f1, f2, threads=4, remove_pref_reads=mode)
            except alga_amd.AlgaError as e:                             # "kept but its reverse complement removed": both must say so
                pr = alga_amd.parse_files(f1, f2, threads=4, remove_pref_reads=mode)
                try:
                    eng.preprocess_nodes(pr["rows"], pr["len"], mode, 3 + pr["li_kmer_length"])
                    bad += 1
                    print("HOST FAILED, GPU DID NOT", seed, e, flush=True)
                except alga_amd.AlgaError:
                    pass
                continue
            pr = alga_amd.parse_files(f1, f2, threads=4, remove_pref_reads=mode)
            ds = eng.preprocess_nodes(pr["rows"], pr["len"], mode, 3 + pr["li_kmer_length"])
            n, st = ds.n, ds.stride_words
            ok = n == host["n"] and ds.removed_prefix == host["removed_prefix"] and ds.removed_short == host["removed_short"]
            if ok and n:
                words = device_view(ds.d_words, (n, st)).cpu().numpy().view(np.uint32)
                lens = device_view(ds.d_len, (n,)).cpu().numpy()
                pair = device_view(ds.d_pair_off, ((n + 3) // 4,)).cpu().numpy().view(np.uint8)[:n]
                W = min(st, host["stride"])
                ok = (lens == host["len"]).all() and (pair == host["pair_off"]).all() and (words[:, :W] == host["words"][:, :W]).all() \
                    and not words[:, W:].any() and not host["words"][:, W:].any()
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, "mode", mode, "gpu n", n, "host n", host["n"], flush=True)
    print("cases %d, mismatches %d" % (n_cases, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
