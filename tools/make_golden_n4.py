#!/usr/bin/env python3
"""Golden vector for the contig-trimming call of the reference (src/main.cpp:633-725): contig-shaped
sequences (kb-long, neighbours sharing 30-450 nt, a few contained / branching ones) through oracle/_ref/ref_driver (mode `trim`: the
reference's own GraphCreatorPrefSuf with both thresholds at 25, trimLeft, cut sequences) -> tests/golden/n4_contigs.*.
Needs /root/reference (oracle/Makefile).  usage: tools/make_golden_n4.py"""
import gzip
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import alga_amd  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DRV = os.path.join(ROOT, "oracle", "_ref", "ref_driver")


def contigs(seed=71):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, 90000, dtype=np.uint8)
    seqs, p = [], 0
    while p < len(g) - 7000:
        L = int(rng.integers(900, 5000))
        seqs.append(g[p: p + L].copy())
        p += L - int(rng.integers(30, 450))          # the next contig starts inside the overlap
    for k in (3, 9, 14):                             # branches: a second contig that shares the end of contig k and then diverges
        tail = seqs[k][-int(rng.integers(40, 300)):]
        seqs.append(np.concatenate([tail, rng.integers(0, 4, int(rng.integers(600, 2000)), dtype=np.uint8)]))
    seqs.append((3 - seqs[5])[::-1].copy())          # a contig given on the other strand
    seqs.append(rng.integers(0, 4, 20, dtype=np.uint8))   # shorter than the threshold: takes no part
    order = rng.permutation(len(seqs))
    return [seqs[i] for i in order]


def write_nodes(path, seqs):
    maxlen = max(len(s) for s in seqs)
    codes = np.zeros((len(seqs), maxlen), dtype=np.uint8)
    lens = np.array([len(s) for s in seqs], dtype=np.int32)
    for i, s in enumerate(seqs):
        codes[i, : len(s)] = s
    words = alga_amd.pack_reads(codes, lens)
    with open(path, "wb") as f:
        f.write(np.array([len(lens), words.shape[1]], dtype=np.int32).tobytes())
        f.write(lens.tobytes())
        f.write(words.tobytes())
    return words, lens


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    seqs = contigs()
    with tempfile.TemporaryDirectory() as wd:
        nodes = os.path.join(wd, "c.bin")
        write_nodes(nodes, seqs)
        out = os.path.join(wd, "trim.txt")
        print(subprocess.run([DRV, "trim", nodes, out, "144"], capture_output=True, text=True, check=True).stdout.strip())
        with open(nodes, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "n4_contigs.nodes.bin.gz"), "wb", mtime=0) as g:
            g.write(f.read())
        with open(out, "rb") as f, gzip.GzipFile(os.path.join(GOLD, "n4_contigs.trim.txt.gz"), "wb", mtime=0) as g:
            g.write(f.read())
    trims = [int(l.split()[0]) for l in gzip.open(os.path.join(GOLD, "n4_contigs.trim.txt.gz"), "rt")]
    print("contigs", len(seqs), "trimmed", sum(t > 0 for t in trims), "max trim", max(trims))


if __name__ == "__main__":
    main()
