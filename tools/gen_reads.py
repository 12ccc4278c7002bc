#!/usr/bin/env python3
"""Deterministic synthetic read generator.

genome  = iid uniform ACGT of length G from `seed`
read    = `length`-nt window at a uniform random start, reverse-complemented with p=0.5,
          iid substitution errors at rate `err` (uniform over the 3 other bases)
output  = FASTA, one line per record, header `>r<i>`
paired  = `frag`-nt fragment; /1 = first `length` nt, /2 = first `length` nt of the fragment's
          reverse complement; mates randomly swapped between the two files
varlen  = read lengths uniform in [min_length, length]  (fixture F4)

The same function is used by the tests (small fixtures), by tools/make_golden.py and by bench.py
(which skips the FASTA step and packs the codes directly).
"""
import argparse
import numpy as np

_ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_genome(G, rng):
    return rng.integers(0, 4, size=G, dtype=np.uint8)


def revcomp_codes(c):
    """codes are A0 C1 G2 T3 -> complement is 3-c."""
    return (3 - c)[..., ::-1]


def sample_reads(n, length, G, seed, err=0.0, min_length=None):
    """Return (codes[n, length] uint8, lens[n] int32). codes beyond lens[i] are 0."""
    rng = np.random.default_rng(seed)
    genome = make_genome(G, rng)
    starts = rng.integers(0, G - length + 1, size=n)
    idx = starts[:, None] + np.arange(length)[None, :]
    codes = genome[idx]
    flip = rng.random(n) < 0.5
    codes[flip] = revcomp_codes(codes[flip])
    if err > 0:
        mask = rng.random(codes.shape) < err
        shift = rng.integers(1, 4, size=codes.shape, dtype=np.uint8)
        codes = np.where(mask, (codes + shift) & 3, codes).astype(np.uint8)
    lens = np.full(n, length, dtype=np.int32)
    if min_length is not None and min_length < length:
        lens = rng.integers(min_length, length + 1, size=n).astype(np.int32)
        codes = np.where(np.arange(length)[None, :] < lens[:, None], codes, 0).astype(np.uint8)
    return codes, lens


def sample_pairs(npairs, length, G, seed, frag=400, err=0.0):
    rng = np.random.default_rng(seed)
    genome = make_genome(G, rng)
    starts = rng.integers(0, G - frag + 1, size=npairs)
    idx = starts[:, None] + np.arange(frag)[None, :]
    fr = genome[idx]
    m1 = fr[:, :length].copy()
    m2 = revcomp_codes(fr)[:, :length].copy()
    swap = rng.random(npairs) < 0.5
    a = np.where(swap[:, None], m2, m1)
    b = np.where(swap[:, None], m1, m2)
    if err > 0:
        for arr in (a, b):
            mask = rng.random(arr.shape) < err
            shift = rng.integers(1, 4, size=arr.shape, dtype=np.uint8)
            arr[...] = np.where(mask, (arr + shift) & 3, arr)
    return a.astype(np.uint8), b.astype(np.uint8)


def write_fasta(path, codes, lens=None, prefix="r", suffix=""):
    n, length = codes.shape
    txt = _ALPHA[codes]
    with open(path, "wb") as f:
        chunk = 1 << 16
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            parts = []
            for i in range(s, e):
                li = length if lens is None else int(lens[i])
                parts.append(b">%s%d%s\n" % (prefix.encode(), i, suffix.encode()))
                parts.append(txt[i, :li].tobytes())
                parts.append(b"\n")
            f.write(b"".join(parts))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--n", type=int, required=True)
    ap.add_argument("--length", type=int, default=150)
    ap.add_argument("--min-length", type=int, default=None)
    ap.add_argument("--genome", type=int, required=True)
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--err", type=float, default=0.0)
    ap.add_argument("--paired", action="store_true", help="--n counts pairs; writes OUT_1.fasta / OUT_2.fasta")
    ap.add_argument("--frag", type=int, default=400)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    if a.paired:
        m1, m2 = sample_pairs(a.n, a.length, a.genome, a.seed, a.frag, a.err)
        base = a.out[:-6] if a.out.endswith(".fasta") else a.out
        write_fasta(base + "_1.fasta", m1, suffix="/1")
        write_fasta(base + "_2.fasta", m2, suffix="/2")
    else:
        codes, lens = sample_reads(a.n, a.length, a.genome, a.seed, a.err, a.min_length)
        write_fasta(a.out, codes, lens)


if __name__ == "__main__":
    main()
