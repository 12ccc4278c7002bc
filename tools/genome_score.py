#!/usr/bin/env python3
"""Contigs judged against the GENOME they were sequenced from (the synthetic genome is known: tools/gen_reads.py,
alga_amd/workload.py: device_build): every contig is placed on the genome or on its reverse complement over its FULL length on one
diagonal (the synthetic reads carry substitutions only, no indels) with at most `max_mismatch` of its positions differing.

  aligned      contigs with such a placement
  misjoined    contigs WITHOUT one although exact seeds of theirs hit the genome: the contig is a chimera of two loci (a false edge was
               followed) or carries more substitutions than the bound; `chimeric` of them have seeds on two or more distinct diagonals
               each backed by >= 2 seeds
  unplaced     no seed of the contig occurs in the genome at all
  genome_fraction   genome positions covered by an aligned contig / genome length
  duplicated_bp     aligned contig bp beyond the first cover of a position

This is what decides whether a higher N50 is a better assembly or a worse one (a false join raises N50): tools/score_supplement.py
reports it for the engine's graph and for the reference's, tests/test_gpu_score.py asserts engine misjoins <= reference misjoins and
genome fraction >= reference - 0.5 %.  Test infrastructure; nothing of the product imports it."""
import numpy as np

K = 24                       # seed length (48 bits)
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


def kmer_values(codes, k=K):
    """value of the k-mer starting at every position: sum codes[i + j] << 2j (uint64), len(codes) - k + 1 entries"""
    n = len(codes) - k + 1
    if n <= 0:
        return np.zeros(0, dtype=np.uint64)
    v = np.zeros(n, dtype=np.uint64)
    c = codes.astype(np.uint64)
    for j in range(k):
        v |= c[j:j + n] << np.uint64(2 * j)
    return v


class GenomeIndex:
    def __init__(self, genome_codes):
        self.g = np.ascontiguousarray(genome_codes, dtype=np.uint8)
        v = kmer_values(self.g)
        self.order = np.argsort(v, kind="stable").astype(np.int64)
        self.sorted = v[self.order]
        del v

    def hits(self, value, limit=8):
        lo = np.searchsorted(self.sorted, value, "left")
        hi = np.searchsorted(self.sorted, value, "right")
        return self.order[lo:min(hi, lo + limit)]


def _place(idx, c, max_mismatch):
    """-> (genome start or -1, mismatches, diagonals with >= 2 seeds, any seed hit)"""
    L = len(c)
    if L < K:
        return -1, 0, 0, False
    offs = np.unique(np.concatenate([np.arange(0, L - K + 1, 32), [L - K]]))
    vals = kmer_values(c)[offs]
    diag = {}
    for o, v in zip(offs, vals):
        for p in idx.hits(v):
            d = int(p) - int(o)
            diag[d] = diag.get(d, 0) + 1
    best, best_mm = -1, None
    G = len(idx.g)
    budget = int(max_mismatch * L)
    for d, _cnt in sorted(diag.items(), key=lambda kv: -kv[1])[:8]:
        if d < 0 or d + L > G:
            continue
        mm = int(np.count_nonzero(idx.g[d:d + L] != c))
        if mm <= budget and (best_mm is None or mm < best_mm):
            best, best_mm = d, mm
    return best, (best_mm or 0), sum(1 for x in diag.values() if x >= 2), bool(diag)


def genome_report(contigs, idx, max_mismatch=0.02):
    """contigs: iterable of bytes (ACGT).  idx: GenomeIndex."""
    G = len(idx.g)
    cover = np.zeros(G + 1, dtype=np.int32)
    rep = dict(contigs=0, aligned=0, misjoined=0, chimeric=0, unplaced=0, aligned_bp=0, misjoined_bp=0, mismatches=0, other_letters=0)
    mis_lens, al_lens = [], []
    for s in contigs:
        rep["contigs"] += 1
        c = _CODE[np.frombuffer(s, dtype=np.uint8)]
        if (c == 255).any():                                  # a letter outside ACGT: placed as a mismatch
            rep["other_letters"] += 1
            c = np.where(c == 255, 0, c).astype(np.uint8)
        L = len(c)
        fw = _place(idx, c, max_mismatch)
        rv = _place(idx, (3 - c)[::-1].copy(), max_mismatch)
        cand = [x for x in (fw, rv) if x[0] >= 0]
        if cand:
            d, mm = min(cand, key=lambda x: x[1])[:2]
            rep["aligned"] += 1
            rep["aligned_bp"] += L
            al_lens.append(L)
            rep["mismatches"] += mm
            cover[d] += 1
            cover[d + L] -= 1
        elif fw[3] or rv[3]:
            rep["misjoined"] += 1
            rep["misjoined_bp"] += L
            mis_lens.append(L)
            if max(fw[2], rv[2]) >= 2:
                rep["chimeric"] += 1
        else:
            rep["unplaced"] += 1
    depth = np.cumsum(cover[:-1])
    covered = int(np.count_nonzero(depth))
    rep["genome_fraction"] = covered / max(1, G)
    rep["duplicated_bp"] = int(rep["aligned_bp"] - covered)
    rep["misjoined_longest"] = int(max(mis_lens)) if mis_lens else 0
    rep["mismatch_rate_aligned"] = rep["mismatches"] / max(1, rep["aligned_bp"])
    al = np.sort(np.array(al_lens, dtype=np.int64))[::-1]
    tot = int(al.sum())
    rep["aligned_n50"] = int(al[np.searchsorted(np.cumsum(al), tot / 2)]) if tot else 0     # N50 once the false joins are taken out
    return rep
