"""Synthetic workloads for bench.py and the full-size tests (SURVEY.md section 8(d) generator).

`make_nodes` reproduces, with numpy, what the reference's input stages hand to the graph creator for
FIXED-LENGTH error-free or noisy reads without N: trim 3 nt per side (src/IO/InputReader.cpp:298-303),
reverse-complement doubling with node 2i = revcomp / 2i+1 = forward (:78-80, :363-377), removal of duplicate
reads together with their twins (src/IO/ReadPreprocess.cpp:13-77: of identical sequences the one with the
largest id survives) and id compaction (src/main.cpp:150-232).  tests/test_workload_cpu.py checks it node for
node against the oracle's literal ingest of the same reads written as FASTA.
"""
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(_HERE), "tools"))
import gen_reads  # noqa: E402

from .engine import pack_reads, derive_params  # noqa: E402

CONFIGS = {
    # name: (reads, read length, genome length, seed, substitution rate)      BASELINE.json configs[0..4]
    "cfg1_10k_100bp": (10_000, 100, 20_000, 1, 0.0),
    "cfg2_1M_150bp": (1_000_000, 150, 3_000_000, 3, 0.0),
    "cfg3_5M_150bp": (5_000_000, 150, 4_600_000, 7, 0.0),
    "cfg4_50M_150bp": (50_000_000, 150, 250_000_000, 11, 0.0),
    "cfg5_10M_150bp_err2": (10_000_000, 150, 30_000_000, 13, 0.02),
    # beyond BASELINE: the north-star shape at 2x and 4x the reads (same 30x coverage) -- how far one GPU's 288 GB goes
    "x2_100M_150bp": (100_000_000, 150, 500_000_000, 17, 0.0),
    "x4_200M_150bp": (200_000_000, 150, 1_000_000_000, 19, 0.0),
}


def min_period_le(codes, thr=20):
    """rows whose minimal period is <= thr (the reference drops them as STR reads, InputReader.cpp:341-354)."""
    n, m = codes.shape
    out = np.zeros(n, dtype=bool)
    for p in range(1, thr + 1):
        out |= (codes[:, p:] == codes[:, :-p]).all(axis=1)
    return out


def make_nodes(codes, trim=3, stride_words=None):
    """codes[n, L] uint8 (A0 C1 G2 T3) as sampled -> (words[N, stride] u32, lens[N] i32, kept_read_ids)."""
    n, L = codes.shape
    if trim and L >= 2 * trim + 10:
        codes = codes[:, trim: L - trim]
    m = codes.shape[1]
    keep = ~min_period_le(codes)
    rc = (3 - codes)[:, ::-1]
    fw = pack_reads(codes)
    rv = pack_reads(rc)
    # canonical key = lexicographically smaller of (forward, revcomp) words as bytes: reads with the same key
    # are duplicates of each other on one strand or the other; the reference keeps the one with the largest id
    W = fw.shape[1]
    # compare as big-endian byte strings: any total order works as long as it is the same for both strands
    lt = np.zeros(n, dtype=bool)
    fbytes = fw.view(np.uint8).reshape(n, 4 * W)
    rbytes = rv.view(np.uint8).reshape(n, 4 * W)
    diff = fbytes != rbytes
    first = diff.argmax(axis=1)
    anyd = diff.any(axis=1)
    idx = np.arange(n)
    lt[anyd] = fbytes[idx[anyd], first[anyd]] < rbytes[idx[anyd], first[anyd]]
    canon = np.where(lt[:, None], fw, rv)
    palin = ~anyd                                   # read == its own reverse complement: the reference loses both nodes
    cb = np.ascontiguousarray(canon).view(np.dtype((np.void, 4 * W))).ravel()
    valid = np.flatnonzero(keep)                    # STR reads are nullptr before the sort and never take part
    cbk = cb[valid]
    o = np.argsort(cbk, kind="stable")
    s = cbk[o]
    last = np.ones(len(valid), dtype=bool)
    last[:-1] = s[1:] != s[:-1]
    survivors = np.zeros(n, dtype=bool)
    survivors[valid[o[last]]] = True
    survivors &= ~palin
    ids = np.flatnonzero(survivors)
    N = 2 * len(ids)
    if stride_words is None:
        stride_words = W
    elif stride_words == "aligned":                 # the engine's own HBM layout (prefsuf_common.h: hbm_row_stride)
        stride_words = 4 if W <= 4 else (8 if W <= 8 else (W + 15) & ~15)
    words = np.zeros((N, stride_words), dtype=np.uint32)
    words[0::2, :W] = rv[ids]
    words[1::2, :W] = fw[ids]
    lens = np.full(N, m, dtype=np.int32)
    return words, lens, ids


def build(config, scale=1, stride_words=None):
    """-> dict(words, lens, n_reads, read_len, min_overlap, rsoemo, name).  `scale` multiplies reads AND genome
    (coverage fixed): the weak-scaling workload for `scale` GPUs."""
    n, L, G, seed, err = CONFIGS[config]
    n, G = n * scale, G * scale
    codes, _ = gen_reads.sample_reads(n, L, G, seed, err)
    words, lens, ids = make_nodes(codes, stride_words=stride_words)
    lo, rs = derive_params(float(L - 6))
    return dict(words=words, lens=lens, n_reads=n, read_len=L, genome=G, seed=seed, err=err, min_overlap=lo, rsoemo=rs,
                name=config, codes=codes, kept=ids)


def write_fasta_fast(path, codes):
    """one record per read, fixed-width header, vectorised (150 MB/s-ish)."""
    n, L = codes.shape
    hdr = np.frombuffer((">r%09d\n" % 0).encode(), dtype=np.uint8)
    H = len(hdr)
    rec = np.empty((n, H + L + 1), dtype=np.uint8)
    rec[:, :H] = hdr
    digits = np.arange(n)[:, None] // (10 ** np.arange(8, -1, -1))[None, :] % 10
    rec[:, 2:11] = (digits + 48).astype(np.uint8)
    rec[:, H:H + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[codes]
    rec[:, H + L] = 10
    with open(path, "wb") as f:
        f.write(rec.tobytes())


def write_fasta_chunked(path, n, L, G, seed, chunk=1 << 20):
    """n error-free reads of L nt over an iid genome of G nt straight to a FASTA file, a chunk of reads at a time (the 50 M-read
    north-star file is 8 GB: never held in memory).  Same distribution as gen_reads.sample_reads, not the same draws."""
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    ar = np.arange(L)[None, :]
    with open(path, "wb") as f:
        for s0 in range(0, n, chunk):
            m = min(chunk, n - s0)
            st = rng.integers(0, G - L + 1, m)
            codes = genome[st[:, None] + ar]
            fl = rng.random(m) < 0.5
            codes[fl] = (3 - codes[fl])[:, ::-1]
            rec = np.empty((m, 12 + L + 1), dtype=np.uint8)          # ">" + 10 digits + "\n" + sequence + "\n"
            rec[:, 0] = ord(">")
            ids = np.arange(s0, s0 + m)
            for k in range(10):
                rec[:, 10 - k] = ord("0") + (ids // 10 ** k) % 10
            rec[:, 11] = ord("\n")
            rec[:, 12:12 + L] = lut[codes]
            rec[:, 12 + L] = ord("\n")
            f.write(rec.tobytes())


def _mix64(x):
    """splitmix-style mixer on int64 tensors (wrapping arithmetic, logical shifts emulated)"""
    x = x * -7046029254386353131                           # 0x9E3779B97F4A7C15
    x = x ^ ((x >> 32) & 0xFFFFFFFF)
    x = x * -4658895280553007687                           # 0xBF58476D1CE4E5B9
    x = x ^ ((x >> 29) & 0x7FFFFFFFF)
    return x


def _with_errors(codes, read_idx, err, seed):
    """substitution errors at rate `err` as a pure function of (seed, original read index, position): regenerable in any order"""
    import torch
    L = codes.shape[1]
    x = _mix64((read_idx[:, None] * L + torch.arange(L, device=codes.device)[None, :]) ^ (int(seed) * 1000003))
    u = ((x >> 11) & ((1 << 40) - 1)).to(torch.float64) / float(1 << 40)
    delta = (1 + ((x >> 3) & 0xFFFF) % 3).to(torch.uint8)
    return torch.where(u < err, (codes + delta) & 3, codes)


def device_build(n_reads, read_len, genome_len, seed, device="cuda", trim=3, chunk=1 << 21, sample_reads=0, err=0.0, return_genome=False):
    """The SURVEY.md section 8(d) workload for error-free fixed-length reads, generated ON THE DEVICE (torch is only the array
    library) straight into the engine's HBM row layout: iid genome, uniform starts, strand flips; one read per start
    position (on an iid genome two reads are duplicates of each other -- on one strand or the other -- exactly when they
    start at the same position, which is what the reference's duplicate removal keeps one of: src/IO/ReadPreprocess.cpp:13-77;
    STR reads and reverse-complement palindromes have probability < 4^-20 per read and are ignored); node ids shuffled so that
    they carry no positional information, node 2i = reverse complement, 2i+1 = forward (src/IO/InputReader.cpp:78-80).
    err > 0: substitution errors at that rate per nucleotide (applied in genome orientation, before the strand flip); two reads
    are then duplicates when they start at the same position AND carry the same errors (in the part that survives the end
    trimming), so the set is made unique on (start, hash of that erroneous sequence) instead of on the start alone.
    -> dict(words int32[N, 16|stride] (device), lens int32[N] (device), n_reads, unique_reads, min_overlap, rsoemo,
            sample_codes uint8[~sample_reads, read_len] (host): a genomic window of the SAME read set for the CPU baseline's
            FASTA -- every (untrimmed) read that starts in the first genome_len * sample_reads / n_reads positions, in node
            order: same coverage, same overlap statistics per read as the whole set)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    genome = torch.randint(0, 4, (genome_len,), dtype=torch.uint8, device=device, generator=g)
    starts = torch.randint(0, genome_len - read_len + 1, (n_reads,), device=device, generator=g)
    flip = torch.rand(n_reads, device=device, generator=g) < 0.5
    ridx = None
    if err > 0:
        # the key of a read: start and a 64-bit hash of its erroneous sequence
        trm = read_len - 2 * trim if read_len >= 2 * trim + 10 else read_len
        tr0 = trim if trm != read_len else 0
        ridx = torch.arange(n_reads, device=device)
        hsh = torch.empty(n_reads, dtype=torch.int64, device=device)
        wts = _mix64(torch.arange(read_len, device=device) + 12345)[None, :]
        arL = torch.arange(read_len, device=device)[None, :]
        for s0 in range(0, n_reads, chunk):
            c = _with_errors(genome[starts[s0:s0 + chunk][:, None] + arL], ridx[s0:s0 + chunk], err, seed)
            hsh[s0:s0 + chunk] = ((c.to(torch.int64) + 1) * wts)[:, tr0:tr0 + trm].sum(dim=1)   # of what is left after the end trimming
            del c
        o1 = torch.argsort(hsh, stable=True)
        o2 = torch.argsort(starts[o1], stable=True)
        order = o1[o2]
        starts, hsh = starts[order], hsh[order]
        first = torch.ones_like(starts, dtype=torch.bool)
        first[1:] = (starts[1:] != starts[:-1]) | (hsh[1:] != hsh[:-1])
        starts, flip, ridx = starts[first], flip[order][first], ridx[order][first]
        del o1, o2, order, first, hsh
    else:
        starts, order = torch.sort(starts, stable=True)
        first = torch.ones_like(starts, dtype=torch.bool)
        first[1:] = starts[1:] != starts[:-1]
        starts, flip = starts[first], flip[order][first]
        del order, first
    R = int(starts.shape[0])
    perm = torch.randperm(R, device=device, generator=g)
    starts, flip = starts[perm], flip[perm]
    if ridx is not None:
        ridx = ridx[perm]
    del perm
    m = read_len - 2 * trim if read_len >= 2 * trim + 10 else read_len
    t0 = trim if m != read_len else 0
    W = (2 * m + 31) // 32
    stride = 4 if W <= 4 else (8 if W <= 8 else (W + 15) & ~15)
    words = torch.zeros((2 * R, stride), dtype=torch.int32, device=device)
    pad = W * 16 - m
    shifts = (2 * torch.arange(16, device=device, dtype=torch.int64))[None, None, :]

    def pack(codes):
        if pad:
            codes = torch.cat([codes, torch.zeros((codes.shape[0], pad), dtype=codes.dtype, device=device)], dim=1)
        x = (codes.view(-1, W, 16).to(torch.int64) << shifts).sum(dim=2)
        return torch.where(x >= 2 ** 31, x - 2 ** 32, x).to(torch.int32)

    ar = torch.arange(m, device=device)[None, :]
    for s0 in range(0, R, chunk):
        st = starts[s0:s0 + chunk]
        if err > 0:
            full = _with_errors(genome[st[:, None] + torch.arange(read_len, device=device)[None, :]], ridx[s0:s0 + chunk], err, seed)
            codes = full[:, t0:t0 + m]
            del full
        else:
            codes = genome[(st[:, None] + t0 + ar)]
        f = flip[s0:s0 + chunk][:, None]
        rc = (3 - codes).flip(1)
        fw = torch.where(f, rc, codes)
        rv = torch.where(f, codes, rc)
        words[2 * s0 + 1: 2 * (s0 + st.shape[0]) + 1: 2, :W] = pack(fw)
        words[2 * s0: 2 * (s0 + st.shape[0]): 2, :W] = pack(rv)
        del codes, rc, fw, rv
    lens = torch.full((2 * R,), m, dtype=torch.int32, device=device)
    lo, rs = derive_params(float(m))
    out = dict(words=words, lens=lens, n_reads=n_reads, unique_reads=R, read_len=read_len, genome=genome_len, seed=seed,
               min_overlap=lo, rsoemo=rs, sample_codes=None)
    if sample_reads:
        window = int(genome_len * min(1.0, sample_reads / max(1, n_reads)))
        sel = torch.nonzero(starts < window).flatten()
        st = starts[sel]
        codes = genome[(st[:, None] + torch.arange(read_len, device=device)[None, :])]
        if err > 0:
            codes = _with_errors(codes, ridx[sel], err, seed)
        codes = torch.where(flip[sel][:, None], (3 - codes).flip(1), codes)
        out["sample_codes"] = codes.cpu().numpy()
    if return_genome:
        out["genome_codes"] = genome.cpu().numpy()          # the truth the contigs are scored against (tools/genome_score.py)
    return out
