#!/usr/bin/env python3
"""Contig-level score of the approximate path (error_rate > 0.01; SURVEY.md section 8(c) fixture F5, BASELINE configs[4]).

The engine's supplement is order independent, the reference's depends on the order in which it walks its k-mer groups and on
std::sort's tie order: the two post-supplement graphs differ in a few per cent of their edges.  What that does
DOWNSTREAM is measured here on the same reads:
  engine : alga_hip --error_rate=R (graph on the GPU: exact path + supplement) -> stock ALGA --deserialize_graph=1 (unchanged
           simplifier + contig stages) -> contigs
  ref #1 : stock ALGA --error_rate=R --threads=T, the whole pipeline
  ref #2 : the same command again -- with T > 1 the reference races on ties: the distance between two of
           its own runs is the noise floor the engine's distance has to be read against
and reported as contig statistics (count, total bp, N50, longest) plus the share of contig bp that sits in contigs found
IDENTICALLY (up to strand) by both sides of a pair -- and, the referee that knows the truth, against the GENOME the reads were drawn
from (tools/genome_score.py): every contig of either side is placed on the synthetic genome or its reverse complement over its full
length with <= 2 % mismatches; reported per side: aligned / misjoined (chimeric) / unplaced contigs, genome fraction, duplicated bp,
N50 of the aligned contigs.  A false join raises N50; it also shows up here as a misjoined contig.

usage: tools/score_supplement.py [n_reads=1000000] [genome=3*n] [threads=16] [--seed S]
Runs on the GPU box (oracle/_ref/ALGA and alga_amd/bin/alga_hip travel with the repository snapshot)."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

COMP = bytes.maketrans(b"ACGT", b"TGCA")


def read_contigs(path):
    seqs, cur = [], []
    with open(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if cur:
                    seqs.append(b"".join(cur))
                cur = []
            else:
                cur.append(line.strip())
    if cur:
        seqs.append(b"".join(cur))
    return seqs


def canon(s):
    r = s.translate(COMP)[::-1]
    return s if s <= r else r


def stats(seqs):
    lens = np.sort(np.array([len(s) for s in seqs], dtype=np.int64))[::-1]
    total = int(lens.sum())
    n50 = int(lens[np.searchsorted(np.cumsum(lens), total / 2)]) if total else 0
    return dict(contigs=len(seqs), total_bp=total, n50=n50, longest=int(lens[0]) if len(lens) else 0)


def shared(a, b):
    """bp of a's contigs that b holds identically (either strand), as a share of a's bp"""
    sb = set(canon(s) for s in b)
    tot = sum(len(s) for s in a)
    hit = sum(len(s) for s in a if canon(s) in sb)
    return hit / max(1, tot)


def run(cmd, cwd):
    t = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
    if r.returncode != 0:
        raise SystemExit("failed: %s\n%s" % (" ".join(cmd), r.stderr[-3000:]))
    return time.perf_counter() - t, r.stderr


def score(n, G, T=16, seed=13, rate=0.02, ref_runs=2):
    import gen_reads
    from alga_amd import workload
    exe_ref = os.path.join(ROOT, "oracle", "_ref", "ALGA")
    exe_hip = os.path.join(ROOT, "alga_amd", "bin", "alga_hip")
    out = dict(reads=n, genome=G, error_rate=rate, threads=T, seed=seed)
    names = ["engine"] + ["ref%d" % (k + 1) for k in range(ref_runs)]
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as wd:
        if n >= 4_000_000:
            wl = workload.device_build(n, 150, G, seed, err=rate, sample_reads=n, return_genome=True)
            codes, genome = wl["sample_codes"], wl["genome_codes"]
            del wl
        else:
            codes, _ = gen_reads.sample_reads(n, 150, G, seed, rate)
            genome = gen_reads.make_genome(G, np.random.default_rng(seed))      # the generator's first draw (gen_reads.sample_reads)
        fasta = os.path.join(wd, "s.fasta")
        workload.write_fasta_fast(fasta, codes)
        out["reads_written"] = int(len(codes))
        del codes
        dirs = {}
        for k in names:
            dirs[k] = os.path.join(wd, k)
            os.mkdir(dirs[k])
        keep = ("Before supplement", "After supplement", "Before first simplifier")
        t, err = run([exe_hip, "--file1=" + fasta, "--threads=%d" % T, "--error_rate=%g" % rate, "--output=c.fasta", "--alga=" + exe_ref], dirs["engine"])
        out["engine_wall_s"] = t
        out["engine_log"] = [line.strip() for line in err.splitlines() if line.startswith(keep)]
        for k in names[1:]:
            t, err = run([exe_ref, "--file1=" + fasta, "--threads=%d" % T, "--error_rate=%g" % rate, "--output=c.fasta"], dirs[k])
            out[k + "_wall_s"] = t
            out[k + "_log"] = [line.strip() for line in err.splitlines() if line.startswith(keep)]
        C = {k: read_contigs(os.path.join(dirs[k], "c.fasta")) for k in dirs}
    for k, v in C.items():
        out[k] = stats(v)
    import genome_score
    idx = genome_score.GenomeIndex(genome)
    out["genome"] = {k: genome_score.genome_report(v, idx) for k, v in C.items()}
    rel = lambda a, b: abs(a - b) / max(1, b)      # noqa: E731
    out["bp_in_identical_contigs"] = {"engine_in_ref1": shared(C["engine"], C["ref1"]), "ref1_in_engine": shared(C["ref1"], C["engine"])}
    out["relative_difference"] = {"engine_vs_ref1": {m: rel(out["engine"][m], out["ref1"][m]) for m in ("contigs", "total_bp", "n50")}}
    if ref_runs > 1:
        out["bp_in_identical_contigs"].update({"ref2_in_ref1": shared(C["ref2"], C["ref1"]), "ref1_in_ref2": shared(C["ref1"], C["ref2"])})
        out["relative_difference"]["ref2_vs_ref1"] = {m: rel(out["ref2"][m], out["ref1"][m]) for m in ("contigs", "total_bp", "n50")}
    return out


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    seed = int(sys.argv[sys.argv.index("--seed") + 1]) if "--seed" in sys.argv else 13
    if "--seed" in sys.argv:
        argv = [a for a in argv if a != str(seed)] if str(seed) in argv[3:] else argv
    n = int(argv[0]) if len(argv) > 0 else 1_000_000
    G = int(argv[1]) if len(argv) > 1 else 3 * n
    T = int(argv[2]) if len(argv) > 2 else 16
    print(json.dumps(score(n, G, T, seed)))


if __name__ == "__main__":
    main()
