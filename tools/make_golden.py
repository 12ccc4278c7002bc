#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Needs oracle/_ref/ALGA (built from the reference's own sources by `make -C oracle ref`; only
possible in the container that has /root/reference).  For every fixture it
  1. writes the synthetic input (tools/gen_reads.py, seeded) -> <name>[_1|_2].<ext>.gz
  2. runs   ALGA --file1=... [--file2=...] --threads=1 --serialize=1 --output=o.fasta [extra]
  3. stores the reference's `<TEST_NAME>_beforeSimplifier.graph` dump      -> <name>.graph.gz
     and the numbers the reference printed on stderr                       -> <name>.json
A fixture is data only: inputs and the reference's outputs.  Nothing of the reference's source
is stored.  tests/test_oracle_golden.py pins oracle/ against these; tests/test_gpu_parity.py pins
the HIP engine against them.
"""
import gzip
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import gen_reads  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "ALGA")
OUT = os.path.join(ROOT, "tests", "golden")
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)


def seqs_from_codes(codes, lens=None):
    txt = ALPHA[codes]
    n, length = codes.shape
    return [txt[i, : (length if lens is None else int(lens[i]))].tobytes().decode() for i in range(n)]


def write_records(path, seqs, fmt="fasta", suffix=""):
    with open(path, "w") as f:
        for i, s in enumerate(seqs):
            if fmt == "fasta":
                f.write(">r%d%s\n%s\n" % (i, suffix, s))
            elif fmt == "fastq":
                f.write("@r%d%s\n%s\n+\n%s\n" % (i, suffix, s, "I" * len(s)))
            else:
                f.write(s + "\n")


def run_reference(workdir, file1, file2=None, extra=()):
    cmd = [REF, "--file1=" + file1, "--threads=1", "--serialize=1", "--output=o.fasta"]
    if file2:
        cmd.append("--file2=" + file2)
    cmd += list(extra)
    p = subprocess.run(cmd, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace")
    log = p.stderr
    base = os.path.basename(file1)
    stem = base.rsplit(".", 1)[0] if "." in base else base
    graph = os.path.join(workdir, "ALGA_%s_scale55_noN_beforeSimplifier.graph" % stem)
    cands = [f for f in os.listdir(workdir) if f.endswith("_beforeSimplifier.graph")]
    if not os.path.exists(graph) and cands:
        graph = os.path.join(workdir, cands[0])
    meta = {"cmd": [os.path.basename(c) if c == REF else c for c in cmd], "returncode": p.returncode}
    m = re.search(r"MIN_OVERLAP_PREF_SUF: (-?\d+)", log)
    meta["min_overlap"] = int(m.group(1)) if m else None
    m = re.search(r"REMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP: (-?\d+)", log)
    meta["rsoemo"] = int(m.group(1)) if m else None
    # only the first creator run (the contig-trimming call at src/main.cpp:633-656 prints the same lines later)
    first = log.split("retainingOnlySmallestOffset")[0]
    meta["edges_after_iter"] = [[int(a), int(b)] for a, b in
                                re.findall(r"After Iteration (\d+) / \d+\.\s+There are already (\d+) edges", first)]
    m = re.search(r"Before first simplifier graph has (\d+) edges", log)
    meta["edges_before_simplifier"] = int(m.group(1)) if m else None
    m = re.search(r"There are (\d+) reads that are .* of other reads", log)
    meta["removed_prefix_reads"] = int(m.group(1)) if m else None
    m = re.search(r"Global::READS.size\(\): (\d+)\s*\nCreating GraphCreator", log)
    meta["nodes"] = int(m.group(1)) if m else None
    m = re.search(r"Removed dispensible reads - (\d+) reads were removed", log)
    meta["removed_short_reads"] = int(m.group(1)) if m else None
    return graph, meta, log


def gz_copy(src, dst):
    with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
        shutil.copyfileobj(fi, fo)


def emit(name, files, extra=(), fmt="fasta"):
    """files: list of (suffix, seqs) -- one entry (single-end) or two (paired)."""
    ext = {"fasta": "fasta", "fastq": "fastq", "plain": "txt"}[fmt]
    with tempfile.TemporaryDirectory() as wd:
        paths = []
        for k, (suf, seqs) in enumerate(files):
            fn = "%s%s.%s" % (name, "_%d" % (k + 1) if len(files) > 1 else "", ext)
            write_records(os.path.join(wd, fn), seqs, fmt, suf)
            paths.append(fn)
        graph, meta, log = run_reference(wd, paths[0], paths[1] if len(paths) > 1 else None, extra)
        if not os.path.exists(graph):
            sys.stderr.write(log[-3000:])
            raise SystemExit("reference produced no graph dump for %s" % name)
        meta["inputs"] = [p + ".gz" for p in paths]
        meta["extra_args"] = list(extra)
        meta["graph"] = name + ".graph.gz"
        meta["graph_bytes"] = os.path.getsize(graph)
        for p in paths:
            gz_copy(os.path.join(wd, p), os.path.join(OUT, p + ".gz"))
        gz_copy(graph, os.path.join(OUT, name + ".graph.gz"))
        with open(os.path.join(OUT, name + ".json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print("%-14s nodes=%s edges=%s Lmin=%s rsoemo=%s" % (name, meta["nodes"], meta["edges_before_simplifier"],
                                                             meta["min_overlap"], meta["rsoemo"]))


def main():
    if not os.path.exists(REF):
        raise SystemExit("build the reference first: make -C oracle ref")
    os.makedirs(OUT, exist_ok=True)

    # F1: BASELINE.json configs[0]: 10k x 100 bp, error-free, genome 20 kb, seed 1
    c, _ = gen_reads.sample_reads(10000, 100, 20000, 1)
    emit("f1_cfg1", [("", seqs_from_codes(c))])

    # F2: 150 bp with 2 % substitutions: exercises the small-overlap cap ties (SOES) and spurious vias
    c, _ = gen_reads.sample_reads(6000, 150, 18000, 2, err=0.02)
    emit("f2_err2", [("", seqs_from_codes(c))])

    # F3: paired files (--file1/--file2), groups of four nodes
    a, b = gen_reads.sample_pairs(2500, 150, 15000, 6)
    emit("f3_paired", [("/1", seqs_from_codes(a)), ("/2", seqs_from_codes(b))])

    # F4: variable read lengths 100..150 with contained reads: pins equal-offset / right-offset rules
    c, l = gen_reads.sample_reads(5000, 150, 12000, 4, min_length=100)
    emit("f4_varlen", [("", seqs_from_codes(c, l))])

    # F5: messy FASTQ: reads with N, periodic (STR) reads, exact duplicates, reads below the length
    #     threshold, lower-case-free U bases absent; a repeat-rich genome (two copies of a 3 kb block)
    rng = np.random.default_rng(5)
    block = rng.integers(0, 4, size=3000, dtype=np.uint8)
    genome = np.concatenate([rng.integers(0, 4, size=4000, dtype=np.uint8), block,
                             rng.integers(0, 4, size=4000, dtype=np.uint8), block,
                             rng.integers(0, 4, size=2000, dtype=np.uint8)])
    seqs = []
    for i in range(6000):
        ln = int(rng.integers(60, 151))
        st = int(rng.integers(0, len(genome) - ln + 1))
        cc = genome[st:st + ln]
        if rng.random() < 0.5:
            cc = (3 - cc)[::-1]
        s = ALPHA[cc].tobytes().decode()
        r = rng.random()
        if r < 0.03:
            p = int(rng.integers(0, ln)); s = s[:p] + "N" + s[p + 1:]
        elif r < 0.06:
            unit = s[: int(rng.integers(1, 21))]; s = (unit * (ln // len(unit) + 1))[:ln]
        elif r < 0.10 and seqs:
            s = seqs[int(rng.integers(0, len(seqs)))]
        elif r < 0.13:
            s = s[: int(rng.integers(12, 50))]
        seqs.append(s)
    emit("f5_messy", [("", seqs)], fmt="fastq")

    # F6: explicit -l / --rsoemo on error-free 100 bp reads (parameter plumbing of src/main.cpp:112-115)
    c, _ = gen_reads.sample_reads(4000, 100, 9000, 8)
    emit("f6_l40", [("", seqs_from_codes(c))], extra=["-l", "40", "--rsoemo=70"])


if __name__ == "__main__":
    main()
