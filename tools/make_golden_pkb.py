#!/usr/bin/env python3
"""Golden vectors for the approximate supplement, produced by the REFERENCE'S OWN CODE:
oracle/_ref/ref_driver is my main() linked against the reference's objects (see oracle/ref_driver.cpp).

Fixture f7_pkb (reads: the 2 %-error set of f2_err2):
  f7_pkb.nodes.bin.gz        node set handed to the creators (oracle ingest of the f2_err2 input)
  f7_pkb.supplement.graph.gz graph after the reference's supplement (GraphCreatorLI, 4 rounds, --threads=1) run on
                             the f2_err2 dump; cross-checked against the edge count stock ALGA prints with
                             --error_rate=0.02 ("After supplement G has E edges", src/main.cpp:352)
  f7_pkb.canalign.npz        20 000 (r1, r2, offset) triples + AlignmentControllerHybrid::canAlign verdicts
  f7_pkb.likmers.bin.gz      Read::getKmers (LI minimizers, k=35, 6 intervals) of 400 nodes under the 4 priority rotations
  f7_pkb.json                parameters the supplement ran with
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402

DRV = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
REF = os.path.join(ROOT, "oracle", "_ref", "ALGA")
OUT = os.path.join(ROOT, "tests", "golden")


def write_nodes(path, words, lens):
    n, W = words.shape
    with open(path, "wb") as f:
        f.write(np.array([n, W], dtype=np.int32).tobytes())
        f.write(lens.astype(np.int32).tobytes())
        f.write(np.ascontiguousarray(words, dtype=np.uint32).tobytes())


def gz(src, dst):
    with open(src, "rb") as fi, gzip.GzipFile(dst, "wb", mtime=0) as fo:
        shutil.copyfileobj(fi, fo)


def main():
    fx = O.Fixture(OUT, "f2_err2")
    f1, _ = fx.inputs()
    nd = O.ingest(f1)
    with tempfile.TemporaryDirectory() as wd:
        nodes = os.path.join(wd, "nodes.bin")
        write_nodes(nodes, nd["words"], nd["len"])
        gin = os.path.join(wd, "in.graph")
        open(gin, "wb").write(fx.ref_graph())
        gout = os.path.join(wd, "out.graph")
        r = subprocess.run([DRV, "supplement", nodes, gin, gout, "2", str(nd["li_kmer_length"])], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, check=True)
        m = re.search(r"MIN_OVERLAP_AREA (\d+) MAX_OFFSET (\d+) MIN_IDENTITY (\d+) avg ([\d.]+) edges_before (\d+)", r.stdout)
        moa, mo, mi, avg, eb = int(m.group(1)), int(m.group(2)), int(m.group(3)), float(m.group(4)), int(m.group(5))
        ea = int(re.search(r"edges_after (\d+)", r.stdout).group(1))
        # cross-check the driver against the stock binary's own count
        shutil.copy(f1, os.path.join(wd, "x.fasta"))
        p = subprocess.run([REF, "--file1=x.fasta", "--threads=1", "--error_rate=0.02", "--output=o.fasta"], cwd=wd,
                           stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
        stock = int(re.search(r"After supplement G has (\d+) edges", p.stderr).group(1))
        stock_before = int(re.search(r"Before supplement, G has (\d+) edges", p.stderr).group(1))
        assert stock_before == eb, (stock_before, eb)
        assert stock == ea, "driver flow differs from src/main.cpp:300-352: %d vs %d" % (stock, ea)
        n, sup = O.parse_graph(open(gout, "rb").read())
        # canAlign triples: neighbourhoods of real supplement edges (accepts and near-misses) + uniform random ones
        rng = np.random.default_rng(17)
        live = np.flatnonzero(nd["len"] > 0)
        e = sup[rng.integers(0, len(sup), size=12000)]
        t1 = np.stack([e[:, 0], e[:, 1], np.clip(e[:, 2] + rng.integers(-3, 4, size=len(e)), 0, None)], axis=1)
        t2 = np.stack([live[rng.integers(0, len(live), 4000)], live[rng.integers(0, len(live), 4000)], rng.integers(0, 60, 4000)], axis=1)
        t3 = np.stack([e[:4000, 1], e[:4000, 0], e[:4000, 2]], axis=1)
        tri = np.concatenate([t1, t2, t3]).astype(np.int32)
        tpath, opath = os.path.join(wd, "tri.bin"), os.path.join(wd, "tri.out")
        tri.tofile(tpath)
        subprocess.run([DRV, "canalign", nodes, tpath, opath, str(moa), str(mo), str(mi)], check=True, stderr=subprocess.DEVNULL)
        verdict = np.fromfile(opath, dtype=np.uint8)
        assert len(verdict) == len(tri)
        # LI k-mers of the first 400 nodes
        sub = os.path.join(wd, "sub.bin")
        write_nodes(sub, nd["words"][:400], nd["len"][:400])
        kout = os.path.join(wd, "km.bin")
        subprocess.run([DRV, "likmers", sub, kout, "35", "6"], check=True, stderr=subprocess.DEVNULL)
        gz(nodes, os.path.join(OUT, "f7_pkb.nodes.bin.gz"))
        gz(gout, os.path.join(OUT, "f7_pkb.supplement.graph.gz"))
        gz(kout, os.path.join(OUT, "f7_pkb.likmers.bin.gz"))
        np.savez_compressed(os.path.join(OUT, "f7_pkb.canalign.npz"), triples=tri, verdict=verdict)
        meta = dict(min_overlap_area=moa, max_offset_pct=mo, min_identity_pct=mi, avg_len=avg, edges_before=eb, edges_after=ea,
                    kmer_length_bucket=int(nd["li_kmer_length"]), li_k=35, li_intervals=6, same_ends=3, error_rate_percent=2,
                    pre_graph="f2_err2.graph.gz", accepted=int(verdict.sum()), triples=int(len(tri)), likmer_nodes=400)
        json.dump(meta, open(os.path.join(OUT, "f7_pkb.json"), "w"), indent=1, sort_keys=True)
        print(meta)
    fx.cleanup()


if __name__ == "__main__":
    main()
