#!/usr/bin/env python3
"""Golden vectors for the first simplifier step: the reference's own
Graph::sortEdgesByIncreasingOffset + GraphSimplifier::cutNonAndWeaklyMetricTriangles run on the committed graph dumps through
oracle/_ref/ref_driver (mode `triangles`) -> tests/golden/<name>.aftercut.graph.gz.  Needs /root/reference (oracle/Makefile).
usage: tools/make_golden_n3.py"""
import gzip
import json
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DRV = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
MOPP = 262            # Params::MAX_OFFSET_PARALLEL_PATHS = max(250, int(1.75 * LEN)), LEN = 150 (src/main.cpp:93-96, src/Params.cpp:687)
SETS = {"f1_cfg1": ("f1_cfg1.graph.gz", 250), "f2_err2": ("f2_err2.graph.gz", MOPP), "f4_varlen": ("f4_varlen.graph.gz", MOPP),
        "f5_messy": ("f5_messy.graph.gz", MOPP), "f7_pkb": ("f7_pkb.supplement.graph.gz", MOPP)}


def main():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    meta = {}
    with tempfile.TemporaryDirectory() as wd:
        for name, (src, mopp) in SETS.items():
            gin, gout = os.path.join(wd, "in.graph"), os.path.join(wd, "out.graph")
            with gzip.open(os.path.join(GOLD, src), "rb") as f, open(gin, "wb") as g:
                g.write(f.read())
            log = subprocess.run([DRV, "triangles", gin, gout, str(mopp)], capture_output=True, text=True, check=True).stdout
            w = log.split()
            meta[name] = dict(graph_in=src, max_offset_parallel_paths=mopp, edges_before=int(w[1]), edges_after=int(w[3]))
            with open(gout, "rb") as g, gzip.GzipFile(os.path.join(GOLD, name + ".aftercut.graph.gz"), "wb", mtime=0) as f:
                f.write(g.read())
            print(name, meta[name])
    json.dump(meta, open(os.path.join(GOLD, "n3_aftercut.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
