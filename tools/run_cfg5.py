#!/usr/bin/env python3
"""Config-5-style run (reads with substitution errors, --error_rate=0.02): exact path + approximate supplement on the GPU,
next to the reference binary on the same reads; reports times, edge counts and the size of the symmetric difference
when the reference's dump is available.  usage: tools/run_cfg5.py [n_reads] [genome] [threads]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
import gen_reads  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else n * 3
    threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    codes, _ = gen_reads.sample_reads(n, 150, G, 13, 0.02)
    words, lens, ids = workload.make_nodes(codes, stride_words=12)
    lo, rs = alga_amd.derive_params(144.0)
    eng = alga_amd.Engine(0)
    t0 = time.perf_counter()
    pre = eng.prefsuf_host(words, lens, lo, rs)
    t1 = time.perf_counter()
    st = eng.last_stats()
    p = eng.pkb_params(float(lens[lens > 0].mean()), 0.02, min(2 * lo // 3, 60))
    post = eng.pkb_supplement_host(words, lens, pre, p)
    t2 = time.perf_counter()
    ps = eng.pkb_last_stats()
    warm = []
    for _ in range(int(os.environ.get("ALGA_CFG5_REPEAT", "2"))):       # buffers are allocated by now: steady-state device times
        eng.prefsuf_host(words, lens, lo, rs)
        e_ms = eng.last_stats()["ms_total"]
        eng.pkb_supplement_host(words, lens, pre, p)
        warm.append((e_ms, eng.pkb_last_stats()["ms_total"]))
    out = dict(reads=n, nodes=int(len(lens)), edges_exact=int(len(pre)), edges_after_supplement=int(len(post)),
               gpu_exact_wall_s=t1 - t0, gpu_exact_device_ms=st["ms_total"], gpu_supplement_wall_s=t2 - t1,
               gpu_supplement_device_ms=ps["ms_total"], supplement=ps,
               gpu_exact_device_ms_warm=[w[0] for w in warm], gpu_supplement_device_ms_warm=[w[1] for w in warm])
    exe = os.path.join(ROOT, "oracle", "_ref", "ALGA")
    skip_ref = os.environ.get("ALGA_SKIP_REF") == "1"
    if os.path.exists(exe) and not skip_ref:
        with tempfile.TemporaryDirectory() as wd:
            workload.write_fasta_fast(os.path.join(wd, "s.fasta"), codes)
            t = time.perf_counter()
            r = subprocess.run([exe, "--file1=s.fasta", "--threads=%d" % threads, "--error_rate=0.02", "--output=o.fasta"], cwd=wd,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
            out["ref_total_wall_s"] = time.perf_counter() - t
            m = re.search(r"Before supplement, G has (\d+) edges", r.stderr)
            out["ref_edges_exact"] = int(m.group(1)) if m else None
            m = re.search(r"After supplement G has (\d+) edges", r.stderr)
            out["ref_edges_after_supplement"] = int(m.group(1)) if m else None
            for key in ("GraphCreator PrefSuf", "GraphCreator PKB Supplement"):
                m = re.search(re.escape(key) + r"[^\d\n]*([\d.]+)", r.stderr)
                out["ref_cpu_seconds_" + key.split()[-1]] = float(m.group(1)) if m else None
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv) and not skip_ref:
        # the reference's own supplement code (through oracle/ref_driver.cpp, --threads=1 order) on the SAME nodes and the SAME
        # pre-supplement graph: the symmetric difference is the effect of the reference's order dependence alone
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O
        with tempfile.TemporaryDirectory() as wd:
            nodes = os.path.join(wd, "nodes.bin")
            with open(nodes, "wb") as f:
                f.write(np.array([len(lens), words.shape[1]], dtype=np.int32).tobytes())
                f.write(lens.astype(np.int32).tobytes())
                f.write(np.ascontiguousarray(words, dtype=np.uint32).tobytes())
            gin, gout = os.path.join(wd, "in.graph"), os.path.join(wd, "out.graph")
            open(gin, "wb").write(O.graph_bytes(len(lens), pre))
            t = time.perf_counter()
            subprocess.run([drv, "supplement", nodes, gin, gout, "2", str(min(2 * lo // 3, 60))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            out["ref_driver_supplement_wall_s"] = time.perf_counter() - t
            _, ref_post = O.parse_graph(open(gout, "rb").read())
            a = set(map(tuple, ref_post.tolist())); b = set(map(tuple, post.tolist()))
            out["ref_driver_edges_after_supplement"] = len(a)
            out["supplement_symmetric_difference"] = len(a ^ b)
            out["supplement_jaccard"] = len(a & b) / max(1, len(a | b))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
