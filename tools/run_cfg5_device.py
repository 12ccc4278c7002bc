#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU with the read set generated on the device (alga_amd.workload.device_build with substitution
errors): exact path + approximate supplement, device times with warm buffers.
usage: tools/run_cfg5_device.py [n_reads] [genome] [repeats] [--reference THREADS] [--oracle]
       (defaults: 10 M reads, 30 M genome = cfg5_10M_150bp_err2)
--reference: the same read set written as FASTA and run through the reference binary (oracle/_ref/ALGA --error_rate=0.02, THREADS
threads: its own timers and edge counts), then the reference's supplement code alone (oracle/_ref/ref_driver, its --threads=1
order) on the SAME nodes and the SAME exact graph: the symmetric difference to the engine's result is the effect of the
reference's order dependence.
--oracle: the CPU oracle's supplement (oracle/alga_oracle_pkb.cpp, single thread) on the same nodes and exact graph in its four
semantics: flags 3 (the engine's: ties by id + round snapshot) must equal the engine's result edge for edge; flags 0 (the
reference's sequential order), 1 and 2 show which of the two differences accounts for the distance to the reference."""
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
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402


def reference_leg(out, wl, pre, post, threads, lo):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    exe, drv = os.path.join(ROOT, "oracle", "_ref", "ALGA"), os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    codes = wl["sample_codes"]
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as wd:
        if os.path.exists(exe):
            workload.write_fasta_fast(os.path.join(wd, "s.fasta"), codes)
            t = time.perf_counter()
            r = subprocess.run([exe, "--file1=s.fasta", "--threads=%d" % threads, "--error_rate=0.02", "--output=o.fasta"], cwd=wd,
                               stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
            out["ref_threads"] = threads
            out["ref_total_wall_s"] = time.perf_counter() - t
            m = re.search(r"Before supplement, G has (\d+) edges", r.stderr)
            out["ref_edges_exact"] = int(m.group(1)) if m else None
            m = re.search(r"After supplement G has (\d+) edges", r.stderr)
            out["ref_edges_after_supplement"] = int(m.group(1)) if m else None
            for key in ("GraphCreator PrefSuf", "GraphCreator PKB Supplement"):
                m = re.search(re.escape(key) + r"[^\d\n]*([\d.]+)", r.stderr)
                out["ref_cpu_seconds_" + key.split()[-1]] = float(m.group(1)) if m else None
            os.unlink(os.path.join(wd, "s.fasta"))
        if os.path.exists(drv):
            words = np.ascontiguousarray(wl["words"].cpu().numpy().view(np.uint32))
            lens = wl["lens"].cpu().numpy().astype(np.int32)
            nodes = os.path.join(wd, "nodes.bin")
            with open(nodes, "wb") as f:
                f.write(np.array([len(lens), words.shape[1]], dtype=np.int32).tobytes())
                f.write(lens.tobytes())
                f.write(words.tobytes())
            gin, gout = os.path.join(wd, "in.graph"), os.path.join(wd, "out.graph")
            open(gin, "wb").write(O.graph_bytes(len(lens), pre))
            t = time.perf_counter()
            subprocess.run([drv, "supplement", nodes, gin, gout, "2", str(min(2 * lo // 3, 60))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            out["ref_driver_supplement_wall_s_1_thread"] = time.perf_counter() - t
            _, ref_post = O.parse_graph(open(gout, "rb").read())
            key = lambda e: (e[:, 0].astype(np.int64) << 36) | (e[:, 1].astype(np.int64) << 9) | e[:, 2].astype(np.int64)   # noqa: E731
            a, b = np.unique(key(np.asarray(ref_post))), np.unique(key(post))
            inter = np.intersect1d(a, b, assume_unique=True).size
            out["ref_driver_edges_after_supplement"] = int(a.size)
            out["supplement_symmetric_difference"] = int(a.size + b.size - 2 * inter)
            out["supplement_jaccard"] = inter / max(1, a.size + b.size - inter)


def oracle_leg(out, wl, pre, post, lo):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    words = np.ascontiguousarray(wl["words"].cpu().numpy().view(np.uint32))
    lens = wl["lens"].cpu().numpy().astype(np.int32)
    op = O.pkb_params(float(lens[lens > 0].mean()), error_rate_percent=2)
    kb = min(2 * lo // 3, 60)
    key = lambda e: (e[:, 0].astype(np.int64) << 36) | (e[:, 1].astype(np.int64) << 9) | e[:, 2].astype(np.int64)   # noqa: E731
    mine = np.unique(key(post))
    res = {}
    for flags in (3, 0, 1, 2):
        t = time.perf_counter()
        want, _ = O.supplement(words, lens, pre, op, kb, flags=flags)
        k = np.unique(key(want))
        inter = np.intersect1d(k, mine, assume_unique=True).size
        res["flags_%d" % flags] = dict(edges=int(k.size), symmetric_difference_to_engine=int(k.size + mine.size - 2 * inter), seconds=time.perf_counter() - t)
        if flags == 3:
            res["engine_equals_oracle_engine_semantics"] = bool(want.shape == post.shape and (want == post).all())
    out["oracle"] = res


def main():
    argv = list(sys.argv[1:])
    do_oracle = "--oracle" in argv
    if do_oracle:
        argv.remove("--oracle")
    ref_threads = 0
    if "--reference" in argv:
        k = argv.index("--reference")
        ref_threads = int(argv[k + 1])
        del argv[k:k + 2]
    n = int(argv[0]) if len(argv) > 0 else 10_000_000
    G = int(argv[1]) if len(argv) > 1 else 3 * n
    rep = int(argv[2]) if len(argv) > 2 else 3
    wl = workload.device_build(n, 150, G, 13, err=0.02, sample_reads=n if ref_threads else 0)
    words, lens, lo, rs = wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]
    torch.cuda.synchronize()
    eng = alga_amd.Engine(0)
    p = eng.pkb_params(float(lens.float().mean().item()), 0.02, min(2 * lo // 3, 60))
    exact_ms, supp_ms = [], []
    for _ in range(rep + 1):
        ptr, m = eng.prefsuf_device(words, lens, lo, rs)
        st = eng.last_stats()
        pre = alga_amd.engine.device_view(ptr, (m, 3), words.device).clone()        # the next call reuses the engine's edge buffer
        ptr2, m2 = eng.pkb_supplement_device(words, lens, pre.data_ptr(), m, p)
        ps = eng.pkb_last_stats()
        post = alga_amd.engine.device_view(ptr2, (m2, 3), words.device).to(torch.int64)
        h = (post[:, 0] * 1000003 + post[:, 1]) * 1009 + post[:, 2]
        checksum = [int(h.sum().item()), int((h * h).sum().item())]              # wrapping int64: order-independent fingerprints of the edge set
        ordered = bool(((post[1:, 0] > post[:-1, 0]) | ((post[1:, 0] == post[:-1, 0]) & (post[1:, 1] > post[:-1, 1]))).all().item())
        post_np = post.cpu().numpy().astype(np.int32) if (ref_threads or do_oracle) else None
        del post, h
        exact_ms.append(st["ms_total"]); supp_ms.append(ps["ms_total"])
    out = dict(reads=n, genome=G, nodes=int(lens.shape[0]), edges_exact=int(m), edges_after_supplement=int(m2), probe_used=st["probe_used"], edge_set_checksum=checksum, sorted_unique=ordered,
               deferred_sources=st.get("deferred_sources"), exact_phases={k: round(st[k], 3) for k in ("ms_seed", "ms_probe", "ms_group", "ms_reduce", "ms_emit")},
               exact_device_ms_first=exact_ms[0], supplement_device_ms_first=supp_ms[0], exact_device_ms_warm=exact_ms[1:],
               supplement_device_ms_warm=supp_ms[1:], supplement=ps)
    if ref_threads:
        reference_leg(out, wl, pre.cpu().numpy().astype(np.int32), post_np, ref_threads, lo)
    if do_oracle:
        oracle_leg(out, wl, pre.cpu().numpy().astype(np.int32), post_np, lo)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
