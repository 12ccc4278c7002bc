#!/usr/bin/env python3
"""Both probes of the source-side form on one device-generated read set (error-free 150 bp, SURVEY.md section 8(d) shape):
the bucketised seed table and the clustered minimizer join must give the same edge list, edge for edge; prints their phase
times.  usage: tools/probe_compare.py [n_reads=16000000] [genome=80000000] [steps=3] [probes=table,cluster] [bias=0]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 80_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    probes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["table", "cluster"]
    bias = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    t0 = time.time()
    wl = workload.device_build(n, 150, G, 11)
    torch.cuda.synchronize()
    print("node set: %d nodes (%d unique reads) in %.1f s; min_overlap %d rsoemo %d" %
          (wl["lens"].shape[0], wl["unique_reads"], time.time() - t0, wl["min_overlap"], wl["rsoemo"]), flush=True)
    eng = alga_amd.Engine(0)
    eng.set_option("cluster_bucket_bias", bias)
    out = dict(reads=n, genome=G, nodes=int(wl["lens"].shape[0]))
    keep = {}
    for probe in probes:
        eng.set_option("probe", probe)
        ms = []
        for it in range(steps):
            ptr, E = eng.prefsuf_device(wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"], collect_stats=(it == 0), reduction="source_side")
            st = eng.last_stats()
            if it == 0:
                out[probe + "_counters"] = {k: st[k] for k in ("raw_overlaps", "records", "edges", "generic_sources", "windows_probed", "slots_scanned",
                                                                "big_sources", "probe_used", "deferred_sources")}
            ms.append({k: round(st[k], 3) for k in ("ms_total", "ms_seed", "ms_probe", "ms_emit")})
            print(probe, it, E, ms[-1], flush=True)
        out[probe] = ms
        keep[probe] = device_view(ptr, (E, 3), wl["words"].device).clone()
    if len(probes) == 2:
        a, b = keep[probes[0]], keep[probes[1]]
        out["probes_agree"] = bool(a.shape == b.shape and torch.equal(a, b))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
