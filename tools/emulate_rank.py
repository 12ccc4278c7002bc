#!/usr/bin/env python3
"""One rank's COMPUTE of the strong-scaling bench at world size N, on ONE GPU (no collectives): the north-star node set is
resident; the keys of the other ranks' nodes are put in place once, outside the timing (what the key all-gather delivers); timed
per step: alga_prefsuf_keys_device on the rank's own nodes + alga_prefsuf_build_range_device(keys_shared) on its sources.
DESIGN.md section 7 uses the result as the measured part of T_N; the two collectives (0.7 GB of keys, the rank's share of 1.1 GB
of edges) come on top.
The source range is built in `pieces` pieces as alga_amd.multigpu does (first piece keys_shared = 1, the others reuse the entry array).
  tools/emulate_rank.py [N=8] [n_reads=50000000] [genome=250000000] [steps=5] [pieces=4]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import multigpu, workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 250_000_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
pieces = int(sys.argv[5]) if len(sys.argv) > 5 else 4
wl = workload.device_build(n_reads, 150, G, 11)
torch.cuda.synchronize()                                           # the engine's stream does not order with torch's
dw, dl, lo, rs = wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]
n = int(dl.shape[0])
b = multigpu.shard_bounds(n, N)
eng, other = alga_amd.Engine(0), alga_amd.Engine(0)
ko = other.keys_device(dw, dl, lo, rs, 0, n)                       # every node's keys, once
assert ko is not None
torch.cuda.synchronize()                                           # alga_prefsuf_keys_device does not end in a host sync
all_keys = [device_view(p, (n,)).clone() for p in ko]
other.close()
torch.cuda.synchronize()
out = {"world": N, "reads": n_reads, "nodes": n, "pieces": pieces}
for r in sorted(set([0, N - 1])):
    rows = []
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = eng.keys_device(dw, dl, lo, rs, b[r], b[r + 1])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for mine, full in zip(k, all_keys):                          # stands in for the all-gather (not timed)
            v = device_view(mine, (n,))
            v[:b[r]] = full[:b[r]]
            v[b[r + 1]:] = full[b[r + 1]:]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        pb = multigpu.piece_bounds(b[r], b[r + 1], pieces)
        edges, seed, probe, emit = 0, 0.0, 0.0, 0.0
        for k in range(pieces):
            res = eng.build_range_device(dw, dl, lo, rs, pb[k], pb[k + 1], keys_shared=1 if k == 0 else 2)
            st = eng.last_stats()
            edges += res[1]; seed += st["ms_seed"]; probe += st["ms_probe"]; emit += st["ms_emit"]
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        st = dict(ms_seed=seed, ms_probe=probe, ms_emit=emit)
        res = (None, edges)
        if it:
            rows.append(((t1 - t0) * 1e3, st["ms_seed"], st["ms_probe"], st["ms_emit"], (t1 - t0 + t3 - t2) * 1e3))
    a = [sum(x) / len(x) for x in zip(*rows)]
    out["rank%d" % r] = {"sources": b[r + 1] - b[r], "edges": res[1], "ms_keys_own_nodes_wall": a[0], "ms_store_build": a[1], "ms_probe": a[2],
                         "ms_emit": a[3], "ms_compute_wall": a[4]}
    # the PILE form of the same rank (round 5): every rank computes the target keys of all nodes itself (the key pass of a build the piles keep
    # makes no run lists), builds the piles, and k_pile_probe walks the rank's id range -- one build call, no key all-gather
    rows = []
    for it in range(steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = eng.build_range_device(dw, dl, lo, rs, b[r], b[r + 1], keys_shared=0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = eng.last_stats()
        if it:
            rows.append((st["ms_seed"], st["ms_probe"], st["ms_emit"], st["ms_total"], (t1 - t0) * 1e3))
    a = [sum(x) / len(x) for x in zip(*rows)]
    out["rank%d_pile_form" % r] = {"sources": b[r + 1] - b[r], "edges": res[1], "ms_index": a[0], "ms_probe": a[1], "ms_emit": a[2], "ms_device": a[3],
                                   "ms_compute_wall": a[4], "pile_buckets": st["pile_buckets"], "pile_mixed": st["pile_mixed"],
                                   "same_edges_as_replicated_form": res[1] == out["rank%d" % r]["edges"]}
print(json.dumps(out))
