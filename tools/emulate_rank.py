#!/usr/bin/env python3
"""One rank's compute of the weak-scaling bench at world size N, on ONE GPU: the node set of N x configs[1] is resident,
the rank builds the edges of its 1/N of the sources (alga_prefsuf_build_range_device).  No collectives.
  tools/emulate_rank.py [N=8] [config=cfg2_1M_150bp] [steps=5]"""
import json
import sys
import time

import numpy as np
import torch

import alga_amd
from alga_amd import multigpu, workload

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2_1M_150bp"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t0 = time.time()
wl = workload.build(cfg, scale=N, stride_words="aligned")
print("workload built in %.1f s: %d nodes" % (time.time() - t0, len(wl["lens"])), flush=True)
eng = alga_amd.Engine(0)
dw = torch.from_numpy(wl["words"].view(np.int32)).cuda()
dl = torch.from_numpy(wl["lens"]).cuda()
n = len(wl["lens"])
b = multigpu.shard_bounds(n, N)
out = {"world": N, "nodes": n}
for r in sorted(set([0, N - 1])):
    ms = []
    for it in range(steps + 1):
        res = eng.build_range_device(dw, dl, wl["min_overlap"], wl["rsoemo"], b[r], b[r + 1])
        st = eng.last_stats()
        if it:
            ms.append((st["ms_seed"], st["ms_probe"], st["ms_emit"], st["ms_total"]))
    a = np.mean(ms, axis=0)
    out["rank%d" % r] = {"sources": b[r + 1] - b[r], "edges": res[1], "ms_seed": a[0], "ms_probe": a[1], "ms_emit": a[2], "ms_total": a[3]}
print(json.dumps(out))
