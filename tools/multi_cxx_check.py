#!/usr/bin/env python3
"""The C++ N-GPU driver (alga_multi_*, alga_amd/csrc/engine_multi.hip) at a BASELINE size: N ranks, on as many GPUs as the box has
(ranks beyond them share GPU 0: copy transport), against the one-engine graph of the same node set, edge for edge on the device.
usage: tools/multi_cxx_check.py [ranks=2] [config=cfg4_50M_150bp] [form=replicated|bucket_sharded] [transport=auto|copy|rccl]"""
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

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
config = sys.argv[2] if len(sys.argv) > 2 else "cfg4_50M_150bp"
form = sys.argv[3] if len(sys.argv) > 3 else "replicated"
transport = sys.argv[4] if len(sys.argv) > 4 else "auto"
n_reads, L, G, seed, err = workload.CONFIGS[config]
ngpu = torch.cuda.device_count()
devices = [r if r < ngpu else 0 for r in range(ranks)]
if len(set(devices)) < ranks and transport == "auto":
    transport = "copy"
per_dev = {}
for d in sorted(set(devices)):
    wl = workload.device_build(n_reads, L, G, seed, device="cuda:%d" % d, err=err)
    per_dev[d] = (wl["words"], wl["lens"])
    lo, rs = wl["min_overlap"], wl["rsoemo"]
torch.cuda.synchronize()
dw, dl = per_dev[devices[0]]
e0 = alga_amd.Engine(devices[0])
ptr, m = e0.prefsuf_device(dw, dl, lo, rs)
want = device_view(ptr, (m, 3), dw.device).clone()
e0.close()
mu = alga_amd.MultiEngine(devices, transport=transport)
mu.set_option("form", form)
out = {"config": config, "ranks": ranks, "devices": devices, "form_asked": form, "src_sha256": alga_amd.engine.source_fingerprint(), "nodes": int(dl.shape[0]), "edges_one_engine": int(m), "runs": []}
for it in range(3):
    t0 = time.perf_counter()
    ptr, k = mu.prefsuf_device([per_dev[d] for d in devices], lo, rs)
    dt = time.perf_counter() - t0
    got = device_view(ptr, (k, 3), dw.device)
    st = mu.last_stats()
    out["runs"].append({"wall_ms": dt * 1e3, "equal_to_one_engine": bool(k == m and torch.equal(got, want)),
                        **{x: st[x] for x in st if x != "ranks"},
                        "rank_edges": [r["edges"] for r in st["ranks"]]})
mu.close()
print(json.dumps(out))
