#!/usr/bin/env python3
"""The approximate supplement with option pkb_legacy at several values, on the same resident exact graph (BASELINE configs[4]'s shape by default),
results compared edge for edge:
  tools/ab_pkb.py [values=0,256] [n_reads=10000000] [genome=30000000] [steps=8] [err=0.02] [with_build=0]
prints one JSON line: per value (A B A B) the mean device time of the supplement.  with_build 1: the exact build runs in front of every supplement,
as in bench.py's step (its buffers and the caches are then what the supplement finds there)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402

values = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 256]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 30_000_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
err = float(sys.argv[5]) if len(sys.argv) > 5 else 0.02
with_build = len(sys.argv) > 6 and sys.argv[6] == "1"
wl = workload.device_build(n_reads, 150, G, 11, err=err)
torch.cuda.synchronize()
dw, dl, lo, rs = wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]
eng = alga_amd.Engine(0)
pkb = alga_amd.Engine.pkb_params(float(dl[dl > 0].float().mean().item()), err, min(2 * lo // 3, 60))
ptr, m = eng.prefsuf_device(dw, dl, lo, rs)
exact = device_view(ptr, (m, 3), dw.device).clone()
out = {"reads": n_reads, "nodes": int(dl.shape[0]), "edges_exact": int(m), "src_sha256": alga_amd.engine.source_fingerprint()}
want = None
for rep in range(2):
    for v in values:
        eng.set_option("pkb_legacy", v)
        acc = 0.0
        for it in range(steps + 1):
            if with_build:
                eng.prefsuf_device(dw, dl, lo, rs)
            p2, m2 = eng.pkb_supplement_device(dw, dl, exact.data_ptr(), int(exact.shape[0]), pkb)
            if it:
                acc += eng.pkb_last_stats()["ms_total"]
        got = device_view(p2, (m2, 3), dw.device)
        if want is None:
            want = got.clone()
        out["pkb_legacy=%d run %d" % (v, rep)] = round(acc / steps, 3)
        if got.shape != want.shape or not bool(torch.equal(got, want)):
            print(json.dumps(out))
            raise SystemExit("pkb_legacy = %d changes the graph" % v)
out["edges"] = int(want.shape[0])
out["graphs_equal"] = True
print(json.dumps(out))
