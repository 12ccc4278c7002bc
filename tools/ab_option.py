#!/usr/bin/env python3
"""One engine option switched off and on at a BASELINE size, same resident node set, graphs compared edge for edge:
  tools/ab_option.py option [n_reads=50000000] [genome=250000000] [steps=8] [err=0] [values=0,1]
prints one JSON line: per value the mean device time per build and the phases the engine's own HIP events saw."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402

opt = sys.argv[1]
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 250_000_000
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
err = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
values = [int(v) for v in sys.argv[6].split(",")] if len(sys.argv) > 6 else [0, 1]
wl = workload.device_build(n_reads, 150, G, 11, err=err)
torch.cuda.synchronize()
dw, dl, lo, rs = wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]
eng = alga_amd.Engine(0)
out = {"option": opt, "reads": n_reads, "nodes": int(dl.shape[0]), "src_sha256": alga_amd.engine.source_fingerprint()}
want = None
keys = ("ms_total", "ms_seed", "ms_keys", "ms_sort", "ms_gather", "ms_dir", "ms_pile", "ms_probe", "ms_probe_pairs", "ms_emit")
for rep in range(2):                                        # A B A B: drift of the box shows as a difference between the repeats
    for v in values:
        eng.set_option(opt, v)
        acc = {k: 0.0 for k in keys}
        for it in range(steps + 1):
            ptr, m = eng.prefsuf_device(dw, dl, lo, rs)
            st = eng.last_stats()
            if it:
                for k in keys:
                    acc[k] += st[k]
            last = {k: st[k] for k in ("deferred_sources", "pile_buckets", "pile_irregular", "pile_own_lists", "pile_mixed", "pile_deferred", "edges")}
        got = device_view(ptr, (m, 3), dw.device)
        if want is None:
            want = got.clone()
        out["%s=%d run %d" % (opt, v, rep)] = dict({k: round(acc[k] / steps, 3) for k in keys}, **last)
        if not os.environ.get("AB_NOCOMPARE") and (got.shape != want.shape or not bool(torch.equal(got, want))):
            a, b = got.cpu().numpy(), want.cpu().numpy()
            sa, sb = set(map(tuple, a.tolist())), set(map(tuple, b.tolist()))
            out["DIFFERENT"] = {"edges_got": len(sa), "edges_want": len(sb), "only_got": sorted(sa - sb)[:20], "only_want": sorted(sb - sa)[:20],
                                "n_only_got": len(sa - sb), "n_only_want": len(sb - sa)}
            print(json.dumps(out))
            raise SystemExit("option %s = %d changes the graph" % (opt, v))
out["edges"] = int(want.shape[0])
out["graphs_equal"] = True
print(json.dumps(out))
