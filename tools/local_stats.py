#!/usr/bin/env python3
"""How many sources take which path of the source-side reduction (run on a GPU box): tools/local_stats.py [config]"""
import json
import sys

import numpy as np
import torch

import alga_amd
from alga_amd import workload

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_1M_150bp"
wl = workload.build(cfg, stride_words="aligned")
eng = alga_amd.Engine(0)
dw = torch.from_numpy(wl["words"].view(np.int32)).cuda()
dl = torch.from_numpy(wl["lens"]).cuda()
out = {}
for red in ("source_side", "per_target"):
    ptr, m = eng.prefsuf_device(dw, dl, wl["min_overlap"], wl["rsoemo"], collect_stats=True, reduction=red)
    st = eng.last_stats()
    out[red] = {k: st[k] for k in ("edges", "raw_overlaps", "records", "transitive_compares", "transitive_removed", "generic_sources", "reduction_used", "ms_total")}
out["nodes"] = int(len(wl["lens"]))
print(json.dumps(out))
