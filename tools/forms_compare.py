#!/usr/bin/env python3
"""Both forms of the transitive reduction -- and, for the source-side form, both probes -- on one synthetic set:
tools/forms_compare.py n_reads genome err [steps] [read_len=150]"""
import json
import sys

import numpy as np
import torch

import alga_amd
import gen_reads
from alga_amd import workload

n, G, err = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
RL = int(sys.argv[5]) if len(sys.argv) > 5 else 150
codes, _ = gen_reads.sample_reads(n, RL, G, 13, err)
words, lens, ids = workload.make_nodes(codes, stride_words="aligned")
lo, rs = alga_amd.derive_params(float(RL - 6))
eng = alga_amd.Engine(0)
dw = torch.from_numpy(words.view(np.int32)).cuda()
dl = torch.from_numpy(lens).cuda()
out = {"nodes": int(len(lens)), "read_len": RL, "min_overlap": lo, "rsoemo": rs}
for red, probe in (("source_side", "cluster"), ("source_side", "table"), ("per_target", "auto")):
    ms = []
    eng.set_option("probe", probe)
    name = red + ("_" + probe if red == "source_side" else "")
    for it in range(steps + 1):
        ptr, m = eng.prefsuf_device(dw, dl, lo, rs, collect_stats=(it == 0), reduction=red)
        st = eng.last_stats()
        if it == 0:
            out[name + "_counters"] = {k: st[k] for k in ("edges", "raw_overlaps", "records", "generic_sources", "probe_used")}
        else:
            ms.append(st["ms_total"])
    out[name + "_ms"] = float(np.mean(ms))
    out[name + "_phases_ms"] = {k: round(st[k], 3) for k in ("ms_seed", "ms_probe", "ms_emit")}
print(json.dumps(out))
