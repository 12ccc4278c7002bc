#!/usr/bin/env python3
"""The in-process drop-in at a BASELINE size: oracle/_ref/ref_adapter (the reference's creator call site, src/main.cpp:239-296,
with GraphCreatorPrefSufHIP in place of GraphCreatorPrefSuf) on the node set of a config -- the time of the
creator region as the reference would see it (marshalling vector<Read*> -> alga_nodes, engine incl. PCIe, Graph::V fill,
retainOnlySmallestOffset), and the dump compared with the engine's own graph of the same nodes.
usage: tools/adapter_e2e.py [config=cfg2_1M_150bp]"""
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
sys.path.insert(0, os.path.join(ROOT, "tests"))
import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
import oracle_lib as O  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2_1M_150bp"
wl = workload.build(cfg)
words, lens = np.ascontiguousarray(wl["words"], dtype=np.uint32), wl["lens"].astype(np.int32)
out = {"config": cfg, "nodes": int(len(lens))}
exe = os.path.join(ROOT, "oracle", "_ref", "ref_adapter")
with tempfile.TemporaryDirectory() as wd:
    nodes = os.path.join(wd, "nodes.bin")
    with open(nodes, "wb") as f:
        f.write(np.array([len(lens), words.shape[1]], dtype=np.int32).tobytes())
        f.write(lens.tobytes())
        f.write(words.tobytes())
    g = os.path.join(wd, "hip.graph")
    for rep in range(2):
        t = time.perf_counter()
        r = subprocess.run([exe, "hip", nodes, g, str(wl["min_overlap"]), str(wl["rsoemo"]), "35"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        out["process_wall_s_run%d" % rep] = time.perf_counter() - t
        m = re.search(r"creator_region_ms ([\d.]+)", r.stdout)
        out["creator_region_ms_run%d" % rep] = float(m.group(1)) if m else None
        m = re.search(r"edges (\d+)", r.stdout)
        out["edges"] = int(m.group(1)) if m else None
    eng = alga_amd.Engine(0)
    mine = eng.prefsuf_host(words, lens, wl["min_overlap"], wl["rsoemo"])
    out["dump_equal_to_engine_graph"] = open(g, "rb").read() == O.graph_bytes(len(lens), mine)
print(json.dumps(out))
