#!/usr/bin/env python3
"""A/B of the supplement's k-mer sort at configs[4] size (21.4 M records of 16 bytes): the engine's own sort on 30 bits against rocPRIM's on 32."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import alga_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 21_432_066
g = torch.Generator(device="cuda"); g.manual_seed(3)
k = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, device="cuda", generator=g)
v = torch.arange(n, dtype=torch.int64, device="cuda")
e = alga_amd.Engine(0)
out = {"n": n}
for name, bits, own in (("own_30", 30, True), ("own_40", 40, True), ("rocprim_32", 32, False), ("rocprim_40", 40, False)):
    _, _, ms = e.sort_u64_pairs_device(k, v, bits, own, repeat=10)
    out[name + "_ms"] = round(ms, 4)
    out[name + "_GBps"] = round(n * 32 * ((bits + (9 if own else 7)) // (10 if own else 8)) / ms / 1e6, 1)
print(json.dumps(out))
e.close()
