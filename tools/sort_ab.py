#!/usr/bin/env python3
"""A/B of the (key, id) sort of the index build at the north-star size: the engine's radix sort (alga_amd/csrc/radix_sort.hip) against
rocPRIM's onesweep, same keys (uniform hashes, as the minimizer keys are), 29 significant bits (begin_bit 3) and 32.
  python tools/sort_ab.py [n_pairs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import alga_amd  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 90_621_096
g = torch.Generator(device="cuda")
g.manual_seed(1)
k = torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), dtype=torch.int32, device="cuda", generator=g)
v = torch.arange(n, dtype=torch.int32, device="cuda")
eng = alga_amd.Engine(0)
out = {"n": n, "device": eng.device_name(), "src_sha256": alga_amd.engine.source_fingerprint()}
variants = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
for bb in (3, 0):
    res = {}
    for own in [1 + vv for vv in variants] + [0] + [1 + vv for vv in variants] + [0]:
        if own:
            eng.set_option("rsort_variant", own - 1)
        kp, vp, ms = eng.sort_u32_pairs_device(k, v, bb, 1 if own else 0, repeat=5)
        torch.cuda.synchronize()
        res.setdefault(("own_v%d" % (own - 1)) if own else "rocprim", []).append(round(ms, 4))
        got = device_view(vp, (n,), k.device).clone()
        if own:
            mine = got
        else:
            res["equal"] = bool(torch.equal(mine, got))
        del got
    res["own"] = res["own_v%d" % variants[0]]
    # bytes a pass has to move: 4 (histogram) + 16 per pair
    passes = (32 - bb + 9) // 10
    res["own_algorithmic_GBps"] = round(passes * 20 * n / (min(res["own"]) * 1e-3) / 1e9, 1)
    out["begin_bit_%d" % bb] = res
print(json.dumps(out))
