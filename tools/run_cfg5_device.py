#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU with the read set generated on the device (alga_amd.workload.device_build with substitution
errors): exact path + approximate supplement, device times with warm buffers.
usage: tools/run_cfg5_device.py [n_reads] [genome] [repeats]     (defaults: 10 M reads, 30 M genome = cfg5_10M_150bp_err2)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 3 * n
    rep = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    wl = workload.device_build(n, 150, G, 13, err=0.02)
    words, lens, lo, rs = wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]
    torch.cuda.synchronize()
    eng = alga_amd.Engine(0)
    p = eng.pkb_params(float(lens.float().mean().item()), 0.02, min(2 * lo // 3, 60))
    exact_ms, supp_ms = [], []
    for _ in range(rep + 1):
        ptr, m = eng.prefsuf_device(words, lens, lo, rs)
        st = eng.last_stats()
        pre = alga_amd.engine.device_view(ptr, (m, 3), words.device).clone()        # the next call reuses the engine's edge buffer
        ptr2, m2 = eng.pkb_supplement_device(words, lens, pre.data_ptr(), m, p)
        ps = eng.pkb_last_stats()
        post = alga_amd.engine.device_view(ptr2, (m2, 3), words.device).to(torch.int64)
        h = (post[:, 0] * 1000003 + post[:, 1]) * 1009 + post[:, 2]
        checksum = [int(h.sum().item()), int((h * h).sum().item())]              # wrapping int64: order-independent fingerprints of the edge set
        ordered = bool(((post[1:, 0] > post[:-1, 0]) | ((post[1:, 0] == post[:-1, 0]) & (post[1:, 1] > post[:-1, 1]))).all().item())
        del post, h
        exact_ms.append(st["ms_total"]); supp_ms.append(ps["ms_total"])
    out = dict(reads=n, genome=G, nodes=int(lens.shape[0]), edges_exact=int(m), edges_after_supplement=int(m2), probe_used=st["probe_used"], edge_set_checksum=checksum, sorted_unique=ordered,
               deferred_sources=st.get("deferred_sources"), exact_phases={k: round(st[k], 3) for k in ("ms_seed", "ms_probe", "ms_group", "ms_reduce", "ms_emit")},
               exact_device_ms_first=exact_ms[0], supplement_device_ms_first=supp_ms[0], exact_device_ms_warm=exact_ms[1:],
               supplement_device_ms_warm=supp_ms[1:], supplement=ps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
