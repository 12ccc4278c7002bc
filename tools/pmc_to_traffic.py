#!/usr/bin/env python3
"""Memory-side traffic of every engine kernel from the rocprofv3 PMC passes `r` and `x` of tools/profile_cmd.sh, stored where
bench.py picks it up (profiles/hbm_traffic.json -> roofline.traffic, roofline_kernels[].traffic), tagged with the fingerprint
of the kernel sources it was measured on (alga_amd.engine.source_fingerprint): bench.py quotes it only for that very code.

  tools/pmc_to_traffic.py gpurun_out/prof_<tag> <config name>

gfx950 counts the L2's memory-side requests BY SIZE (`rocprofv3 -L`: TCC_EA0_RDREQ_32B / _64B / _128B, TCC_EA0_WRREQ / _64B):
    bytes read    = 32 * RDREQ_32B + 64 * RDREQ_64B + 128 * RDREQ_128B
    bytes written = 64 * WRREQ_64B + 32 * (WRREQ - WRREQ_64B)
Measured on this engine (profiles/r03_*): EVERY read request is a 128-byte one (RDREQ_128B == RDREQ, none of 32 or 64 bytes), for
streaming and for isolated accesses alike.  Calibration on kernels of known byte count, same run: k_node_runs streams every node
row (64 B) and length once: 6.16 GB expected, 128 * RDREQ = 6.16 GB; k_tgt_gather reads one isolated 64-byte row per node and
fetches 128 bytes for it (12.6 GB for 5.8 GB of rows + 0.7 GB of keys).  FETCH_SIZE, which ROCm 7.2 derives with the gfx94x
formula (64 bytes per request), therefore reports HALF of the bytes read for every access pattern: the guide's "x2" holds
everywhere, and the round-2 bracket of this repository (78 .. 153 GB for the probe phase) resolves to its upper end.
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name):
    name = name.split("(")[0]
    for p in ("void alga::", "alga::"):
        if name.startswith(p):
            name = name[len(p):]
    return name


def means(root, sub):
    acc = {}
    # (one pass = one rocprofv3 process = one file; gpurun merges every call's files into the same local directory, so only the NEWEST
    # file of a pass directory belongs to the build at hand -- averaging over older ones mixed kernels of different builds)
    files = sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        for r in csv.DictReader(open(f)):
            acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    from alga_amd.engine import source_fingerprint
    root, config = sys.argv[1], sys.argv[2]
    rd, wr = means(root, "pmc_rdreq"), means(root, "pmc_wrreq")
    kernels = {}
    for k in sorted(set(rd) | set(wr)):
        if not (k.startswith("k_") or "rocprim" in k) or "<true" in k:          # engine kernels and the library sort; not the statistics builds
            continue
        r, w = rd.get(k, {}), wr.get(k, {})
        n128, n64, n32 = r.get("TCC_EA0_RDREQ_128B_sum", 0.0), r.get("TCC_EA0_RDREQ_64B_sum", 0.0), r.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        other = r.get("TCC_EA0_RDREQ_sum", 0.0) - n128 - n64 - n32                # requests of no listed size (none seen): priced at 64 bytes
        w64 = w.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        kernels[k[:70]] = {"read_bytes": int(128 * n128 + 64 * n64 + 32 * n32 + 64 * max(0.0, other)),
                           "write_bytes": int(64 * w64 + 32 * max(0.0, w.get("TCC_EA0_WRREQ_sum", 0.0) - w64)),
                           "read_requests": {"128B": int(n128), "64B": int(n64), "32B": int(n32)}}
    out_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[config] = {"src_sha256": source_fingerprint(), "source": os.path.basename(root.rstrip("/")), "per_dispatch": kernels}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["read_bytes"] + v["write_bytes"] for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
