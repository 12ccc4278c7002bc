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
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for p in ("void alga::", "alga::", "void ", "(anonymous namespace)::"):       # (radix_sort.hip keeps its kernels in an anonymous namespace)
        if name.startswith(p):
            name = name[len(p):]
    return name


def collect(root, sub):
    """{kernel: {counter: [values in dispatch order]}} of the newest file of a pass directory, from the engine's FIRST dispatch on:
    everything before it is the workload generator (torch kernels, and torch's own rocPRIM sorts, which carry the same names as the
    library sorts the supplement still uses)."""
    # (one pass = one rocprofv3 process = one file; gpurun merges every call's files into the same local directory, so only the NEWEST
    # file of a pass directory belongs to the build at hand -- averaging over older ones mixed kernels of different builds)
    files = sorted(glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    rows = []
    for f in files[-1:]:
        rows = [(int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"], float(r["Counter_Value"])) for r in csv.DictReader(open(f))]
    first = min([d for d, k, _, _ in rows if k.startswith("k_")], default=0)
    acc = {}
    for d, k, c, v in sorted(rows):
        if d >= first:
            acc.setdefault(k, {}).setdefault(c, []).append(v)
    return acc


def main():
    from alga_amd.engine import source_fingerprint
    argv = [a for a in sys.argv[1:] if not a.startswith("--builds")]
    builds = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--builds=")]
    root, config = argv[0], argv[1]
    rd, wr = collect(root, "pmc_rdreq"), collect(root, "pmc_wrreq")
    kernels = {}
    total = 0.0
    for k in sorted(set(rd) | set(wr)):
        if not (k.startswith("k_") or "rocprim" in k):                           # engine kernels and the library sorts
            continue
        r, w = rd.get(k, {}), wr.get(k, {})
        tot = lambda d, c: float(sum(d.get(c, [])))
        calls = max([len(v) for v in list(r.values()) + list(w.values())] or [1])
        n128, n64, n32 = tot(r, "TCC_EA0_RDREQ_128B_sum"), tot(r, "TCC_EA0_RDREQ_64B_sum"), tot(r, "TCC_EA0_RDREQ_32B_sum")
        other = tot(r, "TCC_EA0_RDREQ_sum") - n128 - n64 - n32                    # requests of no listed size (none seen): priced at 64 bytes
        w64 = tot(w, "TCC_EA0_WRREQ_64B_sum")
        rb = 128 * n128 + 64 * n64 + 32 * n32 + 64 * max(0.0, other)
        wb = 64 * w64 + 32 * max(0.0, tot(w, "TCC_EA0_WRREQ_sum") - w64)
        total += rb + wb
        if "<true" in k and k.startswith("k_probe"):                             # the statistics builds of the probes (bench.py's counted pass): not a timed kernel
            continue
        kernels[k[:70]] = {"read_bytes": int(rb / calls), "write_bytes": int(wb / calls), "dispatches": calls, "bytes_per_build": int((rb + wb) / builds[0]) if builds else None,
                           "read_requests": {"128B": int(n128 / calls), "64B": int(n64 / calls), "32B": int(n32 / calls)}}
    out_path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    ent = {"src_sha256": source_fingerprint(), "source": os.path.basename(root.rstrip("/")), "per_dispatch": kernels}
    if builds:
        # the profiled command was `bench.py --traffic-pass`: W + K builds of the timed form and nothing else of the engine -- the sum over every
        # dispatch / builds = the HBM bytes of ONE step (bench.py: roofline_step.traffic_bytes)
        ent["builds"] = builds[0]
        ent["per_step_bytes"] = int(total / builds[0])
    data[config] = ent
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps({"per_step_bytes": ent.get("per_step_bytes"), **{k: v["read_bytes"] + v["write_bytes"] for k, v in kernels.items()}}))


if __name__ == "__main__":
    main()
