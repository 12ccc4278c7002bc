#!/usr/bin/env python3
"""Extract the probe phase's memory-side traffic from the rocprofv3 PMC passes of tools/profile_cmd.sh (passes f and w of bench.py)
and store it where bench.py picks it up (profiles/probe_hbm_bytes.json -> roofline.traffic), tagged with the fingerprint of the
kernel sources it was measured on (alga_amd.engine.source_fingerprint): bench.py reports it only for that very code.

  tools/pmc_to_traffic.py gpurun_out/prof_<tag> <config name>

FETCH_SIZE / WRITE_SIZE are KiB per dispatch (MI355X_MICROARCH.md, section HBM).  Calibration on this build, same run, kernels of
known byte count: k_node_runs streams every node row once (coalesced dword loads, 64 B per row + the lengths) and FETCH_SIZE shows
HALF of those bytes -- the guide's gfx950 correction (x2) applies to coalesced streams; k_tgt_gather reads one random 64-byte line per
node (+ one random 4-byte meta word = one more line) and FETCH_SIZE shows them in FULL -- no correction for isolated 64-byte lines.
The probe kernels read a mix (a 64-byte row, a 64-byte run list, random index lines: isolated lines; the entries of a cluster:
runs of adjacent lines), so both figures are stored: `hbm_bytes_per_launch` = FETCH + WRITE as counted, and the upper bound with
FETCH doubled.
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PROBE_KERNELS = ("k_probe_pairs", "k_probe_clustered", "k_probe_sources")


def per_launch(root, sub, counter):
    """sum over the probe kernels of (mean per dispatch), non-statistics instantiations only"""
    acc = {}
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            k = next((p for p in PROBE_KERNELS if p in name), None)
            if k is None or r["Counter_Name"] != counter or "<true" in name or "(bool)1" in name:
                continue
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    from alga_amd.engine import source_fingerprint
    root, config = sys.argv[1], sys.argv[2]
    fetch, write = per_launch(root, "pmc_fetch", "FETCH_SIZE"), per_launch(root, "pmc_write", "WRITE_SIZE")
    lib = os.path.join(ROOT, "alga_amd", "lib", "libalga_amd.so")
    out_path = os.path.join(ROOT, "profiles", "probe_hbm_bytes.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    f_kib, w_kib = sum(fetch.values()), sum(write.values())
    data[config] = {"kernels": sorted(set(fetch) | set(write)), "fetch_kib": fetch, "write_kib": write,
                    "hbm_bytes_per_launch": int((f_kib + w_kib) * 1024),
                    "hbm_bytes_per_launch_if_fetch_doubled": int((2 * f_kib + w_kib) * 1024),
                    "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16], "src_sha256": source_fingerprint(),
                    "source": os.path.basename(root.rstrip("/"))}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[config]))


if __name__ == "__main__":
    main()
