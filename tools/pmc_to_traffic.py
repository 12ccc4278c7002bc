#!/usr/bin/env python3
"""Extract the dominant kernel's HBM-side traffic from the rocprofv3 PMC passes of tools/profile_bench.sh and store it
where bench.py picks it up (profiles/probe_hbm_bytes.json -> roofline.traffic).

  tools/pmc_to_traffic.py gpurun_out/prof_<tag> <config name> [kernel substring]

FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch (MI355X_MICROARCH.md, section HBM).  The guide's x2 correction
of FETCH_SIZE applies to wide coalesced STREAMING reads (16 B/lane, 128-B requests tallied at 64 B).  k_probe_sources
does not stream: every request is one random 64-byte line read by four adjacent lanes, so each request is tallied at
its true 64 B; the uncorrected figure also matches the kernel's own byte model, and is stored
here as measured together with the doubled upper bound.
"""
import csv
import glob
import json
import os
import sys


def mean_counter(root, sub, counter, kernel):
    vals = []
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"] and "<true" not in r["Kernel_Name"].replace("(bool)1", "<true"):
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


def main():
    root, config = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "k_probe_sources"
    fetch = mean_counter(root, "pmc_fetch", "FETCH_SIZE", kernel)
    write = mean_counter(root, "pmc_write", "WRITE_SIZE", kernel)
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "probe_hbm_bytes.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    data[config] = {"kernel": kernel, "fetch_kib": fetch, "write_kib": write,
                    "hbm_bytes_per_launch": int((fetch + write) * 1024),
                    "hbm_bytes_per_launch_if_fetch_doubled": int((2 * fetch + write) * 1024),
                    "source": os.path.basename(root.rstrip("/"))}
    json.dump(data, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(data[config]))


if __name__ == "__main__":
    main()
