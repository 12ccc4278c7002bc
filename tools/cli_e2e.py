#!/usr/bin/env python3
"""End-to-end time of the drop-in CLI at BASELINE configs[1]: FASTA on disk -> alga_hip (ingest on the host cores, overlap graph
on the GPU, .graph dump) next to the reference binary's own time to the same point.
usage: tools/cli_e2e.py [config] [threads] [--paired] [--compare-dump] [--no-ref]
--no-ref: alga_hip alone (the reference needs ten minutes for the 50 M reads of cfg4_50M_150bp; profiles/r04_a_cfg4_50M_dump_vs_reference.log has it);
configs above 20 M reads are written to the FASTA file chunk by chunk (alga_amd.workload.write_fasta_chunked);
--paired: the reads go into two files (--file1 / --file2, record i of file f = read 2i + f: BASELINE configs[2]'s input form);
--compare-dump: the reference runs with --serialize=1 and its *_beforeSimplifier.graph must equal alga_hip's byte for byte."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_reads  # noqa: E402
from alga_amd import workload  # noqa: E402

import glob  # noqa: E402
import shutil  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
paired, compare, no_ref = "--paired" in sys.argv, "--compare-dump" in sys.argv, "--no-ref" in sys.argv
cfg = argv[0] if len(argv) > 0 else "cfg2_1M_150bp"
threads = argv[1] if len(argv) > 1 else "16"
n, L, G, seed, err = workload.CONFIGS[cfg]
big = n > 20_000_000 and not paired and err == 0
codes = None if big else gen_reads.sample_reads(n, L, G, seed, err)[0]
out = {"config": cfg, "threads": int(threads), "paired": paired}
with tempfile.TemporaryDirectory(dir=os.environ.get("ALGA_TMP", None)) as wd:
    if big:
        t = time.perf_counter()
        workload.write_fasta_chunked(os.path.join(wd, "s.fasta"), n, L, G, seed)
        out["fasta_bytes"], out["fasta_write_s"] = os.path.getsize(os.path.join(wd, "s.fasta")), time.perf_counter() - t
        files = ["--file1=s.fasta"]
    elif paired:
        k = len(codes) // 2
        workload.write_fasta_fast(os.path.join(wd, "s.fasta"), codes[0:2 * k:2])
        workload.write_fasta_fast(os.path.join(wd, "s2.fasta"), codes[1:2 * k:2])
        files = ["--file1=s.fasta", "--file2=s2.fasta"]
    else:
        workload.write_fasta_fast(os.path.join(wd, "s.fasta"), codes)
        files = ["--file1=s.fasta"]
    exe = os.path.join(ROOT, "alga_amd", "bin", "alga_hip")
    for rep in range(3):                     # second run: file in the page cache, HIP runtime warm on disk
        t = time.perf_counter()
        r = subprocess.run([exe] + files + ["--threads=" + threads, "--output=o.fasta"], cwd=wd, stdout=subprocess.DEVNULL,
                           stderr=subprocess.PIPE, text=True)
        out["alga_hip_wall_s_run%d" % rep] = time.perf_counter() - t
        m = re.search(r"HIP start-up ([\d.]+) ms", r.stderr)
        if m:
            out["alga_hip_hip_startup_ms"] = float(m.group(1))
        m = re.search(r"parse ([\d.]+) ms, duplicate/prefix removal ([\d.]+) ms wall \(device ([\d.]+) ms\), overlap graph ([\d.]+) ms wall \(device ([\d.]+) ms", r.stderr)
        if m:
            (out["alga_hip_parse_ms"], out["alga_hip_dedupe_wall_ms"], out["alga_hip_dedupe_device_ms"], out["alga_hip_graph_wall_ms"],
             out["alga_hip_graph_device_ms"]) = map(float, m.groups())
        m = re.search(r"Before first simplifier graph has (\d+) edges", r.stderr)
        out["alga_hip_edges"] = int(m.group(1)) if m else None
        out["alga_hip_log_run%d" % rep] = [ln.strip() for ln in r.stderr.splitlines() if "ms" in ln][-12:]
        if r.returncode != 0:
            out["alga_hip_error"] = r.stderr[-2000:]
            break
    mine = glob.glob(os.path.join(wd, "*_beforeSimplifier.graph"))
    if compare and mine:
        shutil.move(mine[0], os.path.join(wd, "gpu.graph"))
    ref = os.path.join(ROOT, "oracle", "_ref", "ALGA")
    if os.path.exists(ref) and not no_ref:
        t = time.perf_counter()
        p = subprocess.Popen([ref] + files + ["--threads=" + threads, "--output=r.fasta"] + (["--serialize=1"] if compare else []), cwd=wd,
                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace")
        for line in p.stderr:
            if "Creating GraphCreator" in line:
                out["ref_to_graph_creator_s"] = time.perf_counter() - t
            if "Before first simplifier" in line:
                out["ref_to_graph_done_s"] = time.perf_counter() - t
                m = re.search(r"(\d+) edges", line)
                out["ref_edges"] = int(m.group(1)) if m else None
                p.kill()
                break
        p.wait()
        if compare:
            theirs = glob.glob(os.path.join(wd, "*_beforeSimplifier.graph"))
            out["dump_bytes"] = os.path.getsize(theirs[0]) if theirs else None
            out["dump_equal_to_reference"] = bool(theirs) and open(theirs[0], "rb").read() == open(os.path.join(wd, "gpu.graph"), "rb").read()
print(json.dumps(out))
