#!/usr/bin/env python3
"""Full-size byte-exactness of the exact path on reads WITH errors (BASELINE configs[4], 10 M x 150 bp, 2 % substitutions): the
engine's graph of the seeded host-generated read set, written in the reference's dump format, as size + sha256 -- to be compared
with the `*_beforeSimplifier.graph` the reference binary writes for the same FASTA with `--threads=1 --error_rate=0.02
--serialize=1` (its only run-to-run deterministic order on noisy data; 22 CPU-minutes, so it runs
wherever there is a CPU: the generator is numpy with a fixed seed, the FASTA is identical on every machine).
usage: tools/cfg5_exact_dump_hash.py [n_reads=10000000] [genome=30000000]"""
import hashlib
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import alga_amd  # noqa: E402
from alga_amd import workload  # noqa: E402
import gen_reads  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 30_000_000
codes, _ = gen_reads.sample_reads(n, 150, G, 13, 0.02)
out = {"reads": n, "genome": G, "codes_sha256": hashlib.sha256(codes.tobytes()).hexdigest()[:16], "src_sha256": alga_amd.engine.source_fingerprint()}
# the reference's dump for this seeded FASTA (codes_sha256 903f5f5074d864ce), `--threads=1 --error_rate=0.02 --serialize=1`:
# profiles/r02_d_cfg5_10M_exact_path_vs_reference_threads1.json
REF_10M = {"codes_sha256": "903f5f5074d864ce", "dump_bytes": 196893756, "dump_sha256": "d540520b30e7986cb468790073e228b1b034798b048a2e6657409ed523b3f5ac"}
words, lens, _ = workload.make_nodes(codes, stride_words=12)
lo, rs = alga_amd.derive_params(144.0)
eng = alga_amd.Engine(0)
edges = eng.prefsuf_host(words, lens, lo, rs)
st = eng.last_stats()
with tempfile.TemporaryDirectory() as wd:
    p = os.path.join(wd, "gpu.graph")
    eng.write_graph(p, len(lens), edges)
    h = hashlib.sha256()
    with open(p, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    out.update(nodes=int(len(lens)), edges=int(len(edges)), dump_bytes=os.path.getsize(p), dump_sha256=h.hexdigest(),
               probe_used=st["probe_used"], device_ms=st["ms_total"])
if out["codes_sha256"] == REF_10M["codes_sha256"]:
    out["reference_recorded"] = REF_10M
    out["byte_identical_to_recorded_reference"] = out["dump_bytes"] == REF_10M["dump_bytes"] and out["dump_sha256"] == REF_10M["dump_sha256"]
print(json.dumps(out))
