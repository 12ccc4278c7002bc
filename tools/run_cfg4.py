#!/usr/bin/env python3
"""Config-4-shaped run on ONE GPU (BASELINE.json configs[3]: 50 M x 150 bp over a 250 Mb genome, error-free): the node set
is generated in chunks straight into the engine's HBM layout, duplicates removed by start position (a forward read and a
reverse read of the same interval are the same sequence up to reverse complement: the reference's duplicate removal keeps
one of them; on an iid genome nothing else is a duplicate) -- and the graph is built with BOTH forms of the transitive reduction,
which must agree edge for edge (the size-independent parity check at a size no CPU oracle finishes).
usage: tools/run_cfg4.py [n_reads=50000000] [genome=250000000] [steps=2] [forms=source_side,per_target] [ref_threads=0] [ref_sha256=]
ref_sha256 (with ref_threads = 0): the sha256 of the reference's dump for this seeded read set as recorded by an earlier full run
(error-free data: the reference's graph does not depend on its thread count); only the engine's dump is made and compared with it.
The output carries alga_amd.engine.source_fingerprint(): the kernel sources the result was computed with.
With ref_threads > 0 the same reads are written as FASTA and the real reference (oracle/_ref/ALGA) builds its graph beside it:
wall time of its creator region, its edge count, and its `--serialize=1` dump compared byte for byte (size + sha256) with the engine's
graph written by alga_write_graph."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000_000
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    forms = sys.argv[4].split(",") if len(sys.argv) > 4 else ["source_side", "per_target"]
    ref_threads = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    ref_sha = sys.argv[6] if len(sys.argv) > 6 else ""
    L, trim = 150, 3
    m = L - 2 * trim
    t0 = time.time()
    rng = np.random.default_rng(11)
    genome = rng.integers(0, 4, G, dtype=np.uint8)
    starts = rng.integers(0, G - L + 1, n)
    flip = rng.random(n) < 0.5
    starts, first = np.unique(starts, return_index=True)   # one read per start position, either strand
    flip = flip[first]
    perm = rng.permutation(len(starts))                 # node ids carry no positional information, as in a real read file
    starts, flip = starts[perm], flip[perm]
    R = len(starts)
    N = 2 * R
    stride = 16
    words = np.zeros((N, stride), dtype=np.uint32)
    CH = 1 << 20
    for s0 in range(0, R, CH):
        st = starts[s0:s0 + CH]
        codes = genome[st[:, None] + (trim + np.arange(m))[None, :]]
        f = flip[s0:s0 + CH]
        codes[f] = (3 - codes[f])[:, ::-1]
        words[2 * s0 + 1: 2 * (s0 + len(st)) + 1: 2, :9] = alga_amd.pack_reads(codes)
        words[2 * s0: 2 * (s0 + len(st)): 2, :9] = alga_amd.pack_reads((3 - codes)[:, ::-1])
        if (s0 // CH) % 8 == 0:
            print("packed %d / %d reads, %.0f s" % (s0 + len(st), R, time.time() - t0), flush=True)
    lens = np.full(N, m, dtype=np.int32)
    lo, rs = alga_amd.derive_params(float(m))
    print("node set: %d nodes, %.1f GB, built in %.0f s; min_overlap %d rsoemo %d" % (N, words.nbytes / 1e9, time.time() - t0, lo, rs), flush=True)
    eng = alga_amd.Engine(0)
    dw = torch.from_numpy(words.view(np.int32)).cuda()
    dl = torch.from_numpy(lens).cuda()
    del words
    out = dict(reads=n, unique_reads=R, nodes=N, genome=G, min_overlap=lo, rsoemo=rs, src_sha256=alga_amd.engine.source_fingerprint())
    from alga_amd.engine import device_view
    keep = {}
    for red in forms:
        ms = []
        for it in range(steps):
            ptr, E = eng.prefsuf_device(dw, dl, lo, rs, collect_stats=(it == 0), reduction=red)
            st = eng.last_stats()
            if it == 0:
                out[red + "_counters"] = {k: st[k] for k in ("raw_overlaps", "records", "edges", "generic_sources", "windows_probed")}
            ms.append({k: round(st[k], 3) for k in ("ms_total", "ms_seed", "ms_probe", "ms_group", "ms_reduce", "ms_emit")})
            print(red, it, E, ms[-1], flush=True)
        out[red] = ms
        # the build whose edges are kept (and dumped): the last one -- with steps >= 2 one WITHOUT the work counters, i.e. through the pile path where it applies
        out[red + "_last_build"] = {k: st[k] for k in ("probe_used", "deferred_sources", "pile_buckets", "pile_irregular", "ms_pile")}
        keep[red] = device_view(ptr, (E, 3), dw.device).clone()
    if len(forms) == 2:
        out["forms_agree"] = bool(keep[forms[0]].shape == keep[forms[1]].shape and torch.equal(keep[forms[0]], keep[forms[1]]))
    e = keep[forms[0]]
    out["edges"] = int(e.shape[0])
    out["edges_per_sec_" + forms[0]] = out["edges"] / (out[forms[0]][-1]["ms_total"] * 1e-3)
    out["gbp_per_sec_" + forms[0]] = n * 150 / (out[forms[0]][-1]["ms_total"] * 1e-3) / 1e9
    if ref_threads > 0:
        gpu_edges = e.cpu().numpy().astype(np.int32)
        out["reference"] = run_reference(genome, starts, flip, L, ref_threads, eng, N, gpu_edges)
        out["reference"]["edges_equal_gpu"] = out["reference"].get("edges") == out["edges"]
    elif ref_sha:
        import tempfile
        gpu_edges = e.cpu().numpy().astype(np.int32)
        with tempfile.TemporaryDirectory(dir=os.environ.get("ALGA_TMP", None)) as wd:
            mine = os.path.join(wd, "gpu.graph")
            eng.write_graph(mine, N, gpu_edges)
            out["gpu_dump_bytes"], out["gpu_dump_sha256"] = os.path.getsize(mine), file_sha256(mine)
        out["reference_dump_sha256_recorded"] = ref_sha
        out["dump_byte_identical_to_recorded_reference"] = out["gpu_dump_sha256"] == ref_sha
    print(json.dumps(out))


def file_sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def run_reference(genome, starts, flip, L, threads, eng, n_nodes, gpu_edges):
    import glob
    import re
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ALGA")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/ALGA is not built"}
    res = {"threads": threads}
    with tempfile.TemporaryDirectory(dir=os.environ.get("ALGA_TMP", None)) as wd:
        t0 = time.time()
        lut = np.frombuffer(b"ACGT", dtype=np.uint8)
        path = os.path.join(wd, "s.fasta")
        with open(path, "wb") as f:
            CH = 1 << 20
            for s0 in range(0, len(starts), CH):
                st = starts[s0:s0 + CH]
                codes = genome[st[:, None] + np.arange(L)[None, :]]
                fl = flip[s0:s0 + CH]
                codes[fl] = (3 - codes[fl])[:, ::-1]
                n = len(st)
                rec = np.empty((n, 12 + L + 1), dtype=np.uint8)          # ">" + 10 digits + "\n" + sequence + "\n"
                rec[:, 0] = ord(">")
                ids = np.arange(s0, s0 + n)
                for k in range(10):
                    rec[:, 10 - k] = ord("0") + (ids // 10 ** k) % 10
                rec[:, 11] = ord("\n")
                rec[:, 12:12 + L] = lut[codes]
                rec[:, 12 + L] = ord("\n")
                f.write(rec.tobytes())
        res["fasta_write_s"] = time.time() - t0
        print("FASTA written in %.0f s; starting the reference with %d threads" % (res["fasta_write_s"], threads), flush=True)
        t = time.time()
        p = subprocess.Popen([exe, "--file1=s.fasta", "--threads=%d" % threads, "--output=o.fasta", "--serialize=1"], cwd=wd, stdout=subprocess.DEVNULL,
                             stderr=subprocess.PIPE, text=True, errors="replace")
        last = time.time()
        for line in p.stderr:
            if "Creating GraphCreator" in line:
                res["to_graph_creator_s"] = time.time() - t
            if "After Iteration" in line and time.time() - last > 30:
                print("reference:", line.strip()[:100], "at %.0f s" % (time.time() - t), flush=True)
                last = time.time()
            if "Before first simplifier" in line:
                res["to_graph_done_s"] = time.time() - t
                m = re.search(r"(\d+) edges", line)
                res["edges"] = int(m.group(1)) if m else None
                p.kill()
                break
        p.wait()
        if "to_graph_done_s" in res and "to_graph_creator_s" in res:
            res["graph_creator_s"] = res["to_graph_done_s"] - res["to_graph_creator_s"]
        dumps = glob.glob(os.path.join(wd, "*_beforeSimplifier.graph"))
        if dumps:
            os.unlink(path)                                                  # room for the second dump
            mine = os.path.join(wd, "gpu.graph")
            eng.write_graph(mine, n_nodes, gpu_edges)
            res["dump_bytes"], res["gpu_dump_bytes"] = os.path.getsize(dumps[0]), os.path.getsize(mine)
            res["dump_sha256"], res["gpu_dump_sha256"] = file_sha256(dumps[0]), file_sha256(mine)
            res["dump_byte_identical"] = res["dump_bytes"] == res["gpu_dump_bytes"] and res["dump_sha256"] == res["gpu_dump_sha256"]
            print("dumps: reference %d bytes %s, engine %d bytes %s" % (res["dump_bytes"], res["dump_sha256"][:16], res["gpu_dump_bytes"], res["gpu_dump_sha256"][:16]), flush=True)
    return res


if __name__ == "__main__":
    main()
