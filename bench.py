#!/usr/bin/env python3
"""bench.py -- overlap-graph construction throughput of the HIP engine on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2_1M_150bp] [--no-cpu-baseline]

A "step" is one complete pass of the hot path (GraphCreatorPrefSuf: seed table, probe, per-source cap,
per-target transitive reduction, sorted adjacency) over one synthetic read set that is already resident in
HBM when the timed region starts.  Prints ONE JSON line (rank 0):
  metric   overlap_edges_per_sec   (BASELINE.json: overlap edges/sec; Gbp/s of input reads is `gbp_per_sec`)
  value    edges in the emitted graph x steps / wall time of the K steps (max over ranks), whole job
  roofline dominant kernel (k_probe_sources) algorithmic bytes / its HIP-event duration vs the 8 TB/s HBM peak
  cpu_baseline  the real reference binary (oracle/_ref/ALGA, kind "reference") or, if it is absent, the C
                oracle (kind "port"), timed on this box's host cores on a bounded sample of the same workload.

N=1: BASELINE.json configs[1] (1 M x 150 bp, error-free, 50x coverage).  N>1 (one process per GPU under
torch.distributed.run, RCCL): weak scaling -- N x 1 M reads over an N x 3 Mb genome; sources are sharded
across ranks, overlap records are exchanged to the rank that owns the target (all_to_all), edges are
all-gathered.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)


def algorithmic_bytes(st, W):
    """packed reads once + one 16 B table probe per (node, overlap length) +
    one packed-read fetch per raw candidate (probe kernel); one packed-read fetch per transitive compare
    (reduce kernel); 12 B per emitted edge."""
    probe = st["nodes_live"] * 4 * W + st["windows_probed"] * 16 + st["raw_overlaps"] * 4 * W
    reduce_ = st["transitive_compares"] * 4 * W
    emit = st["edges"] * 12
    return dict(probe=probe, reduce=reduce_, emit=emit, total=probe + reduce_ + emit)


def cpu_baseline_reference(codes, threads, budget_reads):
    """Run the real reference (oracle/_ref/ALGA) on a bounded sample; creator wall time is taken between its
    stderr markers 'Creating GraphCreator' and 'Before first simplifier' (its own timers report CPU-seconds,
    src/Utils/TimeMeasurer.cpp:26-39)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "ALGA")
    if not os.path.exists(exe):
        return None
    from alga_amd import workload
    sample = codes[:budget_reads]
    with tempfile.TemporaryDirectory() as wd:
        workload.write_fasta_fast(os.path.join(wd, "s.fasta"), sample)
        p = subprocess.Popen([exe, "--file1=s.fasta", "--threads=%d" % threads, "--output=o.fasta"], cwd=wd,
                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, errors="replace", bufsize=1)
        t0 = t1 = None
        edges = None
        for line in p.stderr:
            if t0 is None and "Creating GraphCreator" in line:
                t0 = time.perf_counter()
            m = re.search(r"Before first simplifier graph has (\d+) edges", line)
            if m:
                t1 = time.perf_counter()
                edges = int(m.group(1))
                break
        p.kill()
        p.wait()
    if t0 is None or t1 is None:
        return None
    dt = t1 - t0
    return dict(value=edges / dt, unit="edges/s", cores=threads, kind="reference",
                sample="%d x %d bp reads of the same workload (first reads of the set), ALGA --threads=%d, creator region "
                       "src/main.cpp:244-296, %.2f s wall, %d edges" % (len(sample), sample.shape[1], threads, dt, edges),
                seconds=dt, edges=edges, gbp_per_sec=sample.size / dt / 1e9)


def cpu_baseline_port(words, lens, lo, rs, budget_nodes):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    import oracle_lib as O
    w, l = words[:budget_nodes], lens[:budget_nodes]
    t0 = time.perf_counter()
    e, _, _ = O.prefsuf(w, l, lo, rs)
    dt = time.perf_counter() - t0
    return dict(value=len(e) / dt, unit="edges/s", cores=1, kind="port",
                sample="first %d nodes of the workload, single-thread C oracle, %.2f s" % (len(l), dt), seconds=dt, edges=len(e))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg2_1M_150bp")
    ap.add_argument("--stride", type=int, default=0, help="row stride in uint32 words (0 = the engine's HBM layout: rows padded to 16 bytes, as alga_prefsuf_build_host uploads them; -1 = minimal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=1_000_000)
    args = ap.parse_args()

    import torch
    import alga_amd
    from alga_amd import workload
    from alga_amd import multigpu

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # ---- workload (synthetic; built once on rank 0, replicated to every GPU over RCCL) -------------------------
    wl = None
    meta = torch.zeros(4, dtype=torch.int64, device="cuda")
    if rank == 0:
        wl = workload.build(args.config, scale=world, stride_words=(None if args.stride < 0 else (args.stride or "aligned")))
        meta = torch.tensor([len(wl["lens"]), wl["words"].shape[1], wl["min_overlap"], wl["rsoemo"]], dtype=torch.int64, device="cuda")
    if dist is not None:
        dist.broadcast(meta, src=0)
    n_nodes, stride, lo, rs = [int(x) for x in meta.cpu()]
    if rank == 0:
        d_words = torch.from_numpy(wl["words"].view(np.int32)).cuda()
        d_lens = torch.from_numpy(wl["lens"]).cuda()
    else:
        d_words = torch.empty((n_nodes, stride), dtype=torch.int32, device="cuda")
        d_lens = torch.empty(n_nodes, dtype=torch.int32, device="cuda")
    if dist is not None:
        dist.broadcast(d_words, src=0)
        dist.broadcast(d_lens, src=0)
    W = (2 * int(d_lens.max().item()) + 31) // 32 if n_nodes else 0
    eng = alga_amd.Engine(local_rank)
    runner = multigpu.ShardedPrefSuf(multigpu.HipBackend(eng, d_words, d_lens, lo, rs), rank, world, dist)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # one counted pass (work counters for the roofline's algorithmic bytes); not timed
    n_edges, st = runner.step(collect_stats=True)
    stats = dict(st)
    if world > 1:
        stats["nodes_live"] = n_nodes                 # whole-job counters (all_reduced); nodes are replicated
    for _ in range(max(0, args.warmup - 1)):
        runner.step()
    sync_all()
    probe_ms = []
    phase = dict(seed=0.0, probe=0.0, group=0.0, reduce=0.0, emit=0.0, exchange=0.0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_edges, s = runner.step()
        probe_ms.append(s["ms_probe"])
        for k in phase:
            phase[k] += s.get("ms_" + k, 0.0)
    sync_all()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        bases = wl["n_reads"] * wl["read_len"]
        alg = algorithmic_bytes(stats, W)
        probe_avg_ms = float(np.mean(probe_ms))
        alg_probe_launch = alg["probe"] / world           # one launch = one rank's share of the sources
        achieved = alg_probe_launch / (probe_avg_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "probe_hbm_bytes.json")   # from the rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.config, {}).get("hbm_bytes_per_launch") if world == 1 else None
            except Exception:
                traffic = None
        out = {
            "metric": "overlap_edges_per_sec", "value": n_edges * args.steps / dt, "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "gbp_per_sec": bases / (ms_step * 1e-3) / 1e9,
            "config": {"workload": "%s x%d: %d x %d bp reads, genome %d, seed %d, err %.2f -> %d nodes (both strands, "
                                   "duplicates removed), min_overlap %d, rsoemo %d" %
                                   (wl["name"], world, wl["n_reads"], wl["read_len"], wl["genome"], wl["seed"], wl["err"],
                                    n_nodes, lo, rs),
                       "nodes": n_nodes, "edges": int(n_edges), "parallelism": "1 GPU" if world == 1 else
                       "sources sharded over %d ranks, each builds the final edges of its sources (no record exchange), "
                       "edge lists gathered on rank 0 over RCCL" % world},
            "roofline": {"bound": "hbm", "kernel": "k_probe_sources", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes": alg_probe_launch,
                         "kernel_ms": probe_avg_ms,
                         "per_unit": "per source node: 4W + 16 P + 4W * raw/node bytes (W=%d words, P=%.1f windows, raw/node=%.2f)" %
                                     (W, stats["windows_probed"] / max(1, stats["nodes_live"]), stats["raw_overlaps"] / max(1, stats["nodes_live"]))},
            "phases_ms": {k: v / args.steps for k, v in phase.items()},
            "counters": {k: int(stats[k]) for k in ("nodes_live", "windows_probed", "slots_scanned", "raw_overlaps", "records",
                                                    "transitive_listed", "transitive_compares", "transitive_removed", "edges",
                                                    "max_in_records", "table_slots")},
            "algorithmic_bytes_total": alg["total"],
            "device": eng.device_name(),
        }
        if world == 1:
            # the same graph through the host-buffer entry point (pageable host arrays in, host edge list out): never `value`
            ts = []
            for _ in range(3):
                t = time.perf_counter()
                he = eng.prefsuf_host(wl["words"], wl["lens"], lo, rs)
                ts.append(time.perf_counter() - t)
            out["pcie_inclusive"] = {"ms_per_graph": min(ts) * 1e3, "edges_per_sec": len(he) / min(ts),
                                     "note": "alga_prefsuf_build_host: H2D of the packed reads + build + D2H of the edges, best of 3"}
        if not args.no_cpu_baseline and world == 1:        # the CPU baseline is a rank-0, N=1 measurement
            try:
                cores = len(os.sched_getaffinity(0))
            except AttributeError:
                cores = os.cpu_count() or 1
            cores = max(1, min(cores, 16))          # the GPU box gives one GPU's CPU share: 16 cores
            cb = cpu_baseline_reference(wl["codes"], cores, min(args.cpu_sample_reads, wl["n_reads"]))
            if cb is None:
                cb = cpu_baseline_port(wl["words"], wl["lens"], lo, rs, min(n_nodes, 200_000))
            out["cpu_baseline"] = cb
            if cb.get("kind") == "reference" and world == 1 and args.cpu_sample_reads >= wl["n_reads"]:
                out["cpu_baseline"]["edges_equal_gpu"] = bool(cb["edges"] == int(n_edges))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
