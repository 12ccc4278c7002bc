#!/usr/bin/env python3
"""On an N-GPU node: the graph gathered from N ranks (alga_amd.multigpu.ShardedPrefSuf over RCCL, exactly as bench.py --gpus N drives
it) must equal the graph one GPU builds alone, byte for byte.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/multigpu_check.py [n_reads] [genome]

Every rank generates the same seeded read set on its own GPU; rank 0 builds the whole graph once more by itself and compares.
Prints one JSON line on rank 0: edges, equal, per-rank probe / exchange times.

Rehearsal on ONE GPU (no RCCL: N ranks as threads of one process, one engine each, the collectives replaced by the thread
rendezvous of tests/fake_dist.py -- the driver, the engine calls and the sharded key pass are the real ones):

  python tools/multigpu_check.py --rehearse N [n_reads] [genome]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import multigpu, workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402


def rehearse(world, n, G, shard_keys=True):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_dist import run_ranks
    torch.cuda.set_device(0)
    wl = workload.device_build(n, 150, G, 11)

    def rank_main(rank, dist):
        eng = alga_amd.Engine(0)
        try:
            runner = multigpu.ShardedPrefSuf(multigpu.HipBackend(eng, wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]), rank, world, dist,
                                             shard_keys=shard_keys)
            m, st = runner.step()
            return int(m), (runner.edges.clone() if rank == 0 else None), st
        finally:
            eng.close()
    res = run_ranks(world, rank_main)
    eng = alga_amd.Engine(0)
    ptr, m1 = eng.prefsuf_device(wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"], stream=torch.cuda.current_stream().cuda_stream)
    alone = device_view(ptr, (m1, 3), wl["words"].device)
    gathered = res[0][1]
    b = multigpu.shard_bounds(int(wl["lens"].shape[0]), world)
    per_rank = [(int(((gathered[:, 0] >= b[r]) & (gathered[:, 0] < b[r + 1])).sum()), int(((alone[:, 0] >= b[r]) & (alone[:, 0] < b[r + 1])).sum()))
                for r in range(world)]
    out = dict(rehearsal_ranks_on_one_gpu=world, edges_per_rank_gathered_vs_alone=per_rank, reads=n, nodes=int(wl["lens"].shape[0]), edges_gathered=res[0][0], edges_one_gpu=int(m1),
               equal=bool(gathered.shape == alone.shape and torch.equal(gathered, alone)),
               keys_shared=[bool(r[2].get("ms_keys_shared", 0.0) > 0) for r in res])
    print(json.dumps(out))
    sys.exit(0 if out["equal"] else 1)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--rehearse":
        world = int(sys.argv[2])
        rest = [a for a in sys.argv[3:] if a != "--no-shard-keys"]
        return rehearse(world, int(rest[0]) if rest else 2_000_000, int(rest[1]) if len(rest) > 1 else 10_000_000, "--no-shard-keys" not in sys.argv)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 40_000_000
    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    wl = workload.device_build(n, 150, G, 11)
    eng = alga_amd.Engine(local)
    runner = multigpu.ShardedPrefSuf(multigpu.HipBackend(eng, wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"]), rank, world, dist)
    m, st = runner.step()
    out = None
    if rank == 0:
        gathered = runner.edges.clone()
        ptr, m1 = eng.prefsuf_device(wl["words"], wl["lens"], wl["min_overlap"], wl["rsoemo"], stream=torch.cuda.current_stream().cuda_stream)
        alone = device_view(ptr, (m1, 3), wl["words"].device)
        out = dict(n_gpus=world, reads=n, nodes=int(wl["lens"].shape[0]), edges_gathered=int(m), edges_one_gpu=int(m1),
                   equal=bool(gathered.shape == alone.shape and torch.equal(gathered, alone)),
                   rank0_ms={k: round(st.get(k, 0.0), 3) for k in ("ms_keys_shared", "ms_seed", "ms_probe", "ms_emit", "ms_exchange")})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        sys.exit(0 if out["equal"] else 1)


if __name__ == "__main__":
    main()
