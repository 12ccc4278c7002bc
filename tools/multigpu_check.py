#!/usr/bin/env python3
"""On an N-GPU node: the graph gathered from N ranks (alga_amd.multigpu.ShardedPrefSuf over RCCL, exactly as bench.py --gpus N drives
it) must equal the graph one GPU builds alone, byte for byte.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 tools/multigpu_check.py [n_reads] [genome]

Every rank generates the same seeded read set on its own GPU; rank 0 builds the whole graph once more by itself and compares.
Prints one JSON line on rank 0: edges, equal, per-rank probe / exchange times."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import alga_amd  # noqa: E402
from alga_amd import multigpu, workload  # noqa: E402
from alga_amd.engine import device_view  # noqa: E402


def main():
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
                   rank0_ms={k: round(st.get(k, 0.0), 3) for k in ("ms_seed", "ms_probe", "ms_emit", "ms_exchange")})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        sys.exit(0 if out["equal"] else 1)


if __name__ == "__main__":
    main()
