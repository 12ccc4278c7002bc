#!/usr/bin/env python3
"""bench.py's per-rank body (bench.run) as N thread-ranks on ONE GPU, the collectives replaced by the thread rendezvous of
tests/fake_dist.py: the N > 1 branch of the script at any size, on a one-GPU box.  Not a measurement -- the ranks share the GPU.
  tools/rehearse_bench.py [N=4] [config=cfg4_50M_150bp] [steps=2]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from fake_dist import run_ranks  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
config = sys.argv[2] if len(sys.argv) > 2 else "cfg4_50M_150bp"
steps = sys.argv[3] if len(sys.argv) > 3 else "2"
args = bench.parse_args(["--gpus", str(N), "--steps", steps, "--warmup", "1", "--config", config])
res = run_ranks(N, lambda rank, dist: bench.run(args, rank, N, 0, dist))
out = res[0]
print(json.dumps({k: out[k] for k in ("n_gpus", "ms_per_step", "value", "config", "multi_gpu_form", "scaling")}))
