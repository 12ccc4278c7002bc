#!/bin/bash
# bench.py exactly as the driver runs it (no flags): wall time, the contract's fields, the roofline basis
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $REPO/gpurun_out/r05_final; cd $REPO
T0=$(date +%s.%N)
python bench.py > gpurun_out/r05_final/bench_default.json 2> gpurun_out/r05_final/bench_default.err || { tail -5 gpurun_out/r05_final/bench_default.err; exit 1; }
T1=$(date +%s.%N)
echo "wall $(echo "$T1 - $T0" | bc) s"
python3 - <<PY
import json
d=json.loads(open("gpurun_out/r05_final/bench_default.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data")})
print(d["config"]["workload"][:80]); print(d["roofline"]["basis"], d["roofline"]["frac"], d["roofline"]["traffic"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"], d["fits_in_driver_run"])
PY
