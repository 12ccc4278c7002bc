#!/bin/bash
# kernel trace of the default bench (timed form) -> the last step's timeline; then the cfg5 bench line (short)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
TAG=${1:-tl}
mkdir -p $OUT
cd $REPO
bash tools/r05_trace.sh $TAG > $OUT/trace_$TAG.txt 2>&1 || { tail -5 $OUT/trace_$TAG.txt; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_node_stats" in r["Kernel_Name"]]
start=idx[-1]; t0=int(rows[start]["Start_Timestamp"]); pe=t0
for r in rows[start:]:
    n=r["Kernel_Name"].replace("void ","").replace("alga::","").replace("(anonymous namespace)::","").split("(")[0][:44]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if (e-s)>30000 or (s-pe)>15000: print("%-46s start %8.1f dur %8.1f gap %6.1f" % (n,(s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3))
    pe=e
print("step total", (pe-t0)/1e3)
PY
if [ "${2:-}" = cfg5 ]; then
  timeout -k 10 500 python bench.py --config cfg5_10M_150bp_err2 --steps 5 --warmup 2 --no-cpu-baseline --no-first-call > $OUT/bench_cfg5_$TAG.json 2> $OUT/bench_cfg5_$TAG.err || { tail -5 $OUT/bench_cfg5_$TAG.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$OUT/bench_cfg5_$TAG.json").read().strip().splitlines()[-1])
print("cfg5 ms_per_step", d["ms_per_step"], "supplement", d["supplement"]["ms"], "exact", d["supplement"]["exact_path_ms"], "digest", d["timed_edges_digest_equal_pairwise"], d["index_build_ms"])
PY
fi
