#!/bin/bash
# kernel trace of the default bench (timed form only) -> per-kernel averages
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
TAG=${1:-t}; shift || true
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py --traffic-pass --steps 6 --warmup 2 "$@" > $OUT/trace_$TAG.json 2> $OUT/trace_$TAG.err || { tail -5 $OUT/trace_$TAG.err; exit 1; }
cd $REPO
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$f")) if ("alga" in r["Name"] or "rocprim" in r["Name"])]
tot=0
for r in rows:
    n=r["Name"].replace("void ","").replace("alga::","").replace("(anonymous namespace)::","").split("(")[0][:60]
    c=int(r["Calls"]); a=float(r["AverageNs"])/1e3
    if c*a/8 > 20: print("%-62s calls %4d avg %9.1f us  per step %8.1f us" % (n, c, a, c*a/8))
    tot+=c*a/8
print("sum per step (us):", round(tot,1))
PY
cat $OUT/trace_$TAG.json | cut -c1-300
