#!/bin/bash
# kernel trace of the configs[4] bench (timed form) -> the last step's timeline (kernels >= 20 us, gaps >= 10 us)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
TAG=${1:-c5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py --config cfg5_10M_150bp_err2 --traffic-pass --steps 3 --warmup 2 > $OUT/trace_$TAG.json 2> $OUT/trace_$TAG.err || { tail -5 $OUT/trace_$TAG.err; exit 1; }
cd $REPO
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_node_stats" in r["Kernel_Name"]]
start=idx[-1]; t0=int(rows[start]["Start_Timestamp"]); pe=t0
acc={}
for r in rows[start:]:
    n=r["Kernel_Name"].replace("void ","").replace("alga::","").replace("(anonymous namespace)::","").split("(")[0][:60]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if (e-s)>20000 or (s-pe)>10000: print("%-62s start %8.1f dur %8.1f gap %6.1f" % (n,(s-t0)/1e3,(e-s)/1e3,(s-pe)/1e3))
    acc[n]=acc.get(n,0)+(e-s)/1e3
    pe=e
print("step total", (pe-t0)/1e3)
print("by kernel:", sorted(((round(v,1),k) for k,v in acc.items()), reverse=True)[:25])
PY
