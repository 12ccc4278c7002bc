#!/bin/bash
# kernel trace of the configs[4] bench -> the supplement of the last step: where the time between its kernels goes
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
TAG=${1:-gaps}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$TAG -- python3 $REPO/bench.py --config cfg5_10M_150bp_err2 --no-cpu-baseline --no-pcie --no-first-call --steps 3 --warmup 2 > $OUT/trace_$TAG.json 2> $OUT/trace_$TAG.err || { tail -5 $OUT/trace_$TAG.err; exit 1; }
cd $REPO
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_$TAG/**/*kernel_trace.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].replace("void ","").replace("alga::","").replace("(anonymous namespace)::","").split("(")[0][:48]
idx=[i for i,r in enumerate(rows) if "k_pkb_rowptr" in r["Kernel_Name"]]
start=idx[-1]
end=max(i for i,r in enumerate(rows) if "k_pkb_keys_to_edges" in r["Kernel_Name"])
seg=rows[start:end+1]
t0=int(seg[0]["Start_Timestamp"]); t1=int(seg[-1]["End_Timestamp"])
ker=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in seg)
small={}; big=0; gaps_big=[]; gaps_small=0; n_small_gaps=0
pe=None
for r in seg:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if pe is not None:
        g=s-pe
        if g>=10000: gaps_big.append((round(g/1e3,1), prev, nm(r)))
        elif g>0: gaps_small+=g; n_small_gaps+=1
    if e-s<20000: small[nm(r)]=small.get(nm(r),[0,0]); small[nm(r)][0]+=(e-s)/1e3; small[nm(r)][1]+=1
    pe=max(pe or 0,e); prev=nm(r)
print("supplement span us", (t1-t0)/1e3, "kernels", len(seg), "kernel sum", ker/1e3)
print("gaps >= 10 us:", len(gaps_big), "sum", round(sum(g for g,_,_ in gaps_big),1))
print("gaps < 10 us:", n_small_gaps, "sum", round(gaps_small/1e3,1))
print("kernels under 20 us: sum", round(sum(v[0] for v in small.values()),1))
for k,v in sorted(small.items(), key=lambda kv:-kv[1][0]): print("   %-50s %7.1f us in %d calls" % (k,v[0],v[1]))
for g in gaps_big: print("   gap", g)
PY
