#!/bin/bash
# kernel durations of the 16-byte-record sort A/B (tools/sort_ab64.py)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_s64
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_s64 -- python3 $REPO/tools/sort_ab64.py > $OUT/s64.json 2> $OUT/s64.err || { tail -5 $OUT/s64.err; exit 1; }
cat $OUT/s64.json
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/prof_s64/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:90], r["Calls"], r["AverageNs"], r["MinNs"])
PY
