#!/bin/bash
# sort tests + A/B + kernel trace of the A/B (run through gpurun)
set -u
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
timeout -k 10 600 python -m pytest tests/test_gpu_sort.py -x -q -m gpu > $OUT/pytest_sort.log 2>&1; rc=$?
tail -3 $OUT/pytest_sort.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/sort_ab.py 90621096 0,1 > $OUT/sort_ab.json 2> $OUT/sort_ab.err || { tail -5 $OUT/sort_ab.err; exit 1; }
cat $OUT/sort_ab.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_sort
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sort -- python3 $REPO/tools/sort_ab.py 90621096 0,1 > $OUT/sort_ab_prof.json 2> $OUT/sort_ab_prof.err || { tail -5 $OUT/sort_ab_prof.err; exit 1; }
cd $REPO
f=$(find $OUT/prof_sort -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
for r in csv.DictReader(open("$f")):
    n=r["Name"]
    if "rs_" in n: print(n.split("(")[0][-45:], r["Calls"], r["AverageNs"])
PY
