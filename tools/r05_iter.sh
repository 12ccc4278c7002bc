#!/bin/bash
# Round 5 iteration helper (run through gpurun): a subset of the GPU suite + the default bench line.
# usage: tools/r05_iter.sh <tag> "<pytest args>" [bench args...]
set -u
TAG=$1; PYT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
if [ -n "$PYT" ]; then
  timeout -k 10 1000 python -m pytest $PYT -x -q -m gpu > $OUT/pytest_$TAG.log 2>&1; rc=$?
  tail -5 $OUT/pytest_$TAG.log
  [ $rc -ne 0 ] && exit $rc
fi
if [ $# -gt 0 ]; then
  timeout -k 10 600 python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; rc=$?
  [ $rc -ne 0 ] && { tail -20 $OUT/bench_$TAG.err; exit $rc; }
  python3 - <<PY
import json
d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1])
print("ms_per_step", d['ms_per_step'], "value", d['value'], "digest_equal", d.get('timed_edges_digest_equal_pairwise'))
print("roofline", {k: d['roofline'].get(k) for k in ('kernel','frac','basis','kernel_ms','frac_pairwise_equivalent')})
print("roofline_step", d.get('roofline_step'))
print("index_build_ms", d.get('index_build_ms'), "phases", d.get('phases_ms'))
pi=d.get("pcie_inclusive",{}); print("e2e", d.get("ms_end_to_end"), d.get("end_to_end_form"), "triples", pi.get("ms_per_graph"), pi.get("phases_ms"), "compact", pi.get("compact_edges",{}).get("ms_per_graph"), pi.get("compact_edges",{}).get("phases_ms"), "pinned", pi.get("compact_edges",{}).get("pinned_node_arrays"))
PY
fi
[ -f $OUT/bench_$TAG.json ] && python3 - <<PY
import json
d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1])
if d.get("supplement"): print("supplement", {k: d["supplement"].get(k) for k in ("ms","exact_path_ms","groups","group_hist_2_3_4_7_15_31_64_more","can_align_calls","kmers")})
PY
exit 0
