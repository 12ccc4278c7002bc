#!/bin/bash
# One iteration on the GPU box: parity tests of the exact path, then a kernel trace (and optional counter passes) of the north-star bench.
# usage: tools/r03_iter.sh <tag> [passes (default t)] [pytest-args...]
set -u
TAG=$1; shift
PASSES=${1:-t}; [ $# -gt 0 ] && shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $REPO/gpurun_out
cd $REPO
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu "$@" > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 bash tools/profile_cmd.sh $TAG $PASSES bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie || exit 1
grep -E "k_|rocprim" gpurun_out/prof_$TAG/summary.txt | head -60
tail -1 gpurun_out/prof_$TAG/out_trace.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('ms_per_step','phases_ms','roofline','counters')}))"
