#!/bin/bash
# Full GPU check of a build: the whole -m gpu suite, then the driver's bench line (default config) and the configs[4] line.
set -u
TAG=$1
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $REPO/gpurun_out
cd $REPO
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?
tail -4 gpurun_out/${TAG}_pytest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/${TAG}_bench_cfg4.json 2> gpurun_out/${TAG}_bench_cfg4.err || { tail -20 gpurun_out/${TAG}_bench_cfg4.err; exit 1; }
tail -c 6000 gpurun_out/${TAG}_bench_cfg4.json
timeout -k 10 600 python bench.py --config cfg5_10M_150bp_err2 > gpurun_out/${TAG}_bench_cfg5.json 2> gpurun_out/${TAG}_bench_cfg5.err || { tail -20 gpurun_out/${TAG}_bench_cfg5.err; exit 1; }
tail -c 4000 gpurun_out/${TAG}_bench_cfg5.json
