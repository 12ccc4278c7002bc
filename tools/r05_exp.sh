#!/bin/bash
# one short bench line of the build at hand: step, dominant kernel, deferred sources (experiments)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $REPO/gpurun_out/r05; cd $REPO
TAG=${1:-exp}; shift
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pcie --no-first-call "$@" > gpurun_out/r05/$TAG.json 2> gpurun_out/r05/$TAG.err || { tail -5 gpurun_out/r05/$TAG.err; exit 1; }
python3 -c "
import json
d=json.loads(open('gpurun_out/r05/$TAG.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d['roofline'].get('pile_path',{}).get('deferred_sources'), d['phases_ms'], d.get('index_build_ms'), d.get('timed_edges_digest_equal_pairwise'))"
