#!/bin/bash
# usage: tools/pmc_probe.sh <tag> : instruction-mix counters of the probe kernel (separate PMC pass, no tracing)
set -u
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/a" -- python3 "$REPO/bench.py" $ARGS > "$OUT/a.json" 2> "$OUT/a.err" || { tail -20 "$OUT/a.err"; exit 1; }
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/b" -- python3 "$REPO/bench.py" $ARGS > "$OUT/b.json" 2> "$OUT/b.err" || { tail -20 "$OUT/b.err"; exit 1; }
