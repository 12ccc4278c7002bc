#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes of `bench.py`.
# usage: tools/profile_bench.sh <tag> [bench args...]
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/trace.err" || { tail -20 "$OUT/trace.err"; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_sq.json" 2> "$OUT/pmc_sq.err" || { tail -20 "$OUT/pmc_sq.err"; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || { tail -20 "$OUT/pmc_fetch.err"; exit 1; }
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_write.json" 2> "$OUT/pmc_write.err" || { tail -20 "$OUT/pmc_write.err"; exit 1; }
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mem" -- python3 "$REPO/bench.py" $ARGS > "$OUT/bench_pmc_mem.json" 2> "$OUT/pmc_mem.err" || { tail -20 "$OUT/pmc_mem.err"; exit 1; }
find "$OUT" -name "*.csv" | head -40
