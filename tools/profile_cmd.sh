#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of one python tool of this repo.
# usage: tools/profile_cmd.sh <tag> <passes: t=trace,s=sq,a=active,f=fetch,w=write,m=mem,r=read requests by size,x=write requests by size,c=TA/L1,d=L2> <script> [args...]
# Outputs under gpurun_out/prof_<tag>/ ; summarise with tools/summarize_prof.py and copy what is to be judged into profiles/.
set -u
TAG=$1; PASSES=$2; SCRIPT=$3; shift 3
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, rocprof args...
  local name=$1; shift
  timeout -k 10 ${PROFILE_PASS_TIMEOUT:-240} rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$REPO/$SCRIPT" $ARGS > "$OUT/out_$name.log" 2> "$OUT/$name.err" || { tail -20 "$OUT/$name.err"; exit 1; }
}
ARGS="$*"
case $PASSES in *t*) run trace --kernel-trace --stats ;; esac
case $PASSES in *s*) run pmc_sq --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU ;; esac
case $PASSES in *a*) run pmc_act --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES ;; esac
case $PASSES in *f*) run pmc_fetch --pmc FETCH_SIZE ;; esac
case $PASSES in *w*) run pmc_write --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum ;; esac
# gfx950 counts the memory-side requests of the L2 by size: bytes = 32 * N_32B + 64 * N_64B + 128 * N_128B (FETCH_SIZE, which falls back
# to the gfx94x formula, tallies a 128-byte request as 64 bytes)
case $PASSES in *r*) run pmc_rdreq --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum ;; esac
case $PASSES in *x*) run pmc_wrreq --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_sum ;; esac
case $PASSES in *c*) run pmc_l1 --pmc TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum ;; esac
case $PASSES in *d*) run pmc_l2 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_BUSY_avr ;; esac
case $PASSES in *i*) run pmc_ic --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE ;; esac
case $PASSES in *l*) run pmc_lds --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS ;; esac
case $PASSES in *m*) run pmc_mem --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE ;; esac
python3 "$REPO/tools/summarize_prof.py" "$OUT" > "$OUT/summary.txt"
