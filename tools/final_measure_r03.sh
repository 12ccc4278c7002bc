#!/bin/bash
# Round 3's measurement suite on one GPU box (run through gpurun; everything lands under gpurun_out/r03_final/):
#   part A: the whole -m gpu suite; kernel trace + memory-request counters of the default bench (BASELINE configs[3]) -> traffic file -> the bench
#           line itself (with first-call legs, PCIe leg, CPU baseline); the same for configs[4]; bench lines of configs[1], configs[2]
#   part B: one rank's compute at world sizes 2 / 4 / 8 on one GPU; the C++ N-rank driver at the north-star size against one engine
# usage: tools/final_measure_r03.sh A|B
set -u
PART=${1:-A}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r03_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
if [ "$PART" = A ]; then
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
  tail -3 $OUT/pytest_gpu.log
  [ $rc -ne 0 ] && exit $rc
  for CFG in cfg4_50M_150bp cfg5_10M_150bp_err2; do
    timeout -k 10 900 bash tools/profile_cmd.sh r03_final_$CFG trx bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-pcie --no-first-call || exit 1
    python tools/pmc_to_traffic.py gpurun_out/prof_r03_final_$CFG $CFG > $OUT/traffic_$CFG.json || exit 1
    cp gpurun_out/prof_r03_final_$CFG/summary.txt $OUT/rocprof_$CFG.txt
    timeout -k 10 600 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err || { tail -20 $OUT/bench_$CFG.err; exit 1; }
    echo "$CFG: $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$CFG.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'])")"
  done
  cp profiles/hbm_traffic.json $OUT/hbm_traffic.json
  for CFG in cfg2_1M_150bp cfg3_5M_150bp; do
    timeout -k 10 600 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err || { tail -20 $OUT/bench_$CFG.err; exit 1; }
    echo "$CFG: $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$CFG.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])")"
  done
else
  : > $OUT/emulated_rank_compute.jsonl
  for N in 2 4 8; do
    timeout -k 10 300 python tools/emulate_rank.py $N 50000000 250000000 5 1 >> $OUT/emulated_rank_compute.jsonl 2> $OUT/emulate_$N.err || { tail -5 $OUT/emulate_$N.err; exit 1; }
  done
  cat $OUT/emulated_rank_compute.jsonl | cut -c1-400
  timeout -k 10 600 python tools/multi_cxx_check.py 2 cfg4_50M_150bp > $OUT/multi_cxx_2ranks_50M.json 2> $OUT/multi_cxx.err || { tail -5 $OUT/multi_cxx.err; exit 1; }
  cut -c1-900 $OUT/multi_cxx_2ranks_50M.json
  timeout -k 10 600 python tools/multi_cxx_check.py 4 cfg2_1M_150bp > $OUT/multi_cxx_4ranks_1M.json 2>> $OUT/multi_cxx.err || { tail -5 $OUT/multi_cxx.err; exit 1; }
  cut -c1-600 $OUT/multi_cxx_4ranks_1M.json
fi
