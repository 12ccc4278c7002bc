#!/bin/bash
# Round 5's measurement suite on one GPU box (run through gpurun; everything lands under gpurun_out/r05_final/):
#   A1: the whole -m gpu suite; kernel trace + memory-request counters of the default bench in its TIMED form (bench.py --traffic-pass: W + K
#       steps and nothing else of the engine) -> profiles/hbm_traffic.json (per kernel and per step) -> the bench line itself
#   A2: the same for configs[4]; bench lines of configs[1], configs[2]
#   B : one rank's compute at world sizes 2 / 4 / 8 on one GPU (replicated index: tools/emulate_rank.py; bucket-sharded: tools/emulate_shard.py)
#   C : full-size parity on THIS tree against the recorded reference dumps (sha256): 50 M reads error-free, 10 M reads with 2 % errors
# usage: tools/final_measure_r05.sh A1|A2|B|C
# (A1 and A2 each rewrite ONE config's entry of profiles/hbm_traffic.json on the box; run as separate gpurun calls, copy gpurun_out/r05_final/hbm_traffic.json
#  into profiles/ after A1 and before A2 -- or regenerate an entry from the merged counter files: tools/pmc_to_traffic.py gpurun_out/prof_r05_final_<cfg> <cfg> --builds=7)
set -u
PART=${1:-A1}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/r05_final
mkdir -p $OUT
cd $REPO
export TMPDIR=/tmp
W=2; K=5
if [ "$PART" = A1 ] || [ "$PART" = A2 ]; then
  if [ "$PART" = A1 ]; then
    timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; rc=$?
    tail -3 $OUT/pytest_gpu.log
    [ $rc -ne 0 ] && exit $rc
    CFGS="cfg4_50M_150bp"
  else
    CFGS="cfg5_10M_150bp_err2"
  fi
  for CFG in $CFGS; do
    timeout -k 10 900 bash tools/profile_cmd.sh r05_final_$CFG trx bench.py --config $CFG --traffic-pass --steps $K --warmup $W || exit 1
    python tools/pmc_to_traffic.py gpurun_out/prof_r05_final_$CFG $CFG --builds=$((W + K)) > $OUT/traffic_$CFG.json || exit 1
    cp gpurun_out/prof_r05_final_$CFG/summary.txt $OUT/rocprof_$CFG.txt
    timeout -k 10 600 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err || { tail -20 $OUT/bench_$CFG.err; exit 1; }
    echo "$CFG: $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$CFG.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['basis'], d['roofline_step'].get('traffic_amplification'), d.get('ms_end_to_end'), d.get('timed_edges_digest_equal_pairwise'))")"
  done
  cp profiles/hbm_traffic.json $OUT/hbm_traffic.json
  [ "$PART" = A1 ] && exit 0
  for CFG in cfg2_1M_150bp cfg3_5M_150bp; do
    timeout -k 10 600 python bench.py --config $CFG > $OUT/bench_$CFG.json 2> $OUT/bench_$CFG.err || { tail -20 $OUT/bench_$CFG.err; exit 1; }
    echo "$CFG: $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_$CFG.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d.get('ms_end_to_end'))")"
  done
elif [ "$PART" = B ]; then
  : > $OUT/emulated_rank_compute.jsonl
  : > $OUT/emulated_bucket_sharded.jsonl
  for N in 2 4 8; do
    timeout -k 10 300 python tools/emulate_rank.py $N 50000000 250000000 5 1 >> $OUT/emulated_rank_compute.jsonl 2> $OUT/emulate_$N.err || { tail -5 $OUT/emulate_$N.err; exit 1; }
    timeout -k 10 400 python tools/emulate_shard.py $N 50000000 250000000 3 >> $OUT/emulated_bucket_sharded.jsonl 2> $OUT/emulate_shard_$N.err || { tail -5 $OUT/emulate_shard_$N.err; exit 1; }
  done
  cut -c1-500 $OUT/emulated_rank_compute.jsonl
  cut -c1-700 $OUT/emulated_bucket_sharded.jsonl
else
  timeout -k 10 600 python tools/run_cfg4.py 50000000 250000000 2 source_side 0 a21745295ac286551ab5ce9fd8a1b9b2107f53cab6c329ed32e819deb3aed925 > $OUT/cfg4_50M_dump_vs_recorded_reference.log 2>&1 || { tail -5 $OUT/cfg4_50M_dump_vs_recorded_reference.log; exit 1; }
  tail -1 $OUT/cfg4_50M_dump_vs_recorded_reference.log | cut -c1-900
  timeout -k 10 600 python tools/cfg5_exact_dump_hash.py > $OUT/cfg5_10M_exact_path_vs_recorded_reference.json 2> $OUT/cfg5_hash.err || { tail -5 $OUT/cfg5_hash.err; exit 1; }
  cat $OUT/cfg5_10M_exact_path_vs_recorded_reference.json
fi
