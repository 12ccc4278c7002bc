#!/bin/bash
# experiment: requested occupancy of the probe kernel (run on the GPU box; rebuilds the library per setting)
set -e
for o in 4 5 6; do
  make -C alga_amd/csrc -B -j16 EXTRA=-DPROBE_OCC=$o > gpurun_out/build_$o.log 2>&1
  echo "occ=$o"
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['phases_ms'])"
done
