#!/bin/bash
# experiment: time of the probe kernel with later phases cut off (results are wrong for ABLATE != 0; timing only)
set -e
for a in "$@"; do
  make -C alga_amd/csrc -B -j16 EXTRA=-DABLATE=$a > gpurun_out/build_ab$a.log 2>&1
  echo "ablate=$a"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['phases_ms']['probe'], d['counters']['records'])"
done
