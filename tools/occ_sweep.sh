#!/bin/bash
# usage: occ_test.sh "<extra flags 1>" "<extra flags 2>" ...   (runs on the GPU box)
for f in "$@"; do
  make -C alga_amd/csrc -B -j16 EXTRA="$f" > gpurun_out/build_occ.log 2>&1 || { tail -5 gpurun_out/build_occ.log; exit 1; }
  echo "== EXTRA=$f"
  timeout -k 10 200 python tools/probe_compare.py 16000000 80000000 3 cluster 2>/dev/null | grep -v "^{" | tail -2
done
