#!/bin/bash
# write-through (shipping: store16_wt, "sc0 sc1") against plain 16-byte stores (build with -DRTPE_WT_STORES=0): the default bench,
# alternating, three rounds
mkdir -p gpurun_out/wt_ab
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh plain -DRTPE_WT_STORES=0 > /dev/null 2>&1 || exit 1
for rep in 1 2 3; do
  a=$(python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  b=$(RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_plain.so python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])")
  echo "rep $rep: write-through $a | plain $b" | tee -a gpurun_out/wt_ab/bench.txt
done
