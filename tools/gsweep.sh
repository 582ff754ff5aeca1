#!/bin/bash
mkdir -p gpurun_out/r4m
for cfg in "32 0 2" "24 0 2" "16 1 2" "24 1 2" "16 1 1" "24 1 1" "32 1 1" "20 1 2" "16 1 3"; do
  set -- $cfg
  RTPE_PERSIST_G=$1 RTPE_STREAM_LANES=$2 RTPE_FORWARDS_IN_FLIGHT=$3 RTPE_AUTOTUNE_CACHE=/tmp/at_g$1.json timeout -k 10 240 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/r4m/b_$1_$2_$3.json 2> gpurun_out/r4m/b_$1_$2_$3.err
  echo "G=$1 lanes_in_stream=$2 in_flight=$3: $(python -c "import json,sys; d=json.loads(open('gpurun_out/r4m/b_$1_$2_$3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
done
