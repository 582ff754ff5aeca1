#!/bin/bash
# bench.py by workgroups per XCD of the persistent kernels (RTPE_PERSIST_G: 32 = one per CU; 16 = two kernels of different
# forwards can run side by side on disjoint halves of the chip) and forwards in flight; lanes off inside the loop.
out=${1:-gpurun_out/gsweep}; mkdir -p $out
for cfg in ${CFGS:-"32 2" "16 2" "16 3" "16 4" "20 3" "12 3" "32 2"}; do
  set -- $cfg
  RTPE_PERSIST_G=$1 RTPE_STREAM_LANES=0 RTPE_FORWARDS_IN_FLIGHT=$2 RTPE_AUTOTUNE_CACHE=$PWD/$out/at_g$1.json timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > $out/b_$1_$2.json 2> $out/b_$1_$2.err
  echo "G=$1 in_flight=$2: $(python -c "import json,sys; d=json.loads(open('$out/b_$1_$2.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['forward_only_images_per_sec_per_gpu'])" 2>/dev/null)"
done
