#!/bin/bash
# same-box A/B of the 48-input stride-2 convs under bench.py (end to end, pipelined loop):
#   A: conv48s2 off (the streaming kernel's launch shapes), B: every such layer on conv48s2, one launch per layer
#   (RTPE_S2_GROUPS=0), C: the same with the sibling convs of a fuse layer as one launch (default)
out=${1:-gpurun_out/s2_ab.txt}
: > $out
run () {
  name=$1; shift
  for r in 1 2; do
    env "$@" RTPE_AUTOTUNE_CACHE=$PWD/gpurun_out/s2_ab_tune_$name.json timeout -k 10 300 python bench.py --no-cpu-baseline --steps 60 > gpurun_out/s2_ab_$name.json 2> gpurun_out/s2_ab_$name.err || exit 1
    echo "$name run $r: $(python -c "import json; d=json.loads(open('gpurun_out/s2_ab_$name.json').read().strip().splitlines()[-1]); print(d['value'], 'img/s', d['ms_per_step'], 'ms; forward only', d['config']['forward_only_images_per_sec_per_gpu'])")" >> $out
  done
}
run A RTPE_CONV48S2=0
run B RTPE_CONV48S2=1 RTPE_S2_GROUPS=0
run C RTPE_CONV48S2=1
cat $out
