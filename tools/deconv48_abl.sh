#!/bin/bash
# persistent transposed-conv kernel (csrc/deconv48.hip): time of the layer with parts switched off (RTPE_STREAM_ABL bits: 1 k
# loops, 2 output stores, 4 halo DMA of the next tile) in a diagnostic build; batch 32 at 640x640.
#   DEFS="-DRTPE_D48_ORDER=1" tools/deconv48_abl.sh out.txt      (extra defines of the diagnostic build)
out=${1:-gpurun_out/deconv48_abl.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh diag -DRTPE_DIAG $DEFS > /dev/null 2>&1 || exit 1
echo "# DEFS=$DEFS" > $out
for abl in ${ABLS:-0 1 2 4 3 6 7}; do
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_diag.so RTPE_AUTOTUNE=0 RTPE_STREAM_ABL=$abl timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_$abl.txt > /dev/null 2>&1 || exit 1
  echo "abl=$abl $(grep 'deconv 82->48' /tmp/ops_$abl.txt | awk '{print $(NF-1)}') us" >> $out
done
cat $out
