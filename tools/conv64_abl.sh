#!/bin/bash
# persistent 64 -> 64 3x3 kernel (csrc/conv64.hip): time of the layer with parts switched off (RTPE_STREAM_ABL bits: 1 k loops,
# 2 output stores, 4 halo DMA of the next tile) in a diagnostic build; batch 32 at 640x640, un-tuned launches
out=${1:-gpurun_out/conv64_abl.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh diag -DRTPE_DIAG > /dev/null 2>&1 || exit 1
: > $out
for abl in ${ABLS:-0 1 2 4 3 6 7}; do
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_diag.so RTPE_AUTOTUNE=0 RTPE_STREAM_ABL=$abl timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_$abl.txt > /dev/null 2>&1 || exit 1
  echo "abl=$abl $(grep 'conv 64->64 k3s1' /tmp/ops_$abl.txt | awk '{print $(NF-1)}') us" >> $out
done
cat $out
