#!/bin/bash
# all-couts 96 -> 96 3x3 kernel (csrc/conv96.hip): time of the layer with parts switched off (RTPE_STREAM_ABL bits: 1 k loops,
# 2 output stores, 4 halo DMA of the next tile) in a diagnostic build; batch 32 at 640x640, un-tuned launches (option conv96 = 2)
out=${1:-gpurun_out/conv96_abl.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh diag -DRTPE_DIAG > /dev/null 2>&1 || exit 1
: > $out
for abl in ${ABLS:-0 1 2 4 3 6 7}; do
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_diag.so RTPE_AUTOTUNE=0 RTPE_CONV96=2 RTPE_STREAM_ABL=$abl timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_$abl.txt > /dev/null 2>&1 || exit 1
  echo "abl=$abl $(grep 'conv 96->96 k3s1' /tmp/ops_$abl.txt | awk '{printf "%s ", $(NF-1)}') us (+res+relu / +relu)" >> $out
done
cat $out
