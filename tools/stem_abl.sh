#!/bin/bash
# fused stem kernel: time of the kernel with parts of it switched off (RTPE_STEM_ABL bits: 1 conv1, 2 conv2's k loop,
# 4 patch prefetch, 8 stores) in a diagnostic build; batch 32 at 640x640
out=${1:-gpurun_out/stem_abl.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh diag -DRTPE_DIAG > /dev/null 2>&1 || exit 1
: > $out
for abl in ${ABLS:-0 1 2 4 8 3 7 15}; do
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_diag.so RTPE_STEM_ABL=$abl timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_$abl.txt > /dev/null 2>&1 || exit 1
  echo "abl=$abl $(grep -i '^stem' /tmp/ops_$abl.txt | awk '{print $(NF-1)}') us" >> $out
done
cat $out
