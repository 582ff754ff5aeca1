#!/bin/bash
# fused 96-channel block (csrc/conv_block96.hip): time of a block with parts switched off in a diagnostic build
# (RTPE_BLOCK96_ABL bits: 1 k loops, 2 epilogue A, 4 output stores, 8 x-tile requests, 16 weight requests); batch 32 at 640x640
out=${1:-gpurun_out/b96_abl.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh diag -DRTPE_DIAG > /dev/null 2>&1 || exit 1
: > $out
for abl in ${ABLS:-0 1 2 4 8 16 24 17 25 6 7 31}; do
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_diag.so RTPE_AUTOTUNE=0 RTPE_BLOCK96_ABL=$abl timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_$abl.txt > /dev/null 2>&1 || exit 1
  echo "abl=$abl $(grep 'conv 96->96 k3s1+relu' /tmp/ops_$abl.txt | awk '{print $(NF-1)}') us" >> $out
done
cat $out
