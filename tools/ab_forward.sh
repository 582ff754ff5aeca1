#!/bin/bash
# A/B of two builds of librtpe_hip.so on ONE box: per-op tables of the forward, interleaved (base, new, base, new).
#   tools/ab_forward.sh <out_dir> [batch] [size]      base = realtime-pose-estimation_amd/librtpe_base.so
out=$1; B=${2:-32}; S=${3:-640}
mkdir -p $out
for r in 1 2; do
  RTPE_LIBRARY=$PWD/realtime-pose-estimation_amd/librtpe_base.so python tools/forward_profile.py $B $S $out/base_$r.txt > $out/base_$r.log 2>&1 || exit 1
  python tools/forward_profile.py $B $S $out/new_$r.txt > $out/new_$r.log 2>&1 || exit 1
done
tail -1 $out/base_1.txt $out/new_1.txt $out/base_2.txt $out/new_2.txt
