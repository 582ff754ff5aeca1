#!/bin/bash
# conv48s2.hip: per-op times of the 48-input stride-2 convs by the number of halo tiles in flight (RTPE_C48S2_DEPTH at build time)
out=${1:-gpurun_out/c48s2_depth.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
: > $out
for d in ${DEPTHS:-1 2 3}; do
  tools/build_variant.sh d$d -DRTPE_C48S2_DEPTH=$d > /dev/null 2>&1 || exit 1
  RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_d$d.so RTPE_AUTOTUNE=0 timeout -k 10 200 python tools/forward_profile.py 32 640 /tmp/ops_d$d.txt > /dev/null 2>&1 || exit 1
  echo "== depth $d" >> $out
  grep "CONV48S2" /tmp/ops_d$d.txt | cut -c1-140 >> $out
done
cat $out
