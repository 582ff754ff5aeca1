#!/bin/bash
# fused 96-channel block: in-kernel cycle stamps per phase.  The stamps build has NO -DRTPE_DIAG (the ablation switches
# put a select behind every MFMA); ABLS="0 1 16" adds a diagnostic build with those.    bash tools/b96_stamps.sh <out file>
out=${1:-gpurun_out/b96_stamps.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh stamps -DRTPE_CONV_STAMPS $STAMP_DEFS > /dev/null 2>&1 || exit 1
export RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_stamps.so
: > $out
timeout -k 10 300 python tools/conv_probe.py block96,80,80,32 96,96,3,1,80,80,32,1 >> $out 2>&1 || exit 1
for pp in 1 2; do
  echo "== RTPE_PROBE_PLANE=$pp (plane-major views: 1 output, 2 both)" >> $out
  RTPE_PROBE_PLANE=$pp timeout -k 10 300 python tools/conv_probe.py block96,80,80,32 >> $out 2>&1 || exit 1
done
if [ -n "$ABLS" ]; then
  tools/build_variant.sh stampsd -DRTPE_DIAG -DRTPE_CONV_STAMPS > /dev/null 2>&1 || exit 1
  export RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_stampsd.so
  for abl in $ABLS; do
    echo "== RTPE_BLOCK96_ABL=$abl" >> $out
    RTPE_BLOCK96_ABL=$abl timeout -k 10 300 python tools/conv_probe.py block96,80,80,32 >> $out 2>&1 || exit 1
  done
fi
cat $out
