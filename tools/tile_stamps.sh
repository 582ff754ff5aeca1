#!/bin/bash
# one-workgroup-per-tile conv kernel: where a workgroup's cycles go (setup / staging incl. barriers / k loops / epilogue),
# per layer shape of the forward that still runs on it; diagnostic build with in-kernel cycle stamps
out=${1:-gpurun_out/tile_stamps.txt}
root=$(cd "$(dirname "$0")/.." && pwd)
tools/build_variant.sh stamps -DRTPE_DIAG -DRTPE_CONV_STAMPS > /dev/null 2>&1 || exit 1
export RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_stamps.so
for th in "" 8; do
  echo "== RTPE_CONV_TH=$th" >> $out
  RTPE_CONV_TH=$th timeout -k 10 300 python tools/conv_probe.py 64,64,3,1,160,160,32,0 256,48,3,1,160,160,32,0 256,96,3,2,160,160,32,0 \
      64,64,3,2,320,320,32,0 96,192,3,2,80,80,32,0 192,384,3,2,40,40,32,0 48,17,1,1,320,320,32,0 >> $out 2>&1 || exit 1
done
cat $out
