#!/bin/bash
# one interleaved A/B pass of the forward per-op table (base = librtpe_base.so) + the stream conv rows
out=$1; mkdir -p $out
RTPE_LIBRARY=$PWD/realtime-pose-estimation_amd/librtpe_base.so python tools/forward_profile.py 32 640 $out/base.txt > $out/base.log 2>&1 || exit 1
python tools/forward_profile.py 32 640 $out/new.txt > $out/new.log 2>&1 || exit 1
for f in base new; do echo "== $f"; grep -E "conv (96->96|192->192|384->384|48->96|48->48|48->192) k3|forward total" $out/$f.txt; done
