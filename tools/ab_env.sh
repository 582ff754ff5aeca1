#!/bin/bash
# Same-box A/B of one environment switch of the library (boxes of the pool differ by up to 10 %, so two
# builds or two settings are only comparable inside one gpurun call):
#   gpurun -- 'bash tools/ab_env.sh RTPE_PLANE_MAJOR 0 1'
# runs the forward profile (batch 32 at 640x640) twice per value and prints the forward totals; the per-op
# tables stay in gpurun_out/ab_<value>_<run>.txt.
var=${1:?variable}; shift
for r in 1 2; do for v in "$@"; do
  env "$var=$v" timeout -k 10 300 python tools/forward_profile.py 32 640 "gpurun_out/ab_${v}_$r.txt" > /dev/null 2>&1 || exit 1
  echo "$var=$v run $r: $(tail -1 "gpurun_out/ab_${v}_$r.txt")"
done; done
