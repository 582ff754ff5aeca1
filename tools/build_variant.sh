#!/bin/bash
# A second build of the extension with extra defines, beside the shipping one:
#   tools/build_variant.sh diag -DRTPE_DIAG     ->  realtime-pose-estimation_amd/librtpe_diag.so   (use with RTPE_LIBRARY=...)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
pkg=$root/realtime-pose-estimation_amd
mkdir -p $pkg/build_$name
pids=()
for src in $pkg/csrc/*.hip $pkg/csrc/*.cpp; do
  obj=$pkg/build_$name/$(basename $src).o
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -I$root/include -I$pkg/csrc "$@" $([[ $src == *.hip ]] && echo "-x hip") -c $src -o $obj &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $pkg/librtpe_$name.so $pkg/build_$name/*.o
echo built $pkg/librtpe_$name.so
