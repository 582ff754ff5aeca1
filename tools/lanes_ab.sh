# forward wall time with the first / the wider parallel regions (tuned shapes), interleaved
export RTPE_AUTOTUNE_CACHE=$PWD/gpurun_out/lanes_autotune2.json
for rep in 1 2 3; do for cfg in 0 1; do echo "== RTPE_WIDE_REGIONS=$cfg"; RTPE_WIDE_REGIONS=$cfg timeout -k 10 300 python tools/host_overhead.py 2>/dev/null | grep "^batch"; done; done
