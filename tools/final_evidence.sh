#!/bin/bash
# Round-end evidence on the GPU box: default bench (with cpu_baseline), the other configurations, forward op table,
# list mode, 2-rank gloo rehearsal, batch-1 latency.     bash tools/final_evidence.sh r03
set -e
TAG=${1:-r03}
OUT=gpurun_out/${TAG}f
mkdir -p $OUT
export RTPE_AUTOTUNE_CACHE=$PWD/$OUT/autotune.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default: $(cut -c1-160 $OUT/bench_default.json)"
python bench.py --no-cpu-baseline > $OUT/bench_default_2.json 2> $OUT/bench_default_2.err
echo "default again: $(cut -c1-160 $OUT/bench_default_2.json)"
python bench.py --config 1 --no-cpu-baseline > $OUT/bench_config1.json 2> $OUT/bench_config1.err
echo "config 1: $(cut -c1-160 $OUT/bench_config1.json)"
python bench.py --config 4 --no-cpu-baseline > $OUT/bench_config4.json 2> $OUT/bench_config4.err
echo "config 4: $(cut -c1-160 $OUT/bench_config4.json)"
python tools/forward_profile.py 32 640 $OUT/forward_ops.txt > $OUT/forward_profile.log 2>&1 || tail -3 $OUT/forward_profile.log
tail -2 $OUT/forward_ops.txt
python bench.py --no-cpu-baseline --list tests/golden/coco_minival2017_100.txt --steps 5 --warmup 1 > $OUT/bench_list.json 2> $OUT/bench_list.err
echo "list: $(cut -c1-160 $OUT/bench_list.json)"
# bench.py starts its own two ranks (no WORLD_SIZE in the environment): the form the driver uses for --gpus N
RTPE_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2rank: $(tail -1 $OUT/bench_2rank_gloo.json | cut -c1-160)"
for lanes in 0 2; do echo "== RTPE_LANES=$lanes"; RTPE_LANES=$lanes python tools/latency_probe.py 2>/dev/null | grep -v amdgpu; done > $OUT/latency.log
echo "== RTPE_LANES=2 RTPE_TILE_DMA=0" >> $OUT/latency.log
RTPE_TILE_DMA=0 python tools/latency_probe.py 2>/dev/null | grep -v amdgpu >> $OUT/latency.log
python tools/host_overhead.py 2>/dev/null | grep "^batch" >> $OUT/latency.log
cat $OUT/latency.log
