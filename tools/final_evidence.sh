#!/bin/bash
# Round-end evidence on the GPU box: default bench (with cpu_baseline), forward op table, list mode, 2-rank gloo rehearsal.
set -e
OUT=gpurun_out/r02f
mkdir -p $OUT
export RTPE_AUTOTUNE_CACHE=$PWD/$OUT/autotune.json
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default: $(cut -c1-160 $OUT/bench_default.json)"
python tools/forward_profile.py 32 640 $OUT/forward_ops.txt > $OUT/forward_profile.log 2>&1 || tail -3 $OUT/forward_profile.log
python bench.py --no-cpu-baseline --list tests/golden/coco_minival2017_100.txt --steps 5 --warmup 1 > $OUT/bench_list.json 2> $OUT/bench_list.err
echo "list: $(cut -c1-160 $OUT/bench_list.json)"
RTPE_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "2rank: $(tail -1 $OUT/bench_2rank_gloo.json | cut -c1-160)"
