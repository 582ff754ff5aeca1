#!/bin/bash
# fused 96-channel block (conv_block96.hip) against the two streaming launches: parity tests, then the bench's per-op table
# with the option on and off (separate processes: each autotunes in its own mode).    bash tools/b96_ab.sh <out_dir>
out=${1:-gpurun_out/b96}
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "block96 or fused_96" > $out/t.log 2>&1 || { tail -30 $out/t.log; exit 1; }
tail -2 $out/t.log
for on in 1 0; do
  RTPE_BLOCK96=$on RTPE_AUTOTUNE_CACHE=$PWD/$out/tune_$on.json timeout -k 10 600 python bench.py --no-cpu-baseline --steps 40 --dump-ops $out/ops_$on.txt > $out/bench_$on.json 2> $out/bench_$on.err || { tail -5 $out/bench_$on.err; exit 1; }
  echo "block96=$on: $(python -c "import json;d=json.load(open('$out/bench_$on.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_us'], d['config']['forward_only_images_per_sec_per_gpu'])")"
  grep -E "96->96|block 96" $out/ops_$on.txt
  tail -1 $out/ops_$on.txt
done
