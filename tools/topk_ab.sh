#!/bin/bash
# tile pass of the fused top-k (topk_tile_kernel) under the bench: duration (kernel trace) and HBM fetch (PMC pass) of
# two builds on one box.   tools/topk_ab.sh <out_dir>     base = realtime-pose-estimation_amd/librtpe_base.so
out=$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export RTPE_AUTOTUNE_CACHE=$PWD/$out/autotune.json
python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/bench_plain.json 2> $out/bench_plain.err || exit 1
for v in base new; do
  if [ $v = base ]; then export RTPE_LIBRARY=$PWD/realtime-pose-estimation_amd/librtpe_base.so; else unset RTPE_LIBRARY; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -o t -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/bench_$v.json 2> $out/bench_$v.err || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_$v -o p -- python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $out/bench_pmc_$v.json 2> $out/bench_pmc_$v.err || exit 1
  echo "== $v: $(cut -c1-100 $out/bench_$v.json)"
  grep -h "topk_tile_kernel\|topk_merge\|plane_argmax\|refine_scan" $out/trace_$v/*/*kernel_stats.csv | cut -c1-200
  python3 - $out/pmc_$v <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot = {}
for r in rows:
    if "topk_tile_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        tot.setdefault(r["Dispatch_Id"], 0.0)
        tot[r["Dispatch_Id"]] += float(r["Counter_Value"])
v = sorted(tot.values())
if v:
    print("   HBM fetch per launch of topk_tile_kernel (FETCH_SIZE x 1024 x 2, the gfx950 correction of tools/pmc_summary.py): median %.1f MB over %d launches" % (v[len(v)//2] * 2048 / 1e6, len(v)))
PY
done
