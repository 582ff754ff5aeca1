export RTPE_AUTOTUNE_CACHE=$PWD/gpurun_out/inflight_autotune.json
for rep in 1 2; do for k in 1 2 3; do echo "== bench RTPE_FORWARDS_IN_FLIGHT=$k"; RTPE_FORWARDS_IN_FLIGHT=$k timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-130; done; done
for b in 1 4; do for k in 1 2; do echo "== bench --batch $b RTPE_FORWARDS_IN_FLIGHT=$k"; RTPE_FORWARDS_IN_FLIGHT=$k timeout -k 10 300 python bench.py --no-cpu-baseline --batch $b --steps 40 2>/dev/null | cut -c1-130; done; done
