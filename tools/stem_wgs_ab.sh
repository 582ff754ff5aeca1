mkdir -p gpurun_out/r05l
for w in 1 2 4 8 1; do
  RTPE_STEM_WGS=$w python bench.py --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('wgs=$w', d['value'], d['ms_per_step'])" >> gpurun_out/r05l/stemwgs.txt
done
cat gpurun_out/r05l/stemwgs.txt
