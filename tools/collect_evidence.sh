#!/bin/bash
# after `gpurun -- 'bash tools/pmc_passes.sh r03; bash tools/final_evidence.sh r03'`: copy what is judged into profiles/
TAG=${1:-r03}
cp gpurun_out/$TAG/trace/trace_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
python tools/trace_summary.py --last-forwards 8 --csv profiles/${TAG}_timed_steps.csv gpurun_out/$TAG/trace/trace_kernel_trace.csv > /dev/null
cp gpurun_out/$TAG/bench_traced.json profiles/${TAG}_bench_under_rocprof.json
cp gpurun_out/$TAG/bench_unprofiled.json profiles/${TAG}_bench_before_profiling.json
cp gpurun_out/${TAG}f/bench_default.json profiles/${TAG}_bench.json
cp gpurun_out/${TAG}f/bench_default_2.json profiles/${TAG}_bench_second_run.json
cp gpurun_out/${TAG}f/bench_config1.json profiles/${TAG}_bench_config1.json
cp gpurun_out/${TAG}f/bench_config4.json profiles/${TAG}_bench_config4.json
cp gpurun_out/${TAG}f/bench_list.json profiles/${TAG}_bench_list_mode.json
tail -1 gpurun_out/${TAG}f/bench_2rank_gloo.json > profiles/${TAG}_bench_2rank_gloo_rehearsal.json
cp gpurun_out/${TAG}f/forward_ops.txt profiles/${TAG}_forward_ops.txt
cp gpurun_out/${TAG}f/latency.log profiles/${TAG}_latency_probe.txt
python tools/pmc_summary.py gpurun_out/$TAG profiles/${TAG}_pmc_summary.json > /tmp/pmcsum.log 2>&1; tail -1 /tmp/pmcsum.log
python - $TAG <<'PY'
import json, csv, sys
tag = sys.argv[1]
d = json.load(open('profiles/%s_bench.json' % tag)); r = d['roofline_secondary']; t = d['roofline']   # round 5: roofline = the time-dominant family
print("default", d['value'], d['ms_per_step'], 'frac', r['frac'], 'launch_us', r['launch_us'], 'achieved', r['achieved'], 'phys', r['physical']['hbm_gbs_measured_traffic'], r['physical']['hbm_frac_measured_traffic'], r['physical']['mfma_frac'], r['mfma_tflops'], 'traffic', r['traffic'])
print('build_mode', d.get('build_mode')); print('time-dominant', t['frac'], 'launch_us', t['launch_us'], 'traffic', t['traffic'], t['achieved'], t['ms_per_forward'], t['share_of_forward'], 'cpu', d['cpu_baseline']['value'], 'fwd-only', d['config']['forward_only_images_per_sec_per_gpu'])
for f in ('bench_under_rocprof', 'bench_before_profiling', 'bench_second_run', 'bench_config1', 'bench_config4', 'bench_list_mode', 'bench_2rank_gloo_rehearsal'):
    e = json.load(open('profiles/%s_%s.json' % (tag, f))); print(f, e['value'], e['ms_per_step'], (e.get('roofline') or {}).get('launch_us'), e['config'].get('pipelined_images_per_sec'))
rows = list(csv.DictReader(open('profiles/%s_timed_steps.csv' % tag)))
tot = sum(float(x['total_us']) for x in rows)
fam = {}
for x in rows:
    k = x['kernel']
    f = 'tile' if k.startswith('conv_mfma') else ('stream' if k.startswith('conv_stream') else ('direct' if k.startswith('conv1x1_direct') else ('pair' if k.startswith('conv1x1_pair') else ('block' if k.startswith('conv_block') else ('decode' if any(y in k for y in ('topk', 'refine', 'plane', 'adjust')) else k)))))
    fam[f] = fam.get(f, 0) + float(x['total_us'])
print({f: round(100 * v / tot, 1) for f, v in sorted(fam.items(), key=lambda kv: -kv[1])[:8]})
for x in rows:
    if x['kernel'].startswith(('conv_block', 'conv1x1_pair')): print(x['kernel'], x['calls'], x['mean_us'], x['median_us'], x['min_us'])
PY
grep -A3 "LANES=2$" profiles/${TAG}_latency_probe.txt | head -4; tail -2 profiles/${TAG}_forward_ops.txt
