#!/bin/bash
# A/B of the streaming conv kernels on the C >= 96 layer classes (batch 32): kernel durations from rocprofv3's kernel
# trace of tools/conv_probe.py with option stream_pc = 0 (conv_stream_kernel) and 2 (conv_stream_pc_kernel; with the
# diagnostic build librtpe_diag.so also per RTPE_PC_FLAGS value), then the in-kernel stamps of librtpe_stamps.so.
#   tools/pc_ab.sh <out_dir> ["flag values"]
out=${1:-gpurun_out/pc_ab}; flags=${2:-0}; mkdir -p $out
cases=${PC_CASES:-"96,96,3,1,80,80,32,1 96,96,3,1,80,80,32,0 192,192,3,1,40,40,32,1 192,192,3,1,40,40,32,0"}
root=$GRAFT_REPO_ROOT
lib=$root/realtime-pose-estimation_amd/librtpe_diag.so
[ -f $lib ] || lib=$root/realtime-pose-estimation_amd/librtpe_hip.so
cd /tmp && export TMPDIR=/tmp
export RTPE_LIBRARY=$lib
for f in ${V1_FLAGS:-0}; do
  RTPE_PC_FLAGS=$f RTPE_PROBE_OPTS=stream_pc=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof_v1_$f -o t -- python3 $root/tools/conv_probe.py $cases > $root/$out/probe_v1_$f.log 2>&1 || exit 1
done
for f in $flags; do
  RTPE_PC_FLAGS=$f RTPE_PROBE_OPTS=stream_pc=2 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/prof_pc$f -o t -- python3 $root/tools/conv_probe.py $cases > $root/$out/probe_pc$f.log 2>&1 || exit 1
done
cd $root
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for d in sorted(glob.glob(out + "/prof_*")):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        print("no trace in", d); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "conv_stream" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    print("== " + d.split("/")[-1])
    for i in range(0, len(rows), 3):
        grp = rows[i:i + 3]
        t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
        print("  %-46s %s us (min %.1f)" % (grp[0]["Kernel_Name"][:46], " ".join("%.1f" % x for x in t), min(t)))
PY
if [ -f realtime-pose-estimation_amd/librtpe_stamps.so ]; then
  export RTPE_LIBRARY=$root/realtime-pose-estimation_amd/librtpe_stamps.so
  echo "== stamps, conv_stream_kernel"
  RTPE_PROBE_OPTS=stream_pc=0 timeout -k 10 300 python3 tools/conv_probe.py $cases 2>&1 | grep -E "^stream"
  for f in $flags; do
    echo "== stamps, conv_stream_pc_kernel, flags $f"
    RTPE_PC_FLAGS=$f RTPE_PROBE_OPTS=stream_pc=2 timeout -k 10 300 python3 tools/conv_probe.py $cases 2>&1 | grep -E "^stream"
  done
fi
