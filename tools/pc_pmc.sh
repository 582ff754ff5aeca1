#!/bin/bash
# Memory-path counters of the streaming conv kernels on single layers (tools/conv_probe.py), one rocprofv3 --pmc pass per
# counter group (kernel trace only beside them).   tools/pc_pmc.sh <out_dir> [stream_pc option]
out=${1:-gpurun_out/pc_pmc}; pc=${2:-0}; mkdir -p $out
cases=${PC_CASES:-"96,96,3,1,80,80,32,1 192,192,3,1,40,40,32,1 384,384,3,1,20,20,32,1"}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass () {
  name=$1; shift
  RTPE_PROBE_OPTS=stream_pc=$pc timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/$out/$name -o p -- python3 $root/tools/conv_probe.py $cases > $root/$out/$name.log 2>&1 || echo "pass $name failed"
}
pass tcp1 TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_WRITE_REQ_LATENCY
pass tcp2 TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ
pass tcc1 TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL
pass tcc2 TCC_REQ TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_TAG_STALL
pass ta1 TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS
pass ta2 TA_BUFFER_READ_LDS_WAVEFRONTS TA_BUFFER_WRITE_WAVEFRONTS TA_BUFFER_READ_WAVEFRONTS TA_ADDR_STALLED_BY_TD_CYCLES
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
cd $root
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if "conv_stream" not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:40], r["Dispatch_Id"])
        acc.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("== " + d.split("/")[-2])
    for (k, disp), v in acc.items():
        print("  %-40s #%s  %s" % (k, disp, "  ".join("%s=%.4g" % kv for kv in v.items())))
PY
