#!/bin/bash
# SQ / LDS counters of single layers run by tools/conv_probe.py, one rocprofv3 --pmc pass per counter group (kernel trace
# only beside them).     tools/sq_pmc.sh <out_dir> "<conv_probe cases>" <kernel name substring>
out=${1:-gpurun_out/sq_pmc}; cases=${2:-"block96,80,80,32 96,96,3,1,80,80,32,1"}; kern=${3:-conv_}
mkdir -p $out
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass () {
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/$out/$name -o p -- python3 $root/tools/conv_probe.py $cases > $root/$out/$name.log 2>&1 || echo "pass $name failed"
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
pass sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE
pass sq3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE
pass sq4 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
cd $root
python3 - $out "$kern" <<'PY'
import csv, glob, sys, collections
out, kern = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(out + "/*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if kern not in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:44], r["Dispatch_Id"])
        acc.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    print("== " + d.split("/")[-2])
    for (k, disp), v in acc.items():
        print("  %-44s #%s  %s" % (k, disp, "  ".join("%s=%.4g" % kv for kv in v.items())))
PY
