"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid, block, lds) call count and mean/min us."""
import collections
import csv
import glob
import sys


def main(paths):
    for pat in paths:
        for path in sorted(glob.glob(pat, recursive=True)):
            agg = collections.OrderedDict()
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].replace("void rtpe::", "").split("(")[0][:44]
                key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"], r["LDS_Block_Size"],
                       r["VGPR_Count"])
                agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            print("==", path)
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                if sum(v) < 5:
                    continue
                print("%-44s grid %8s x%-4s wg %4s lds %6s vgpr %4s  n=%-4d mean %9.1f us  min %9.1f" % (
                    k + (len(v), sum(v) / len(v), min(v))))


if __name__ == "__main__":
    main(sys.argv[1:] or ["gpurun_out/**/*kernel_trace.csv"])
