"""Summarise a rocprofv3 kernel_trace.csv: per kernel call count and mean / min / total us.

    python tools/trace_summary.py [--last-forwards K] [--csv out.csv] trace.csv ...

--last-forwards K keeps only the dispatches from the start of the K-th last forward pass on
(one stem_kernel / stem_fused_kernel launch per forward), i.e. the timed steps of bench.py without warm-up and
autotuning passes.
"""
import argparse
import collections
import csv
import glob


def short(name):
    name = name.replace("void rtpe::", "").replace("rtpe::", "")
    if name.startswith("_ZN4rtpe"):
        import re
        m = re.match(r"_ZN4rtpe(\d+)([A-Za-z_0-9]+)", name)
        if m:
            base = m.group(2)[:int(m.group(1))]
            args = re.findall(r"Li(\d+)E", name)
            name = base + ("<" + ",".join(args) + ">" if args else "")
    return name.split("(")[0][:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--last-forwards", type=int, default=0)
    ap.add_argument("--drop-last", type=int, default=0, help="ignore the last N forward passes (bench.py's forward-only tail)")
    ap.add_argument("--csv", default=None)
    ap.add_argument("paths", nargs="*", default=["gpurun_out/**/*kernel_trace.csv"])
    a = ap.parse_args()
    for pat in a.paths:
        for path in sorted(glob.glob(pat, recursive=True)):
            rows = list(csv.DictReader(open(path)))
            rows.sort(key=lambda r: int(r["Start_Timestamp"]))
            if a.last_forwards:
                stems = [int(r["Start_Timestamp"]) for r in rows if "stem_kernel" in r["Kernel_Name"] or "stem_fused_kernel" in r["Kernel_Name"]]
                if len(stems) >= a.last_forwards:
                    t0 = stems[-a.last_forwards]
                    t1 = stems[-a.drop_last] if a.drop_last else None
                    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0 and
                            (t1 is None or int(r["Start_Timestamp"]) < t1)]
            agg = collections.OrderedDict()
            for r in rows:
                key = (short(r["Kernel_Name"]), r["Workgroup_Size_X"], r["VGPR_Count"])
                agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            total = sum(sum(v) for v in agg.values())
            span = (max(int(r["End_Timestamp"]) for r in rows) - min(int(r["Start_Timestamp"]) for r in rows)) / 1e3
            print("== %s: %d dispatches, kernel time %.1f us over a span of %.1f us" % (path, len(rows), total, span))
            out = [("kernel", "workgroup", "vgpr", "calls", "total_us", "mean_us", "median_us", "min_us", "pct")]
            for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
                out.append(k + (len(v), round(sum(v), 1), round(sum(v) / len(v), 2), round(sorted(v)[len(v) // 2], 2), round(min(v), 2),
                                round(100 * sum(v) / total, 2)))
            for o in out[:40]:
                print("%-60s wg %5s vgpr %5s n=%-5s total %11s mean %9s median %9s min %9s %6s" % o)
            if a.csv:
                with open(a.csv, "w", newline="") as f:
                    wr = csv.writer(f)              # kernel names contain commas (template arguments): quoted
                    for o in out:
                        wr.writerow(o)


if __name__ == "__main__":
    main()
