"""Per-kernel summary of the rocprofv3 --pmc passes of tools/pmc_passes.sh.

    python tools/pmc_summary.py gpurun_out/r02 profiles/r02_pmc_summary.json

For every kernel (and, for the fused BasicBlock kernel, per launch class = grid of the 160x160 / 320x320 maps) over the
TIMED forwards of the bench run (the last forwards of the process: warm-up and autotuning passes are dropped):
  * HBM bytes per launch = FETCH_SIZE x 2 (gfx950 tallies the 128-byte requests of wide streaming reads at 64 bytes:
    /opt/skills/guides/MI355X_MICROARCH.md, section HBM) x 1024 (rocprofv3 reports KiB) + WRITE_SIZE x 1024;
  * MFMA utilisation from counters = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 256 CUs x 4 SIMDs), kernel cycles =
    GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs); cross-check: SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 = FLOPs issued;
  * LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (conflict cycles per LDS-array cycle), SQ_LDS_IDX_ACTIVE per kernel cycle
    and CU; SQ_VALU_MFMA_COEXEC_CYCLES / SQ_VALU_MFMA_BUSY_CYCLES; SQ_WAIT_ANY / SQ_WAVE_CYCLES.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def csrc_sha16():
    """the same hash bench.py computes over csrc/ and the header"""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "realtime-pose-estimation_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(os.listdir(d)) + ["../../include/rtpe_hip.h"]:
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    m = re.match(r"_ZN4rtpe(\d+)([A-Za-z_0-9]+)", name)
    if m:
        base = m.group(2)[:int(m.group(1))]
        args = re.findall(r"Li(\d+)E", name)
        return base + ("<" + ",".join(args) + ">" if args else "")
    name = name.replace("void rtpe::", "").replace("rtpe::", "")
    return name.split("(")[0][:80]


def load(path, keep_forwards=6, drop_last=3, raw_out=None):
    """{(kernel, grid): {counter: [values per dispatch]}} over the timed forwards"""
    rows = list(csv.DictReader(open(path)))
    per_disp = collections.OrderedDict()
    for r in rows:
        key = int(r["Dispatch_Id"])
        d = per_disp.setdefault(key, {"name": short(r["Kernel_Name"]), "grid": int(r["Grid_Size"]),
                                      "wg": int(r["Workgroup_Size"]), "t0": int(r["Start_Timestamp"]),
                                      "t1": int(r["End_Timestamp"]), "c": {}})
        d["c"][r["Counter_Name"]] = d["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp = sorted(per_disp.values(), key=lambda d: d["t0"])
    stems = [i for i, d in enumerate(disp) if d["name"].startswith(("stem_kernel", "stem_fused_kernel"))]
    if len(stems) >= keep_forwards + drop_last:
        a = stems[-(keep_forwards + drop_last)]
        b = stems[-drop_last] if drop_last else len(disp)
        disp = disp[a:b]
    out = collections.OrderedDict()
    nth = 0
    for d in disp:
        if d["name"].startswith(("stem_kernel", "stem_fused_kernel")):
            nth = 0
        cls = d["grid"]
        if d["name"].startswith("conv_block"):
            # the fused BasicBlock kernel launches one workgroup per CU whatever the map size: the launch classes
            # are told apart by their place in the forward (32 blocks at 160x160, then 4 at 320x320 for 640x640 inputs)
            cls = "160x160" if nth < 32 else "320x320"
            nth += 1
        d["cls"] = cls
    if raw_out:
        with open(raw_out, "w") as f:
            f.write("kernel,class,grid,workgroup,start_ns,end_ns,counter,value\n")
            for d in disp:
                for k, v in d["c"].items():
                    f.write("%s,%s,%d,%d,%d,%d,%s,%.1f\n" % (d["name"].replace(",", ";"), d["cls"], d["grid"], d["wg"], d["t0"], d["t1"], k, v))
    for d in disp:
        o = out.setdefault((d["name"], d["cls"]), collections.defaultdict(list))
        for k, v in d["c"].items():
            o[k].append(v)
        o["_us"].append((d["t1"] - d["t0"]) / 1e3)
        o["_grid"] = [d["grid"]]
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    merged = collections.OrderedDict()
    raw_dir = os.path.join(os.path.dirname(dst), os.path.basename(dst).replace("_pmc_summary.json", "_pmc_raw"))
    os.makedirs(raw_dir, exist_ok=True)
    for path in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        pass_name = os.path.relpath(path, src).split(os.sep)[0]
        for key, cs in load(path, raw_out=os.path.join(raw_dir, pass_name + "_timed_forwards.csv")).items():
            m = merged.setdefault(key, {})
            for k, v in cs.items():
                if k == "_us":
                    m.setdefault("_us_profiled", []).extend(v)
                else:
                    m[k] = v
    mean = lambda v: sum(v) / len(v) if v else None
    kernels = {}
    print("%-44s %9s %6s %9s %10s %10s %8s %8s %8s %8s" % ("kernel [grid]", "launches", "us*", "MB/launch", "fetchx2 MB", "write MB",
                                                          "mfma%", "ldsconf", "coexec", "waitany"))
    for (name, grid), c in sorted(merged.items(), key=lambda kv: -sum(kv[1].get("_us_profiled", [0]))):
        n = len(c.get("FETCH_SIZE", c.get("_us_profiled", [])))
        e = {"grid": grid, "launches_per_pass": n}
        if isinstance(grid, str):
            e["grid"] = int(c["_grid"][0])               # the launch's real grid size (threads), as in the raw rows
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            e["fetch_bytes"] = mean(c["FETCH_SIZE"]) * 1024 * 2
            e["write_bytes"] = mean(c["WRITE_SIZE"]) * 1024
            e["hbm_bytes_per_launch"] = e["fetch_bytes"] + e["write_bytes"]
        if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            cyc = mean(c["GRBM_GUI_ACTIVE"]) / 8.0
            e["kernel_cycles"] = cyc
            e["mfma_util"] = mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * 256 * 4)
            if "SQ_INSTS_VALU_MFMA_MOPS_F16" in c:
                e["mfma_flops_issued"] = mean(c["SQ_INSTS_VALU_MFMA_MOPS_F16"]) * 512
            if "SQ_INSTS_MFMA" in c:
                e["mfma_instructions"] = mean(c["SQ_INSTS_MFMA"])
            if "SQ_WAVE_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
                e["sq_wave_cycles"], e["sq_busy_cycles"] = mean(c["SQ_WAVE_CYCLES"]), mean(c["SQ_BUSY_CYCLES"])
        if "SQ_LDS_IDX_ACTIVE" in c and mean(c["SQ_LDS_IDX_ACTIVE"]):
            e["lds_bank_conflict_frac"] = mean(c["SQ_LDS_BANK_CONFLICT"]) / mean(c["SQ_LDS_IDX_ACTIVE"])
            e["lds_idx_active"] = mean(c["SQ_LDS_IDX_ACTIVE"])
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in merged[(name, grid)]:
            e["valu_mfma_coexec_frac"] = mean(c["SQ_VALU_MFMA_COEXEC_CYCLES"]) / max(mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]), 1.0)
        if "SQ_WAIT_ANY" in c and "SQ_ACTIVE_INST_ANY" in c:
            tot = mean(c["SQ_WAIT_ANY"]) + mean(c["SQ_ACTIVE_INST_ANY"]) + mean(c.get("SQ_WAIT_INST_ANY", [0]))
            e["wait_any_frac"] = mean(c["SQ_WAIT_ANY"]) / tot if tot else None
        e["us_under_profiler"] = mean(c.get("_us_profiled", []))
        kernels.setdefault(name, {"classes": {}})["classes"][grid if isinstance(grid, str) else "grid%d" % grid] = e
        f = lambda v, s=1.0, p="%.1f": (p % (v * s)) if v is not None else "-"
        print("%-44s %9d %6s %9s %10s %10s %8s %8s %8s %8s" % (
            ("%s [%s]" % (name, grid))[:44], n, f(e["us_under_profiler"]), f(e.get("hbm_bytes_per_launch"), 1e-6),
            f(e.get("fetch_bytes"), 1e-6), f(e.get("write_bytes"), 1e-6), f(e.get("mfma_util"), 100.0),
            f(e.get("lds_bank_conflict_frac"), 1.0, "%.4f"), f(e.get("valu_mfma_coexec_frac"), 1.0, "%.3f"),
            f(e.get("wait_any_frac"), 1.0, "%.3f")))
    bench = {}
    try:
        bench = json.loads(open(os.path.join(src, "bench_unprofiled.json")).read().strip().splitlines()[-1])
    except (OSError, ValueError, IndexError):
        pass
    cfg = bench.get("config", {})
    out = {"command": "rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "
                      "(one pass per group: fetch, write, mfma, lds; tools/pmc_passes.sh)",
           "csrc_sha16": csrc_sha16(),        # the kernel sources these counters belong to (bench.py compares)
           "batch": cfg.get("batch_per_gpu"), "size": 640, "unprofiled_bench": {k: bench.get(k) for k in ("value", "ms_per_step")},
           "notes": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); values are means per "
                    "launch over the 6 timed forwards; kernels are serialised under --pmc, so the times are not the bench's",
           "kernels": kernels}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
