"""GPU probe: per-op-class HIP-event times of the teacher forward ALONE (no decode on a side
stream), batch 32 at 640x640 by default.

    python tools/forward_profile.py [batch] [size] [out.txt]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from rtpe.helpers import build_hrnet_w48_teacher  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
    out = sys.argv[3] if len(sys.argv) > 3 else None
    torch.manual_seed(0)
    if os.environ.get("RTPE_PROFILE_FP32", "0") == "1":         # configs[1]: the fp32 network without the half wrapper
        from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
        model = net = PoseHigherResolutionNet().to("cuda:0").eval()
    else:
        model = build_hrnet_w48_teacher().to("cuda:0")
        net = model[1]
    x = torch.randn(B, 3, S, S, device="cuda:0")
    with torch.no_grad():
        model(x)
    eng = net._engine(x.device)
    reps = 5
    acc = None
    for _ in range(reps):
        _, ms = eng.forward_timed(x)
        acc = np.array(ms) if acc is None else acc + np.array(ms)
    acc /= reps
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        eng.forward(x)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 5 * 1e3
    names = list(eng.program.names)
    print('plane-major inner tensors in this run: %d' % eng.plane_major_tensors(B, S, S), flush=True)
    by = {}
    for i, nm in enumerate(names):
        t = eng.op_tile(i, B, S, S)
        op = eng.program.ops[i]
        ds = eng.program.tensors[op.out_t].ds_log2 if op.out_t >= 0 else 1
        tag = ""
        if t[0]:
            kind = (" PAIR-HEAD (launched with the next op)" if t[7] == -800001 else " PAIR (both 1x1 convs)" if t[7] == -800002 else
                    " FUSED-STEM" if t[7] in (-600001, -600002) else " HEAD-DIRECT" if t[7] == -400001 else " DECONV48" if t[7] == -300001 else " CONV48S2" if t[7] == -200001 else " CONV48S2 (one launch with the next %d)" % (-t[7] - 200001) if t[7] in (-200002, -200003) else " (in the CONV48S2 launch above)" if t[7] == -200009 else " CONV64 %d workgroups" % (-t[7] - 500000) if -600000 < t[7] <= -500000 else
                    " S%d/%d" % (-t[7] % 100000, -t[7] // 100000) if t[7] <= -100000 else " P%d" % -t[7]) if t[7] < 0 else ""
            tag = " [m%s n%d w%d %dx%d cc%d cb%d%s]" % ((("%dx2" % -t[0]) if t[0] < 0 else str(t[0]),) + tuple(t[1:7]) + (kind,))
        key = "%s @/%d%s" % (nm, 1 << ds, tag)
        c = eng.op_cost(i, B, S, S)
        if t[0] and t[7] == -800001:                      # head of a 1x1 pair: its work is done by the launch at the next op
            pair_head = (acc[i], c[0], c[1])
            continue
        if t[0] and t[7] == -800002:
            key = "pair 64->256 k1s1+res+relu, 256->64 k1s1+relu @/%d [one kernel: 8 waves x 16-pixel tiles]" % (1 << ds)
            d = by.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += acc[i] + pair_head[0]
            d[2] += c[0] + pair_head[1]
            d[3] += c[1] + pair_head[2]
            continue
        if t[0] and t[7] == -600001:                      # stem + the conv behind it as one kernel, launched here
            stem_head = (acc[i], c[0])
            continue
        if t[0] and t[7] == -600002:
            key = "stem 3->64 k3s2 + conv 64->64 k3s2+relu @/%d [one kernel: 8 waves, 8x16 tiles, conv1 region in LDS]" % (1 << ds)
            d = by.setdefault(key, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += acc[i] + stem_head[0]
            d[2] += c[0] + stem_head[1]
            d[3] += B * 3 * S * S * 4 + B * (S // 4) * (S // 4) * 64 * 2      # fp32 image in, /4 map out
            continue
        if t[0] and t[7] == -900002:                      # second conv of a fused BasicBlock: its work belongs to the
            d = by[head_key]                              # launch of the first (time 0 here)
            d[1] += acc[i]
            d[2] += c[0]
            d[3] += c[1]
            continue
        if t[0] and t[7] == -900001:
            key = key.replace("conv 48->48 k3s1+relu", "block 48->48 (2 convs)")
            head_key = key
        d = by.setdefault(key, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += acc[i]
        d[2] += c[0]
        d[3] += c[1]
    width = max(68, max(len(k) for k in by))
    lines = ["# forward only, batch %d, %dx%d, per-op HIP events averaged over %d passes" % (B, S, S, reps),
             "# roof = max(FLOPs / 2.5 PFLOP/s, layer-fused bytes / 8 TB/s) / time; a fused BasicBlock is one row (both convs)",
             "%-68s %3s %9s %9s %9s %8s %6s" % ("op", "n", "ms_total", "TFLOP/s", "GB/s", "us/op", "roof")]
    for k, d in sorted(by.items(), key=lambda kv: -kv[1][1]):
        if d[1] > 0:
            roof = max(d[2] / 2.5e15, d[3] / 8e12) / (d[1] * 1e-3)
            lines.append("%-68s %3d %9.3f %9.1f %9.1f %8.1f %6.2f" % (k, d[0], d[1], d[2] / d[1] / 1e9, d[3] / d[1] / 1e6,
                                                                    d[1] / d[0] * 1e3, roof))
        else:
            lines.append("%-68s %3d %9.3f %9s %9s %8.1f %6s" % (k, d[0], 0.0, "-", "-", 0.0, "-"))
    fl, by_ = sum(d[2] for d in by.values()), sum(d[3] for d in by.values())
    lines.append("whole forward: %.1f GFLOP, %.1f GB layer-fused -> %.1f TFLOP/s, %.2f TB/s; max(FLOP, byte) bound / time = %.2f"
                 % (fl / 1e9, by_ / 1e9, fl / acc.sum() / 1e9, by_ / acc.sum() / 1e9,
                    max(fl / 2.5e15, by_ / 8e12) / (acc.sum() * 1e-3)))
    lines.append("forward total (events) %.3f ms; wall %.3f ms = %.1f img/s" % (acc.sum(), wall, B / wall * 1e3))
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
