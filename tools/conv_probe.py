"""GPU probe: runs single conv layers through rtpe_conv2d_nhwc (for rocprofv3).

    python tools/conv_probe.py [case ...]      case = cin,cout,k,stride,H,W[,N[,res]]
"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from rtpe import _native as nat  # noqa: E402

DEFAULT = ["48,48,3,1,160,160,32,1", "96,96,3,1,80,80,32,1", "192,192,3,1,40,40,32,1",
           "384,384,3,1,20,20,32,1", "64,256,1,1,160,160,32,1", "256,96,3,2,160,160,32,0"]


def run_block(case, reps=3):
    """case = block,H,W,N : the fused BasicBlock kernel (rtpe_basicblock_nhwc)"""
    kind, H, W, N = case.split(",")
    H, W, N = int(H), int(W), int(N)
    C = 48
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, C, generator=g).half().to(dev)
    ws = [((torch.rand(C, C, 3, 3, generator=g) * 2 - 1) / (C * 9) ** 0.5).half().contiguous().numpy() for _ in range(2)]
    a = np.ones(C, np.float32)
    b = np.zeros(C, np.float32)
    y = torch.empty_like(x)
    fp = ctypes.POINTER(ctypes.c_float)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nat.check(nat.lib().rtpe_basicblock_nhwc(x.data_ptr(), N, H, W, ws[0].ctypes.data, a.ctypes.data_as(fp),
                                                 b.ctypes.data_as(fp), ws[1].ctypes.data, a.ctypes.data_as(fp),
                                                 b.ctypes.data_as(fp), y.data_ptr(), nat.stream_ptr(dev)))
        ts.append(time.perf_counter() - t0)
    print("%-28s host-inclusive best %.1f us" % (case, min(ts) * 1e6), flush=True)


def run(case, reps=3):
    if case.startswith("block"):
        return run_block(case, reps)
    v = [int(t) for t in case.split(",")]
    cin, cout, k, s, H, W = v[:6]
    N = v[6] if len(v) > 6 else 32
    use_res = bool(v[7]) if len(v) > 7 else False
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, H, W, cin, generator=g).half().to(dev)
    w = ((torch.rand(cout, cin, k, k, generator=g) * 2 - 1) / (cin * k * k) ** 0.5).half().contiguous().numpy()
    a = np.ones(cout, np.float32)
    b = np.zeros(cout, np.float32)
    Ho, Wo = H // s, W // s
    res = torch.randn(N, Ho, Wo, cout, generator=g).half().to(dev) if use_res else None
    y = torch.empty((N, Ho, Wo, cout), dtype=torch.float16, device=dev)
    fp = ctypes.POINTER(ctypes.c_float)
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nat.check(nat.lib().rtpe_conv2d_nhwc(
            x.data_ptr(), N, H, W, cin, w.ctypes.data, a.ctypes.data_as(fp), b.ctypes.data_as(fp), cout, k, s,
            nat.F_RELU | nat.F_ROUND_CONV, res.data_ptr() if use_res else None, y.data_ptr(),
            nat.stream_ptr(dev)))
        ts.append(time.perf_counter() - t0)
    flops = 2.0 * N * Ho * Wo * cout * cin * k * k
    print("%-28s host-inclusive best %.1f us (%.0f TFLOP/s incl. pack+sync)" % (case, min(ts) * 1e6,
                                                                               flops / min(ts) / 1e12), flush=True)


if __name__ == "__main__":
    # RTPE_PROBE_OPTS="stream_pc=2,direct_1x1=0": tuning options (rtpe_set_option) for this run
    for kv in filter(None, os.environ.get("RTPE_PROBE_OPTS", "").split(",")):
        k, v = kv.split("=")
        nat.check(nat.lib().rtpe_set_option(k.encode(), int(v)))
    for c in (sys.argv[1:] or DEFAULT):
        run(c)
