"""GPU probe: the direct 1x1 conv kernel (option direct_1x1 = 1) against the one-workgroup-per-tile kernel (= 0), bit for bit,
through rtpe_conv2d_nhwc, with host-inclusive and device timings.

    python tools/direct_check.py [case ...]      case = cin,cout,H,W,N,res,relu
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from rtpe import _native as nat  # noqa: E402

DEFAULT = ["64,256,160,160,32,1,1", "256,64,160,160,32,0,1", "64,256,160,160,32,0,0", "64,64,160,160,32,0,1", "96,48,80,80,32,0,0",
           "192,48,40,40,32,0,0", "192,96,40,40,32,0,0", "384,48,20,20,32,0,0", "384,96,20,20,32,0,0", "384,192,20,20,32,0,0",
           "64,256,23,37,3,1,1", "256,64,9,7,2,0,1", "96,48,5,3,1,0,0"]


def run(case):
    cin, cout, H, W, N, use_res, relu = [int(t) for t in case.split(",")]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn(N, H, W, cin, generator=g).half().to(dev)
    w = ((torch.rand(cout, cin, 1, 1, generator=g) * 2 - 1) / cin ** 0.5).half().contiguous().numpy()
    a = (torch.rand(cout, generator=g) * 0.4 + 0.8).numpy()
    b = (torch.randn(cout, generator=g) * 0.1).numpy()
    res = torch.randn(N, H, W, cout, generator=g).half().to(dev) if use_res else None
    fp = ctypes.POINTER(ctypes.c_float)
    outs = []
    for opt in (0, 1):
        nat.check(nat.lib().rtpe_set_option(b"direct_1x1", opt))
        y = torch.full((N, H, W, cout), float("nan"), dtype=torch.float16, device=dev)
        nat.check(nat.lib().rtpe_conv2d_nhwc(
            x.data_ptr(), N, H, W, cin, w.ctypes.data, a.ctypes.data_as(fp), b.ctypes.data_as(fp), cout, 1, 1,
            (nat.F_RELU if relu else 0) | nat.F_ROUND_CONV, res.data_ptr() if use_res else None, y.data_ptr(), nat.stream_ptr(dev)))
        outs.append(y.cpu())
    nat.check(nat.lib().rtpe_set_option(b"direct_1x1", 1))
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    print("%-30s identical %s (differing %d, nan %d)" % (case, same, int((outs[0].view(torch.int16) != outs[1].view(torch.int16)).sum()),
                                                        int(torch.isnan(outs[1].float()).sum())), flush=True)
    return same


if __name__ == "__main__":
    ok = all([run(c) for c in (sys.argv[1:] or DEFAULT)])
    print("ALL IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
