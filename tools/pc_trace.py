"""GPU probe (DIAGNOSTIC build only: RTPE_BUILD_DEFS=-DRTPE_DIAG): shader-clock stamps of one workgroup of the
producer / consumer BasicBlock kernel at 32 x 160 x 160 - where the waves of a unit spend their cycles.

    RTPE_BUILD_DEFS=-DRTPE_DIAG python tools/pc_trace.py
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


def main():
    N, H, W = 32, 160, 160
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, H, W, 48, generator=g).half().to("cuda:0")
    ws = [((torch.rand(48, 48, 3, 3, generator=g) * 2 - 1) / (48 * 9) ** 0.5).half().contiguous().numpy() for _ in range(2)]
    al = [(torch.rand(48, generator=g) * 0.4 + 0.8).numpy() for _ in range(2)]
    be = [(torch.randn(48, generator=g) * 0.1).numpy() for _ in range(2)]
    fpt = ctypes.POINTER(ctypes.c_float)
    L = nat.lib()
    st = nat.stream_ptr(x.device)
    y = torch.empty_like(x)
    nat.check(L.rtpe_set_option(b"block_pc", 1))
    for _ in range(5):
        nat.check(L.rtpe_basicblock_nhwc(x.data_ptr(), N, H, W, ws[0].ctypes.data, al[0].ctypes.data_as(fpt),
                                         be[0].ctypes.data_as(fpt), ws[1].ctypes.data, al[1].ctypes.data_as(fpt),
                                         be[1].ctypes.data_as(fpt), y.data_ptr(), st))
    torch.cuda.synchronize()
    raw = ctypes.CDLL(nat.lib()._name)
    if not hasattr(raw, "rtpe_diag_pc_trace"):
        sys.exit("tools/pc_trace.py needs a diagnostic build: RTPE_BUILD_DEFS=-DRTPE_DIAG (touch csrc/conv_block.hip first)")
    buf = (ctypes.c_ulonglong * 512)()
    assert raw.rtpe_diag_pc_trace(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(8, 64)
    t0 = t[t > 0].min()
    for wv in range(4):
        print("producer wave %d: per unit [wait at barrier | tile request | conv1 k loop | epilogue A | wait for tile] (10 ns ticks of the 100 MHz real-time counter)" % wv)
        for i in range(1, 10):
            s5 = t[wv, i * 5:(i + 1) * 5 + 1]
            print("  unit %2d  start %7d | %5d | %5d | %5d | %5d | %5d" % ((i, s5[0] - t0) + tuple(s5[1:] - s5[:-1])))
    for wv in range(4, 8):
        print("consumer wave %d: per interval [wait at barrier | epilogue B + stores | residual request | conv2 k loop | BN fetch]" % wv)
        for v in range(1, 10):
            s = t[wv, v * 5:(v + 1) * 5 + 1]
            print("  unit %2d  start %7d | %5d | %5d | %5d | %5d | %5d" % ((v, s[0] - t0) + tuple(s[1:] - s[:-1])))


if __name__ == "__main__":
    main()
