"""GPU probe: K forwards in flight on K streams (K engines, K workspaces) against one after another."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import torch
torch.set_num_threads(8)
import __graft_entry__ as entry
entry.build()
from rtpe.helpers import build_hrnet_w48_teacher
torch.manual_seed(0)
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
KS = [int(k) for k in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3, 4]
K = max(KS)
ms = [build_hrnet_w48_teacher().to(dev) for _ in range(K)]
xs = [torch.randn(B, 3, 640, 640, device=dev) for _ in range(K)]
ss = [torch.cuda.Stream(dev) for _ in range(K)]
with torch.no_grad():
    for m, x in zip(ms, xs):
        for _ in range(2):
            m(x)
    torch.cuda.synchronize()
    for k in KS:
        n = 24 // k
        t0 = time.perf_counter()
        for _ in range(n):
            for i in range(k):
                with torch.cuda.stream(ss[i]):
                    ms[i](xs[i])
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / (n * k)
        print("batch %d, %d forwards in flight on %d streams: %.2f ms per forward (%.0f img/s)" % (B, k, k, t * 1e3, B / t), flush=True)
