"""GPU probe: host time of one forward call (launching ~290 kernels) against its GPU time, batch 1 at 640x640."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import torch
torch.set_num_threads(8)
import __graft_entry__ as entry
entry.build()
from rtpe.helpers import build_hrnet_w48_teacher
torch.manual_seed(0)
m = build_hrnet_w48_teacher().to("cuda:0")
for B in (1, 32):
    x = torch.randn(B, 3, 640, 640, device="cuda:0")
    with torch.no_grad():
        for _ in range(3):
            m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for _ in range(n):
            m(x)
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        t_all = (time.perf_counter() - t0) / n
        eng = m[1]._engine(x.device)
        t0 = time.perf_counter()
        for _ in range(n):
            eng.forward(x)
        t_eng = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
    print("batch %d: host time per model(x) call %.2f ms, per Engine.forward %.2f ms; wall per forward %.2f ms" % (B, t_host * 1e3, t_eng * 1e3, t_all * 1e3))
