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
        t0, c0 = time.perf_counter(), time.process_time()
        for _ in range(n):
            eng.forward(x)
        t_eng, c_eng = (time.perf_counter() - t0) / n, (time.process_time() - c0) / n
        torch.cuda.synchronize()
        # one call at a time (queue empty when the call starts): what enqueueing a forward costs without back-pressure from
        # a full hardware queue (at batch 32 a forward takes 13.7 ms on the GPU: back-to-back calls block in hipLaunchKernel)
        ts = []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.forward(x)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        n_ops = len(eng.program.ops)
    print("batch %d: host time per model(x) call %.2f ms, per Engine.forward %.2f ms wall / %.2f ms of CPU time (back to back, queue "
          "full at large batches); one call into an empty queue %.2f ms = %.1f us per op (%d ops); wall per forward %.2f ms" % (
              B, t_host * 1e3, t_eng * 1e3, c_eng * 1e3, min(ts) * 1e3, min(ts) * 1e6 / n_ops, n_ops, t_all * 1e3))
