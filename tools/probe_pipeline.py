"""GPU probe: host-side phase times of the pipelined teacher loop."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np, torch
torch.set_num_threads(int(os.environ.get('RTPE_THREADS', '8')))
def thr():
    try:
        return [l.strip() for l in open('/sys/fs/cgroup/cpu.stat') if 'throttled' in l]
    except OSError:
        return []
print('cpu.stat before', thr())
import __graft_entry__ as entry
entry.build()
from rtpe import engine
from rtpe.helpers import build_hrnet_w48_teacher
from rtpe.third_party.group import HeatmapParser

torch.manual_seed(0)
model = build_hrnet_w48_teacher().to("cuda:0")
pipe = engine.TeacherPipeline(model, device="cuda:0")
x = torch.randn(32, 3, 640, 640, device="cuda:0")
P = pipe.parser
for _ in range(2):
    pipe(x)
torch.cuda.synchronize()
T = {"fwd": [], "topk": [], "match": [], "finish": []}
EV = []
ALLOC = []
def mark(tag):
    e = torch.cuda.Event(enable_timing=True); e.record(); EV.append((tag, e))
topk_done = refine_done = None
t_all = time.perf_counter()
with torch.no_grad():
    for k in range(12):
        a = time.perf_counter()
        mark("F%d<" % k)
        preds, refined = model(x)
        mark("F%d>" % k)
        b = time.perf_counter()
        c = b
        if topk_done is not None:
            P.lowres_match(topk_done)
            mark("R%d>" % (k - 1))
        d = time.perf_counter()
        st = P.lowres_topk(refined, preds[:, 17:], (640, 640))
        mark("T%d>" % k)
        if refine_done is not None:
            P.lowres_finish(refine_done)
        e = time.perf_counter()
        refine_done, topk_done = topk_done, st
        ms = torch.cuda.memory_stats()
        ALLOC.append((ms.get("num_device_alloc", 0), ms.get("num_device_free", 0), ms.get("reserved_bytes.all.current", 0) >> 20))
        T["fwd"].append(b - a); T["topk"].append(c - b); T["match"].append(d - c); T["finish"].append(e - d)
torch.cuda.synchronize()
print("total per step %.2f ms" % ((time.perf_counter() - t_all) / 12 * 1e3))
for k, v in T.items():
    print(k, " ".join("%.1f" % (t * 1e3) for t in v))

base = EV[0][1]
prev = 0.0
for tag, e in EV:
    t = base.elapsed_time(e)
    print("%-5s at %8.2f ms (+%.2f)" % (tag, t, t - prev))
    prev = t

print("device allocs/frees/reserved MB per iteration:", ALLOC)

print('cpu.stat after', thr())
