"""GPU probe: batch-1 latency of the teacher (configs[1]: fp32; and the half wrapper) and of the student."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import torch
torch.set_num_threads(8)
import __graft_entry__ as entry
entry.build()
from rtpe.helpers import build_hrnet_w48_teacher
from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
from rtpe.third_party.group import HeatmapParser
from rtpe.engine import HM_PARSER_PARAMS, NUM_HEATMAPS
from rtpe.students import AttentionStudent


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
dev = "cuda:0"
half = build_hrnet_w48_teacher().to(dev)
fp32 = PoseHigherResolutionNet().to(dev).eval()
parser = HeatmapParser(num_joints=NUM_HEATMAPS, **HM_PARSER_PARAMS)
with torch.no_grad():
    for B in (1, 4):
        x = torch.randn(B, 3, 640, 640, device=dev)
        print("teacher half wrapper, batch %d: forward %.2f ms" % (B, timeit(lambda: half(x))))
        print("teacher fp32 (configs[1]), batch %d: forward %.2f ms" % (B, timeit(lambda: fp32(x))))

        def e2e():
            p, r = half(x)
            return parser.parse_lowres(r, p[:, NUM_HEATMAPS:], (640, 640))
        print("teacher half wrapper, batch %d: forward + decode (synchronous) %.2f ms" % (B, timeit(e2e, 10)))
    stu = AttentionStudent(None, "cpu", 100, 17, 1, True, None, False).eval().to(dev)
    for B in (1, 32):
        xs = torch.randn(B, 3, 320, 320, device=dev)
        print("AttentionStudent(100) half stem, batch %d @320: forward %.2f ms" % (B, timeit(lambda: stu(xs))))
