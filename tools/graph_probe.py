"""GPU probe: the teacher forward captured in a HIP graph (torch.cuda.CUDAGraph around the ABI call) against the plain call,
batch 1 / 4 at 640x640, lanes on and off.

    python tools/graph_probe.py [out.txt]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from rtpe import _native as nat  # noqa: E402
from rtpe.helpers import build_hrnet_w48_teacher  # noqa: E402
from rtpe.third_party import pose_higher_hrnet as ph  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    lines = []
    torch.manual_seed(0)
    model = build_hrnet_w48_teacher().to("cuda:0")
    for B in (1, 4):
        x = torch.randn(B, 3, 640, 640, device="cuda:0")
        with torch.no_grad():
            want = model(x)
        for lanes in (1, 0):
            nat.check(nat.lib().rtpe_set_option(b"lanes", lanes))
            with torch.no_grad():
                model(x)
                eager = timeit(lambda: model(x))
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    model(x)
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(g):
                    got = model(x)
                replay = timeit(g.replay)
                g.replay()
                torch.cuda.synchronize()
                same = all(torch.equal(a, b) for a, b in zip(got, want))
            lines.append("batch %d, lanes %d: plain call %.3f ms per forward, graph replay %.3f ms, same bits %s" % (B, lanes, eager, replay, same))
            print(lines[-1], flush=True)
    nat.check(nat.lib().rtpe_set_option(b"lanes", 1))
    if out:
        open(out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
