"""GPU probe: A/B of kernel variants inside ONE process (interleaved rounds, per-op HIP events of the whole
forward at batch 32, 640x640): the fused BasicBlock kernel - resident weights vs weight ring vs producer / consumer.

    python tools/block_ab.py [rounds]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from rtpe import _native as nat  # noqa: E402
from rtpe.helpers import build_hrnet_w48_teacher  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.manual_seed(0)
    model = build_hrnet_w48_teacher().to("cuda:0").eval()
    net = model[1]
    x = torch.randn(32, 3, 640, 640, device="cuda:0")
    with torch.no_grad():
        model(x)
    eng = net._engine(x.device)
    names = list(eng.program.names)
    blocks = {}
    for i, nm in enumerate(names):
        t = eng.op_tile(i, 32, 640, 640)
        if t[7] == -900001:
            ds = eng.program.tensors[eng.program.ops[i].out_t].ds_log2
            blocks.setdefault(ds, []).append(i)
    res = {0: [], 1: [], 2: []}
    for r in range(rounds):
        for ring in (0, 1, 2):
            nat.check(nat.lib().rtpe_set_option(b"block_ring", int(ring == 1)))
            nat.check(nat.lib().rtpe_set_option(b"block_pc", int(ring == 2)))
            eng.forward(x)
            _, ms = eng.forward_timed(x)
            ms = np.array(ms)
            res[ring].append((float(ms.sum()), {ds: float(np.mean([ms[i] + ms[i + 1] for i in idx])) for ds, idx in blocks.items()}))
    nat.check(nat.lib().rtpe_set_option(b"block_ring", 0))
    nat.check(nat.lib().rtpe_set_option(b"block_pc", 1))
    for ring in (0, 1, 2):
        tot = [t for t, _ in res[ring]]
        line = "%-18s forward (events) median %.3f min %.3f ms" % (("resident weights", "weight ring", "producer/consumer")[ring], np.median(tot), min(tot))
        for ds in sorted(blocks):
            v = [b[ds] for _, b in res[ring]]
            line += " | block @/%d median %.1f min %.1f us" % (1 << ds, np.median(v) * 1e3, min(v) * 1e3)
        print(line, flush=True)


if __name__ == "__main__":
    main()
