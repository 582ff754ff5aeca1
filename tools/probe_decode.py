"""GPU probe: statistics of random-weight heat maps and decode timings."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")]
import numpy as np, torch
from oracle import synth
from rtpe.helpers import build_hrnet_w48_teacher
from rtpe.third_party.group import HeatmapParser
from rtpe import engine
shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(ROOT, "tests/golden/w48_shapes.json")))["shapes"].items()}
for variant in sys.argv[1:] or ["W0"]:
    sd = synth.make_state_dict(shapes, 0, variant)
    m = build_hrnet_w48_teacher({"1." + k: v for k, v in sd.items()}).to("cuda:0")
    B = 8
    x = torch.randn(B, 3, 640, 640, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(1))
    with torch.no_grad():
        preds, refined = m(x)
    torch.cuda.synchronize()
    print(variant, "refined mean %.4f std %.4f max %.4f | tags mean %.4f std %.4f" % (
        refined.mean(), refined.std(), refined.max(), preds[:, 17:].mean(), preds[:, 17:].std()), flush=True)
    hp = HeatmapParser(17, 30, 0.1, 1.0, True, False)
    hms = torch.nn.functional.interpolate(refined, (640, 640), mode="bilinear", align_corners=True)
    aes = torch.nn.functional.interpolate(preds[:, 17:], (640, 640), mode="bilinear", align_corners=True)
    tk = hp.top_k(hms, aes.unsqueeze(-1))
    v = tk["val_k"]
    print(" val_k[:, :, 0] per joint (img0):", np.round(v[0, :, 0], 3), flush=True)
    print(" val_k[:, :, 29] per joint (img0):", np.round(v[0, :, 29], 3), flush=True)
    print(" n(val>0.1) per image:", (v > 0.1).sum((1, 2)), flush=True)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = hp.parse_lowres(refined, preds[:, 17:], (640, 640))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        print(" parse_lowres batch %d: %.1f ms; people/img %s" % (B, (t1 - t0) * 1e3, [len(p) if p.ndim == 3 else 0 for p, _ in res]), flush=True)
