#!/usr/bin/env python3
"""configs[0]: the per-image loop of the reference's ``validate_hhrnet.py`` (:84-105) on this repo's MI355X path.

    python tools/validate_two_images.py                          # the two bundled COCO images (fixture pixels)
    python tools/validate_two_images.py a.jpg b.jpg --checkpoint models/pose_higher_hrnet_w48_640.pth.tar

Per image, exactly the reference's steps with the accelerated pieces in their place:

    validate_hhrnet.py:85     PIL load, RGB                          -> PIL here too (or the fixture's pixels)
    :87-89  resize_align_multi_scale + ToTensor + Normalize (cv2)    -> rtpe_warp_normalize (one HIP pass)
    :93     preds, refined = hhrnet(t)                               -> compiled program on the HIP executor
    :94-101 2x F.interpolate to the ORIGINAL (h, w) + parser.parse   -> HeatmapParser.parse_lowres (fused; the
                                                                        upsampled maps are never built)
    :103-105 keep the non-empty people, collect scores               -> same

The COCO evaluation of :116 needs pycocotools and the annotation files, which are out of scope.  Without a
checkpoint (none can be fetched here) the teacher gets the seeded synthetic weights of oracle/synth.py, and
``--check`` compares the result with what the REFERENCE produced for the same weights and pixels on the CPU
(tests/golden/two_images.npz, made by tools/gen_golden.py).  oracle/ is used only for that check and for the seeded
weights, never for the inference itself.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "realtime-pose-estimation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

INPUT_SIZE = 640                      # validate_hhrnet.py:39
NUM_HEATMAPS = 17


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("images", nargs="*", help="image files (default: the two bundled COCO images, from the fixture)")
    ap.add_argument("--checkpoint", default="", help="pose_higher_hrnet_w48_640.pth.tar (1.-prefixed state dict)")
    ap.add_argument("--weights", default="W0", choices=["W0", "W1", "W2"], help="seeded weights when there is no checkpoint")
    ap.add_argument("--check", action="store_true", help="compare with the reference's CPU result in the fixture")
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()

    import __graft_entry__ as entry
    entry.build()
    from rtpe.engine import HM_PARSER_PARAMS
    from rtpe.helpers import build_hrnet_w48_teacher, get_hrnet_w48_teacher
    from rtpe.third_party import transforms
    from rtpe.third_party.group import HeatmapParser

    fixture = np.load(os.path.join(ROOT, "tests", "golden", "two_images.npz"))
    if args.images:
        from PIL import Image
        items = [(os.path.basename(p), np.array(Image.open(p).convert("RGB"))) for p in args.images]
    else:
        items = [(n, fixture[n + "_img"]) for n in ("000000001000", "000000002685")]

    if args.checkpoint:
        hhrnet = get_hrnet_w48_teacher(args.checkpoint).to(args.device)
    else:
        from oracle import synth
        with open(os.path.join(ROOT, "tests", "golden", "w48_shapes.json")) as f:
            shapes = {k: tuple(v) for k, v in json.load(f)["shapes"].items()}
        sd = synth.make_state_dict(shapes, 0, args.weights)
        hhrnet = build_hrnet_w48_teacher({"1." + k: v for k, v in sd.items()}).to(args.device)
    hhrnet.eval()
    hm_parser = HeatmapParser(num_joints=NUM_HEATMAPS, **HM_PARSER_PARAMS)

    all_preds, all_scores = [], []
    for ii, (name, img) in enumerate(items):
        h, w = img.shape[:2]
        t0 = time.perf_counter()
        t, center, scale = transforms.warp_normalize(img, INPUT_SIZE, 1, 1, device=args.device)
        with torch.no_grad():
            preds, refined = hhrnet(t)
        grouped, scores = hm_parser.parse_lowres(refined, preds[:, NUM_HEATMAPS:], (h, w))[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        final_results = [x for x in grouped if x.size > 0] if getattr(grouped, "ndim", 1) == 3 else []
        all_preds.append(final_results)
        all_scores.append(scores)
        print("%d processing %s %s -> network input %s: %d people, top scores %s  (%.1f ms incl. first-shape tuning)"
              % (ii, name, (w, h), tuple(t.shape[2:]), len(final_results),
                 [round(float(s), 4) for s in sorted(scores, reverse=True)[:3]], dt * 1e3))
        key = "%s_%s_" % (name, args.weights)
        if args.check and not args.checkpoint and key + "final" in fixture:
            ref_people, ref_scores = fixture[key + "final"], fixture[key + "scores"]
            hm_err = np.abs(refined.cpu().numpy()[:, :, ::8, ::8] - fixture[key + "refined_s8"].astype(np.float32)).max()
            print("   reference (CPU, same weights and pixels): %d people, top scores %s; heat maps within %.2e"
                  % (len(ref_people), [round(float(s), 4) for s in sorted(ref_scores, reverse=True)[:3]], hm_err))
            assert hm_err <= 1e-3, "heat maps differ from the reference by more than 1e-3"
    return all_preds, all_scores


if __name__ == "__main__":
    main()
