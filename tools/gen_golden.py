#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE itself on the CPU.

Build-container only: imports /root/reference (read-only, never copied; the
GPU box has no such path and nothing at test time reads it).  Run:

    python tools/gen_golden.py            # writes tests/golden/

What gets pinned:
  w48_shapes.json     the 1810-key state-dict contract (names, shapes, dtypes)
  hrnet_small.npz     PoseHigherResolutionNet / half-wrapper outputs, 128x192
  hrnet_640.npz       640x640 half-wrapper outputs, strided sample + checksums
  decode_*.npz        HeatmapParser.top_k / match / adjust / parse results on
                      the synthetic blob maps of oracle/synth.py
  bilinear.npz        F.interpolate(bilinear, align_corners=True) sample

``rtpe/third_party/group.py`` imports the PyPI package ``munkres`` which is not
installed here; following SURVEY.md section 8c a stand-in module is put on
sys.modules for the import.  Equal-cost assignment ties are common on this
path (costs are ``round(dist)*100 - val``), and they change the decoded
people, so the stand-in answers ``Munkres().compute(cost)`` with
oracle/hungarian_ref.py - a restatement of the package's published procedure
including its scan orders - and asserts on every call that scipy's optimal
assignment has the same total cost.  Consequence: decode fixtures pin
everything in group.py against the reference's own code; which of several
*equal-cost* optima the real package would pick is pinned only as far as that
restatement is faithful ("parity unpinned" for ties).
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import synth  # noqa: E402


def _install_munkres_standin():
    from scipy.optimize import linear_sum_assignment
    from oracle.hungarian_ref import munkres_compute

    class Munkres:
        def compute(self, cost):
            cost = np.array(cost, dtype=np.float64)     # copy: the package works in place
            pairs = munkres_compute(cost)
            r, c = linear_sum_assignment(cost)
            best = cost[r, c].sum()
            mine = sum(cost[i, j] for i, j in pairs)
            assert len(pairs) == min(cost.shape) and abs(mine - best) <= 1e-6 * max(1.0, abs(best)), \
                (mine, best)
            return pairs

    m = types.ModuleType("munkres")
    m.Munkres = Munkres
    sys.modules["munkres"] = m


def main():
    os.makedirs(OUT, exist_ok=True)
    _install_munkres_standin()
    sys.path.insert(0, REF)
    import warnings
    warnings.simplefilter("ignore")
    from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
    from rtpe.third_party.fp16_utils.fp16util import network_to_half
    from rtpe.third_party.group import HeatmapParser
    torch.set_num_threads(8)

    # ---- state-dict contract ------------------------------------------------
    net = PoseHigherResolutionNet().eval()
    ref_sd = net.state_dict()
    shapes = {k: list(v.shape) for k, v in ref_sd.items()}
    with open(os.path.join(OUT, "w48_shapes.json"), "w") as f:
        json.dump({"n_keys": len(shapes),
                   "n_params": int(sum(p.numel() for p in net.parameters())),
                   "shapes": shapes}, f)
    print("keys", len(shapes))

    def teacher(sd):
        """body of rtpe/helpers.py:37-72 (the module itself needs torchvision)"""
        m = network_to_half(PoseHigherResolutionNet())
        m.load_state_dict({"1." + k: v for k, v in sd.items()}, strict=True)
        return m.eval()

    # ---- small whole-net outputs -------------------------------------------
    small = {}
    x = synth.make_images(1, 128, 192)
    for variant in ("W0", "W1"):
        sd = synth.make_state_dict({k: tuple(v) for k, v in shapes.items()}, 0, variant)
        net.load_state_dict(sd, strict=True)
        with torch.no_grad():
            p, r = net(x)
            ph, rh = teacher(sd)(x)
        small[variant + "_fp32_preds"] = p.numpy()
        small[variant + "_fp32_refined"] = r.numpy()
        small[variant + "_half_preds"] = ph.numpy().astype(np.float16)
        small[variant + "_half_refined"] = rh.numpy().astype(np.float16)
        print(variant, "small", float(p.abs().max()), float((p - ph).abs().max()))
    np.savez_compressed(os.path.join(OUT, "hrnet_small.npz"), **small)

    # ---- 640x640, half wrapper, W1: sample + checksums -----------------------
    sd = synth.make_state_dict({k: tuple(v) for k, v in shapes.items()}, 0, "W1")
    x = synth.make_images(1, 640, 640)
    with torch.no_grad():
        ph, rh = teacher(sd)(x)
    np.savez_compressed(
        os.path.join(OUT, "hrnet_640.npz"),
        preds_s8=ph.numpy()[:, :, ::8, ::8].astype(np.float16),
        refined_s8=rh.numpy()[:, :, ::8, ::8].astype(np.float16),
        preds_sum=np.float64(ph.double().sum()), preds_abs=np.float64(ph.double().abs().sum()),
        refined_sum=np.float64(rh.double().sum()), refined_abs=np.float64(rh.double().abs().sum()))
    print("640 done", float(ph.abs().max()), float(rh.abs().max()))

    # ---- bilinear -----------------------------------------------------------
    g = torch.Generator().manual_seed(5)
    xb = torch.randn(1, 2, 20, 28, generator=g)
    yb = torch.nn.functional.interpolate(xb, (53, 77), mode="bilinear", align_corners=True)
    np.savez_compressed(os.path.join(OUT, "bilinear.npz"), x=xb.numpy(), y=yb.numpy())

    # ---- decode --------------------------------------------------------------
    hp_kw = dict(max_num_people=30, detection_threshold=0.1, tag_threshold=1.0,
                 use_detection_val=True, ignore_too_much=False, tag_per_joint=True,
                 nms_ksize=5, nms_padding=2)          # validate_hhrnet.py:40-47
    cases = [  # name, persons, h, w, seed, tag_dim
        ("p0", 0, 640, 640, 0, 1), ("p1", 1, 640, 640, 1, 1), ("p3", 3, 640, 640, 0, 1),
        ("p10", 10, 640, 640, 2, 1), ("p30", 30, 640, 640, 3, 1), ("p3_480", 3, 480, 640, 1, 1),
        ("p5_d2", 5, 320, 256, 4, 2), ("p40", 40, 640, 640, 5, 1),
    ]
    for name, P, h, w, seed, D in cases:
        det, tag = synth.make_decode_maps(P, h, w, seed=seed, tag_dim=D)
        det_t, tag_t = torch.from_numpy(det), torch.from_numpy(tag)
        parser = HeatmapParser(num_joints=17, **hp_kw)
        tk = parser.top_k(det_t, tag_t)
        matched = parser.match(**tk)
        adjusted = parser.adjust([m.copy() for m in matched], det_t)
        ans, scores = parser.parse(det_t, tag_t, adjust=True, refine=True)
        np.savez_compressed(
            os.path.join(OUT, "decode_%s.npz" % name),
            meta=np.array([P, h, w, seed, D]),
            val_k=tk["val_k"], loc_k=tk["loc_k"], tag_k=tk["tag_k"],
            matched=matched[0], adjusted=adjusted[0], final=ans[0],
            scores=np.array(scores, np.float32))
        print(name, "people found", len(ans[0]), "scores", np.array(scores)[:4])

    # low-res (network-shaped) maps -> bilinear -> parse, as validate_hhrnet.py:93-101
    for name, P, H, W, oh, ow, seed in [("lowres_p4", 4, 640, 640, 640, 640, 7),
                                        ("lowres_p2_nonsq", 2, 640, 896, 427, 640, 8)]:
        refined, tags = synth.make_lowres_maps(P, H, W, seed=seed)
        hms = torch.nn.functional.interpolate(torch.from_numpy(refined), (oh, ow),
                                              mode="bilinear", align_corners=True)
        aes = torch.nn.functional.interpolate(torch.from_numpy(tags), (oh, ow),
                                              mode="bilinear", align_corners=True)
        parser = HeatmapParser(num_joints=17, **hp_kw)
        tk = parser.top_k(hms, aes.unsqueeze(-1))
        ans, scores = parser.parse(hms, aes.unsqueeze(-1), adjust=True, refine=True)
        np.savez_compressed(
            os.path.join(OUT, "decode_%s.npz" % name),
            meta=np.array([P, H, W, oh, ow, seed]),
            val_k=tk["val_k"], loc_k=tk["loc_k"], tag_k=tk["tag_k"],
            final=ans[0], scores=np.array(scores, np.float32))
        print(name, "people found", len(ans[0]))


    # ---- config 5: AttentionStudent(inplanes=100), students.py:595-771 ---------------
    from rtpe.students import AttentionStudent
    from oracle import student_ref
    torch.manual_seed(0)
    stu = AttentionStudent(None, "cpu", 100, 17, 1, True, None, False).eval()
    st_shapes = {k: list(v.shape) for k, v in stu.state_dict().items()}
    with open(os.path.join(OUT, "student_shapes.json"), "w") as f:
        json.dump({"n_keys": len(st_shapes), "shapes": st_shapes}, f)
    ssd = synth.make_state_dict({k: tuple(v) for k, v in st_shapes.items()}, 3, "W1")
    stu.load_state_dict(ssd, strict=True)          # stem conv weights are rounded to fp16 by the copy
    xs = synth.make_images(2, 320, 320, seed=99)
    with torch.no_grad():
        att, det = stu(xs)
    oa, od = student_ref.student_forward(ssd, xs, half_stem=True)
    print("student synthetic: oracle vs reference max diff", float((oa - att).abs().max()), float((od - det).abs().max()))
    # the bundled trained attention weights (assets/pretrained_segm_4MB) validate the restatement too
    import glob
    pre = glob.glob(os.path.join(REF, "assets", "pretrained_segm_4MB", "*mid_stem.statedict"))[0][:-len("mid_stem.statedict")]
    stu.load_state_dicts(pre)
    with torch.no_grad():
        att_b, det_b = stu(xs)
    ob_a, ob_d = student_ref.student_forward(stu.state_dict(), xs, half_stem=True)
    d_b = (float((ob_a - att_b).abs().max()), float((ob_d - det_b).abs().max()))
    print("student bundled weights: oracle vs reference max diff", d_b)
    np.savez_compressed(os.path.join(OUT, "student.npz"), att=att.numpy(), det=det.numpy(),
                        bundled_oracle_vs_reference_maxdiff=np.array(d_b))


if __name__ == "__main__":
    main()
