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

  hrnet_640_w0.npz    640x640 half-wrapper outputs with W0 for images 0 / 17 / 31 of the batch-32 set
  hrnet_w2.npz        weight set W2 (heat maps of the teacher's span): 128x192 + 640x640
  two_images.npz      the two bundled data/*.jpg (decoded with PIL) through the body of
                      validate_hhrnet.py:84-105 on the CPU (W0 and W2)
  e2e_640.npz         the same loop body on a synthetic 640x640 input (W0 and W2)
  munkres_vectors.npz Munkres().compute() of the REAL PyPI package on tie-heavy cost matrices
  match_vectors.npz   match_by_tag (group.py:26-97) on that package, tie-heavy candidate tables
  student_steps.npz   AttentionStudentSteps (students.py:786-1063) outputs + its state-dict contract
  decode_branches.npz tag_per_joint=False (the AGS branch of legacy/valid_ae1dim.py), adjust / refine switched off
  ref_selfspread.npz  the reference against itself (oneDNN off / 1 thread): the spread the GPU tests bound HIP by

``rtpe/third_party/group.py`` imports the PyPI package ``munkres`` (unpinned, not vendored).  It
is not installed for this interpreter, but the pure-Python module of munkres 1.1.4 sits in the
image at /opt/conda/lib/python3.9/site-packages/munkres.py and loads under Python 3.10; it is put on
sys.modules by path, so every decode fixture comes from the reference's own code running on the
real package.  Only if that file is missing a stand-in backed by oracle/hungarian_ref.py is used
(with a loud note; it asserts scipy-optimal total cost on every call).

    python tools/gen_golden.py [--only base,w0_640,w2,two_images,e2e,munkres,match]
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


MUNKRES_PY = "/opt/conda/lib/python3.9/site-packages/munkres.py"


def _install_munkres():
    """the real package by path when the image has it, else the stand-in"""
    if os.path.exists(MUNKRES_PY):
        import importlib.util
        spec = importlib.util.spec_from_file_location("munkres", MUNKRES_PY)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        sys.modules["munkres"] = m
        print("munkres: real package %s from %s" % (getattr(m, "__version__", "?"), MUNKRES_PY))
        return True
    print("!!! munkres.py not found: using the oracle/hungarian_ref.py stand-in - tie-breaks are then UNPINNED !!!")
    _install_munkres_standin()
    return False


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


HP_KW = dict(max_num_people=30, detection_threshold=0.1, tag_threshold=1.0,
             use_detection_val=True, ignore_too_much=False, tag_per_joint=True,
             nms_ksize=5, nms_padding=2)          # validate_hhrnet.py:40-47
NUM_HEATMAPS = 17


class Ref:
    """the imported reference classes + the teacher factory (body of rtpe/helpers.py:37-72; the module
    itself needs torchvision)"""

    def __init__(self):
        sys.path.insert(0, REF)
        import warnings
        warnings.simplefilter("ignore")
        from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
        from rtpe.third_party.fp16_utils.fp16util import network_to_half
        from rtpe.third_party.group import HeatmapParser
        self.Net, self.to_half, self.Parser = PoseHigherResolutionNet, network_to_half, HeatmapParser
        with open(os.path.join(OUT, "w48_shapes.json")) as f:
            self.shapes = {k: tuple(v) for k, v in json.load(f)["shapes"].items()} \
                if os.path.exists(os.path.join(OUT, "w48_shapes.json")) else None

    def teacher(self, sd):
        m = self.to_half(self.Net())
        m.load_state_dict({"1." + k: v for k, v in sd.items()}, strict=True)
        return m.eval()

    def weights(self, variant):
        return synth.make_state_dict(self.shapes, 0, variant)


def _sample(t, stride):
    return t.numpy()[:, :, ::stride, ::stride].astype(np.float16)     # half-wrapper outputs ARE fp16 values


def gen_w0_640(ref):
    """W0 at the headline size: images 0 / 17 / 31 of the batch-32 set, half wrapper (helpers.py:69-71)"""
    model = ref.teacher(ref.weights("W0"))
    x = synth.make_images(32, 640, 640)
    out = {}
    for i in (0, 17, 31):
        with torch.no_grad():
            ph, rh = model(x[i:i + 1])
        st = 4 if i == 0 else 8
        out.update({"img%d_preds_s%d" % (i, st): _sample(ph, st), "img%d_refined_s%d" % (i, st): _sample(rh, st),
                    "img%d_preds_abs" % i: np.float64(ph.double().abs().sum()),
                    "img%d_refined_abs" % i: np.float64(rh.double().abs().sum())})
        print("W0 640 image", i, "range preds %.3f refined %.3f" % (float(ph.abs().max()), float(rh.abs().max())))
    np.savez_compressed(os.path.join(OUT, "hrnet_640_w0.npz"), **out)


def gen_w2(ref):
    """W2 (oracle/synth.py): inner activations of W1, heat maps of the real teacher's span"""
    sd = ref.weights("W2")
    model = ref.teacher(sd)
    out = {}
    x = synth.make_images(1, 128, 192)
    with torch.no_grad():
        ph, rh = model(x)
    out["small_preds"], out["small_refined"] = ph.numpy().astype(np.float16), rh.numpy().astype(np.float16)
    print("W2 128x192 heat maps [%.3f, %.3f] tags [%.3f, %.3f] refined [%.3f, %.3f]" % (
        float(ph[:, :17].min()), float(ph[:, :17].max()), float(ph[:, 17:].min()), float(ph[:, 17:].max()),
        float(rh.min()), float(rh.max())))
    x = synth.make_images(32, 640, 640)
    for i in (0, 31):
        with torch.no_grad():
            ph, rh = model(x[i:i + 1])
        st = 4 if i == 0 else 8
        out.update({"img%d_preds_s%d" % (i, st): _sample(ph, st), "img%d_refined_s%d" % (i, st): _sample(rh, st),
                    "img%d_preds_abs" % i: np.float64(ph.double().abs().sum()),
                    "img%d_refined_abs" % i: np.float64(rh.double().abs().sum())})
        print("W2 640 image", i, "heat maps [%.3f, %.3f] tags [%.3f, %.3f] refined [%.3f, %.3f]" % (
            float(ph[:, :17].min()), float(ph[:, :17].max()), float(ph[:, 17:].min()), float(ph[:, 17:].max()),
            float(rh.min()), float(rh.max())))
    np.savez_compressed(os.path.join(OUT, "hrnet_w2.npz"), **out)


def _loop_body(ref, model, t, h, w):
    """validate_hhrnet.py:91-101 on the CPU: forward, two bilinear upsamples to the ORIGINAL (h, w), parse.
    Also returns top_k's tables of the same maps (group.py:144-179)."""
    parser = ref.Parser(num_joints=NUM_HEATMAPS, **HP_KW)
    with torch.no_grad():
        preds, refined = model(t)
        hms = torch.nn.functional.interpolate(refined, (h, w), mode="bilinear", align_corners=True)
        aes = torch.nn.functional.interpolate(preds[:, NUM_HEATMAPS:, :, :], (h, w), mode="bilinear",
                                              align_corners=True)
    tk = parser.top_k(hms, aes.unsqueeze(-1))
    grouped, scores = parser.parse(hms, aes.unsqueeze(-1), adjust=True, refine=True)
    return preds, refined, tk, grouped[0], np.array(scores, np.float32)


def _pack(prefix, preds, refined, tk, final, scores, stride):
    d = {"preds_s%d" % stride: _sample(preds, stride), "refined_s%d" % stride: _sample(refined, stride),
         "preds_abs": np.float64(preds.double().abs().sum()), "refined_abs": np.float64(refined.double().abs().sum()),
         "val_k": tk["val_k"][0], "loc_k": tk["loc_k"][0].astype(np.int32), "tag_k": tk["tag_k"][0],
         "final": np.asarray(final, np.float32), "scores": scores}
    return {prefix + k: v for k, v in d.items()}


def gen_two_images(ref):
    """configs[0]: the per-image body of validate_hhrnet.py:84-105 on the two bundled JPEGs.  PIL decodes them
    (:85); cv2 is not in the image, so the warp of :87 is the numpy restatement of this repo's own convention
    (oracle/preprocess_ref.py, grey levels rounded to uint8 as cv2.warpAffine returns them); everything after it
    (:89-101) is the reference's code on the CPU with seeded weights."""
    from PIL import Image
    from oracle import preprocess_ref
    out = {}
    for name in ("000000001000", "000000002685"):
        img = np.array(Image.open(os.path.join(REF, "data", name + ".jpg")).convert("RGB"))
        h, w = img.shape[:2]
        t, center, scale = preprocess_ref.warp_normalize(img, 640, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
        t = torch.from_numpy(t)[None]
        out[name + "_img"] = img
        out[name + "_input_abs"] = np.float64(t.double().abs().sum())
        out[name + "_center_scale"] = np.concatenate([center, scale]).astype(np.float64)
        print(name, "image", img.shape, "network input", tuple(t.shape))
        for variant in ("W0", "W2"):
            res = _loop_body(ref, ref.teacher(ref.weights(variant)), t, h, w)
            out.update(_pack("%s_%s_" % (name, variant), *res, stride=8))
            print("  ", variant, "people", len(res[3]), "candidates > 0.1:", int((res[2]["val_k"] > 0.1).sum()),
                  "scores", res[4][:3])
    np.savez_compressed(os.path.join(OUT, "two_images.npz"), **out)


def gen_e2e(ref):
    """the same loop body at the headline size: synthetic 640x640 input (image 0 of the batch-32 set), decode
    at 640x640"""
    x = synth.make_images(32, 640, 640)[:1]
    out = {}
    for variant in ("W0", "W2"):
        res = _loop_body(ref, ref.teacher(ref.weights(variant)), x, 640, 640)
        out.update(_pack(variant + "_", *res, stride=8))
        print("e2e 640", variant, "people", len(res[3]), "candidates > 0.1:", int((res[2]["val_k"] > 0.1).sum()))
    np.savez_compressed(os.path.join(OUT, "e2e_640.npz"), **out)


def gen_selfspread(ref):
    """How far the reference is from ITSELF: the half wrapper (helpers.py:69-71) run a second time with PyTorch-CPU's
    other convolution kernels (torch.backends.mkldnn off) and with one thread instead of eight, same weights, same
    input.  fp16 storage after every conv / BN / add makes the result depend on the fp32 accumulation order inside the
    convolutions, which differs between those kernels: this spread is a property of the reference's arithmetic and is
    what the GPU parity tests bound the HIP path by.  Cases = the whole-net fixtures above (same sampling strides):
    W0 / W2 at 128x192, images 0 / 17 / 31 of the 640x640 batch-32 set, and the two bundled COCO images.
    Stored per case: the second run's sampled outputs (fp16) and full-tensor statistics of |run2 - run1| per map
    group (heat maps preds[:, :17], tags preds[:, 17:], refined): max, mean, fraction within 1e-3.
    ~200 s per 640x640 image with oneDNN off: the file is rewritten after every case."""
    from PIL import Image
    from oracle import preprocess_ref
    path = os.path.join(OUT, "ref_selfspread.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}

    def stats(a, b):
        d = (a.double() - b.double()).abs()
        return [float(d.max()), float(d.mean()), float((d <= 1e-3).double().mean())]

    def groups(p0, r0, p1, r1):
        return np.array([stats(p0[:, :17], p1[:, :17]), stats(p0[:, 17:], p1[:, 17:]), stats(r0, r1)], np.float64)

    def case(name, model, x, stride):
        if name + "_stats_mkldnn_off" in out:
            return
        import time
        t0 = time.time()
        with torch.no_grad():
            p0, r0 = model(x)                                   # as every other fixture: oneDNN on, 8 threads
            torch.set_num_threads(1)
            p1, r1 = model(x)
            torch.set_num_threads(8)
            with torch.backends.mkldnn.flags(enabled=False):
                p2, r2 = model(x)
        out[name + "_stats_threads1"] = groups(p0, r0, p1, r1)
        out[name + "_stats_mkldnn_off"] = groups(p0, r0, p2, r2)
        if stride:
            out[name + "_alt_preds_s%d" % stride], out[name + "_alt_refined_s%d" % stride] = _sample(p2, stride), _sample(r2, stride)
        else:
            out[name + "_alt_preds"], out[name + "_alt_refined"] = p2.numpy().astype(np.float16), r2.numpy().astype(np.float16)
        np.savez_compressed(path, **out)
        print("%-22s %5.0f s  oneDNN off vs on: heat %.2e / %.4f%%  tags %.2e / %.2f%%  refined %.2e / %.4f%%" % (
            name, time.time() - t0, out[name + "_stats_mkldnn_off"][0, 0], 100 * out[name + "_stats_mkldnn_off"][0, 2],
            out[name + "_stats_mkldnn_off"][1, 0], 100 * out[name + "_stats_mkldnn_off"][1, 2],
            out[name + "_stats_mkldnn_off"][2, 0], 100 * out[name + "_stats_mkldnn_off"][2, 2]), flush=True)

    imgs = {}
    for name in ("000000001000", "000000002685"):
        img = np.array(Image.open(os.path.join(REF, "data", name + ".jpg")).convert("RGB"))
        t, _, _ = preprocess_ref.warp_normalize(img, 640, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225))
        imgs[name] = torch.from_numpy(t)[None]
    xs = synth.make_images(1, 128, 192)
    xb = synth.make_images(32, 640, 640)
    for variant in ("W0", "W2"):
        model = ref.teacher(ref.weights(variant))
        case(variant + "_small", model, xs, 0)
        for i in ((0, 17, 31) if variant == "W0" else (0, 31)):
            case("%s_img%d" % (variant, i), model, xb[i:i + 1], 4 if i == 0 else 8)
        for name, t in imgs.items():
            case("%s_%s" % (variant, name), model, t, 8)
    # W1 (BatchNorm statistics away from identity; maps of +-4.2, beyond the teacher's span): the 128x192 case of
    # hrnet_small.npz and the 640x640 case of hrnet_640.npz (stride 8), so that those two tests are bounded by the
    # reference itself as well
    model = ref.teacher(ref.weights("W1"))
    case("W1_small", model, xs, 0)
    case("W1_640", model, synth.make_images(1, 640, 640), 8)


def gen_decode_branches(ref):
    """The public branches of HeatmapParser.parse the other fixtures never take (group.py:266-287):
    ``tag_per_joint=False`` with ONE tag map for all joints, tag tensor (1,1,h,w,1) - what the reference's
    upstream-faithful script runs (legacy/valid_ae1dim.py:177,191-199, the ``AGS`` branch) - and ``adjust=False`` /
    ``refine=False`` in every combination, on two blob sets each; plus the low-res -> bilinear -> parse pipeline of
    validate_hhrnet.py:93-101 with the same switches (the fused ``parse_lowres`` entry of this repo)."""
    out = {}

    def run(name, det_t, tag_t, per_joint, adjust, refine):
        kw = dict(HP_KW)
        kw["tag_per_joint"] = per_joint
        parser = ref.Parser(num_joints=NUM_HEATMAPS, **kw)
        ans, scores = parser.parse(det_t.clone(), tag_t.clone(), adjust=adjust, refine=refine)
        key = "%s_a%d_r%d" % (name, int(adjust), int(refine))
        out[key + "_n"] = np.array([len(a) if getattr(a, "ndim", 0) == 3 else 0 for a in ans], np.int32)
        out[key + "_final"] = np.asarray(ans[0], np.float32)
        out[key + "_scores"] = np.array(scores, np.float32)
        return len(ans[0])

    for name, P, h, w, seed in (("ags_p4", 4, 480, 640, 11), ("ags_p12", 12, 640, 640, 12)):
        det, tag = synth.make_decode_maps(P, h, w, seed=seed)
        ags = np.ascontiguousarray(tag.max(axis=1, keepdims=True))            # (1,1,h,w,1): one map for all joints
        out[name + "_meta"] = np.array([P, h, w, seed])
        det_t, ags_t = torch.from_numpy(det), torch.from_numpy(ags)
        kw = dict(HP_KW)
        kw["tag_per_joint"] = False
        tk = ref.Parser(num_joints=NUM_HEATMAPS, **kw).top_k(det_t, ags_t)
        out[name + "_val_k"], out[name + "_loc_k"], out[name + "_tag_k"] = tk["val_k"], tk["loc_k"].astype(np.int32), tk["tag_k"]
        n = [run(name, det_t, ags_t, False, a, r) for a, r in ((True, True), (False, True), (True, False), (False, False))]
        print(name, "people", n)
    for name, P, h, w, seed, D in (("sw_p6", 6, 480, 640, 13, 1), ("sw_p9_d2", 9, 320, 384, 14, 2)):
        det, tag = synth.make_decode_maps(P, h, w, seed=seed, tag_dim=D)
        out[name + "_meta"] = np.array([P, h, w, seed, D])
        n = [run(name, torch.from_numpy(det), torch.from_numpy(tag), True, a, r)
             for a, r in ((False, True), (True, False), (False, False))]
        print(name, "people", n)
    for name, P, H, W, oh, ow, seed in (("lowres_p5", 5, 640, 640, 640, 640, 15), ("lowres_p3_nonsq", 3, 640, 768, 555, 640, 16)):
        refined, tags = synth.make_lowres_maps(P, H, W, seed=seed)
        hms = torch.nn.functional.interpolate(torch.from_numpy(refined), (oh, ow), mode="bilinear", align_corners=True)
        aes = torch.nn.functional.interpolate(torch.from_numpy(tags), (oh, ow), mode="bilinear", align_corners=True)
        out[name + "_meta"] = np.array([P, H, W, oh, ow, seed])
        n = [run(name, hms, aes.unsqueeze(-1), True, a, r) for a, r in ((False, True), (True, False), (False, False))]
        print(name, "people", n)
    np.savez_compressed(os.path.join(OUT, "decode_branches.npz"), **out)


def gen_munkres(real):
    """Munkres().compute of the real package on cost matrices as match_by_tag builds them (group.py:57-80):
    round(dist)*100 - val with many equal entries, padded to square with 1e10 when there are more candidates than
    people; plus plain random and constant matrices"""
    from munkres import Munkres
    rng = np.random.default_rng(2024)
    mats, pairs = [], []
    for i in range(700):
        a, g = int(rng.integers(1, 31)), int(rng.integers(1, 31))
        kind = i % 7
        if kind == 0:
            c = rng.random((a, g)) * 1000
        elif kind == 1:
            c = np.round(rng.random((a, g)) * 3) * 100.0                       # only ties
        elif kind == 2:
            c = np.full((a, g), 100.0) - rng.random((a, 1)).astype(np.float32)  # cost depends on the row only
        else:
            c = np.round(rng.random((a, g)) * (2 + kind)) * 100 - rng.random((a, 1)).astype(np.float32)
        if a > g:
            c = np.concatenate((c, np.zeros((a, a - g)) + 1e10), axis=1)
        c = np.ascontiguousarray(c, np.float64)
        res = Munkres().compute(c.tolist() if i % 2 else c.copy())
        mats.append(c)
        pairs.append(np.array(res, np.int32).reshape(-1, 2))
    np.savez_compressed(os.path.join(OUT, "munkres_vectors.npz"),
                        real_package=np.array(int(real)),
                        shapes=np.array([m.shape for m in mats], np.int32),
                        costs=np.concatenate([m.reshape(-1) for m in mats]),
                        n_pairs=np.array([len(q) for q in pairs], np.int32),
                        pairs=np.concatenate(pairs))
    print("munkres vectors:", len(mats), "matrices, real package:", real)


def gen_match(real):
    """match_by_tag (group.py:26-97) of the reference on the real munkres package, on candidate tables built to
    provoke equal-cost assignments (quantised tags and values), the people cap and dict-key collisions"""
    sys.path.insert(0, REF)
    from rtpe.third_party.group import Params, match_by_tag
    rng = np.random.default_rng(77)
    J, K = 17, 30
    out = {"real_package": np.array(int(real))}
    settings = [dict(), dict(use_detection_val=False), dict(ignore_too_much=True, max_num_people=5),
                dict(max_num_people=8)]
    n = 0
    for trial in range(48):
        D = int(rng.choice([1, 1, 1, 2]))
        q = [0.0, 0.05, 0.25][trial % 3]                       # value quantum: 0 = continuous
        val = rng.random((J, K)).astype(np.float32) * (0.3 if trial % 4 == 0 else 1.0)
        if q:
            val = (np.round(val / q) * q).astype(np.float32)
        val = np.ascontiguousarray(np.sort(val, axis=1)[:, ::-1])
        loc = rng.integers(0, 640, (J, K, 2)).astype(np.int64)
        tag = rng.integers(0, 5, (J, K, D)) * (1.5 if trial % 2 else 0.5)
        if trial % 5:
            tag = tag + rng.normal(0, 0.2, (J, K, D))
        tag = tag.astype(np.float32)
        kw = dict(num_joints=J, max_num_people=30, detection_threshold=0.1, tag_threshold=1.0,
                  use_detection_val=True, ignore_too_much=False)
        kw.update(settings[trial % 4])
        ans = match_by_tag((tag, loc, val), Params(**kw))
        out["c%d_tag" % n], out["c%d_loc" % n], out["c%d_val" % n] = tag, loc.astype(np.int32), val
        out["c%d_cfg" % n] = np.array([kw["max_num_people"], int(kw["use_detection_val"]), int(kw["ignore_too_much"])])
        out["c%d_ans" % n] = np.asarray(ans, np.float32)
        n += 1
    out["n_cases"] = np.array(n)
    np.savez_compressed(os.path.join(OUT, "match_vectors.npz"), **out)
    print("match vectors:", n, "cases; people per case", [len(out["c%d_ans" % i]) for i in range(0, n, 6)])


def gen_student_steps(ref):
    """row 8f-3: AttentionStudentSteps(inplanes=48) (students.py:786-1063), seeded weights, x + alt (a seeded RGB image
    in LAB, oracle/student_ref.rgb2lab) at 2x3x320x320, att_divisor = 20"""
    sys.path.insert(0, REF)
    from rtpe.students import AttentionStudentSteps
    from oracle import student_ref
    torch.manual_seed(0)
    stu = AttentionStudentSteps(None, "cpu", 48, 17, 1, True, None, False).eval()
    shapes = {k: list(v.shape) for k, v in stu.state_dict().items()}
    with open(os.path.join(OUT, "student_steps_shapes.json"), "w") as f:
        json.dump({"n_keys": len(shapes), "shapes": shapes}, f)
    sd = synth.make_state_dict({k: tuple(v) for k, v in shapes.items()}, 4, "W1")
    stu.load_state_dict(sd, strict=True)
    x = synth.make_images(2, 320, 320, seed=77)
    rgb = torch.rand(2, 3, 320, 320, generator=torch.Generator().manual_seed(78))
    alt = torch.from_numpy(student_ref.rgb2lab(rgb.permute(0, 2, 3, 1).numpy()).astype(np.float32)).permute(0, 3, 1, 2).contiguous()
    with torch.no_grad():
        att, det = stu(x, alt=alt, att_divisor=20.0)
        att1, det1 = stu(x, alt=alt)
    oa, od = student_ref.student_steps_forward(sd, x, alt, 20.0, half_stem=True)
    print("student steps: keys", len(shapes), "oracle vs reference max diff", float((oa - att).abs().max()),
          float((od - det).abs().max()), "det range", float(det.abs().max()))
    np.savez_compressed(os.path.join(OUT, "student_steps.npz"), att=att.numpy(), det=det.numpy(),
                        att_nodiv_s2=att1.numpy()[:, :, ::2, ::2], det_nodiv_s2=det1.numpy()[:, :, ::2, ::2],
                        oracle_vs_reference_maxdiff=np.array([float((oa - att).abs().max()), float((od - det).abs().max())]))


def gen_base():
    sys.path.insert(0, REF)
    import warnings
    warnings.simplefilter("ignore")
    from rtpe.third_party.pose_higher_hrnet import PoseHigherResolutionNet
    from rtpe.third_party.fp16_utils.fp16util import network_to_half
    from rtpe.third_party.group import HeatmapParser

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
    sys.path.insert(0, REF)
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



def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="base,w0_640,w2,two_images,e2e,munkres,match,student_steps")
    only = set(ap.parse_args().only.split(","))
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    real = _install_munkres()
    if "base" in only:
        gen_base()
    ref = Ref()
    for name, fn in (("w0_640", gen_w0_640), ("w2", gen_w2), ("two_images", gen_two_images), ("e2e", gen_e2e),
                     ("student_steps", gen_student_steps), ("selfspread", gen_selfspread), ("decode_branches", gen_decode_branches)):
        if name in only:
            fn(ref)
    if "munkres" in only:
        gen_munkres(real)
    if "match" in only:
        gen_match(real)


if __name__ == "__main__":
    main()
