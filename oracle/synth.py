"""Seeded synthetic inputs for the hot path.

Test infrastructure (see oracle/__init__.py).  Everything is generated on the
CPU from explicit seeds so that the build container (where the goldens were
made from the reference) and the GPU box regenerate bit-identical inputs.
"""
import zlib

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# weights
# --------------------------------------------------------------------------- #
def _gen_for(seed, key):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 31 - 1))
    return g


def make_state_dict(shapes, seed=0, variant="W0", gain=1.0):
    """Synthetic HigherHRNet weights for a ``{key: shape}`` map.

    ``shapes`` is the un-prefixed key->shape map of the network (what
    ``PoseHigherResolutionNet().state_dict()`` has).  Every tensor is drawn
    from its own generator seeded by (seed, crc32(key)), so the result does
    not depend on module construction order.

    * conv / deconv weights: U(-b, b), b = gain / sqrt(fan_in)  (the PyTorch
      default-init scale; He-normal overflows fp16 through the residual stack,
      SURVEY.md section 8d).
    * biases: U(-b, b) with the same b.
    * W0: BatchNorm at identity.  W1: running_mean ~ N(0, .05),
      running_var ~ U(.9, 1.1), weight ~ U(.9, 1.1), bias ~ N(0, .05).
    * W2: W1, then the heat-map rows of head 0 (``final_layers.0`` outputs [:17], weight and
      bias) times 1/8 and all of head 1 (``final_layers.1``) times 1/4: the inner activations keep
      W1's O(1-4) range while the HEAT MAPS span about +-0.7, the span of the real teacher's
      [0, 1] maps on which BASELINE.json's 1e-3 tolerance is stated; the tag channels keep W1's range.
    """
    assert variant in ("W0", "W1", "W2")
    head_scale = variant == "W2"
    if head_scale:
        variant = "W1"
    sd = {}
    fan_in = {}
    for k, shp in shapes.items():
        if k.endswith(".weight") and len(shp) == 2:       # Linear (out, in)
            fan_in[k[:-7]] = shp[1]
        if k.endswith(".weight") and len(shp) == 4:
            # Conv2d: (Co, Ci, kh, kw) -> Ci*kh*kw.  ConvTranspose2d is
            # (Ci, Co, kh, kw); use the contraction length Ci*kh*kw/stride^2
            # ~ Ci*4 for the k4s2 deconv so activations keep their scale.
            if "deconv_layers" in k and k.endswith(".0.0.weight"):
                fan_in[k[:-7]] = shp[0] * shp[2] * shp[3] // 4
            else:
                fan_in[k[:-7]] = shp[1] * shp[2] * shp[3]
    for k, shp in shapes.items():
        g = _gen_for(seed, k)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith(".weight") and len(shp) in (2, 4):
            b = gain / float(fan_in[k[:-7]]) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        elif k.endswith(".bias") and k[:-5] in fan_in:
            b = gain / float(fan_in[k[:-5]]) ** 0.5
            sd[k] = (torch.rand(shp, generator=g) * 2 - 1) * b
        elif k.endswith("running_mean"):
            sd[k] = torch.zeros(shp) if variant == "W0" else torch.randn(shp, generator=g) * 0.05
        elif k.endswith("running_var"):
            sd[k] = torch.ones(shp) if variant == "W0" else torch.rand(shp, generator=g) * 0.2 + 0.9
        elif k.endswith(".weight"):  # BN gamma
            sd[k] = torch.ones(shp) if variant == "W0" else torch.rand(shp, generator=g) * 0.2 + 0.9
        elif k.endswith(".bias"):    # BN beta
            sd[k] = torch.zeros(shp) if variant == "W0" else torch.randn(shp, generator=g) * 0.05
        else:
            raise KeyError(k)
    if head_scale:
        for k in ("final_layers.0.weight", "final_layers.0.bias"):
            sd[k] = sd[k].clone()
            sd[k][:W2_HEATMAP_ROWS] *= W2_HEAD0_SCALE
        for k in ("final_layers.1.weight", "final_layers.1.bias"):
            sd[k] = sd[k] * W2_HEAD1_SCALE
    return sd


W2_HEATMAP_ROWS, W2_HEAD0_SCALE, W2_HEAD1_SCALE = 17, 0.125, 0.25


def make_images(n, h=640, w=640, seed=1234):
    """X images: CPU ``randn(n,3,h,w)`` float32 (ImageNet-normalised range)."""
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randn(n, 3, h, w, generator=g)


# --------------------------------------------------------------------------- #
# decode maps ("D" sets)
# --------------------------------------------------------------------------- #
def make_decode_maps(persons, h=640, w=640, seed=0, sigma=2.0, tag_dim=1,
                     noise=0.005, joints=17, tag_noise=0.02, drop=0.15):
    """Structured blob heatmaps + tag maps at decode resolution.

    Returns ``det (1,J,h,w) f32`` and ``tag (1,J,h,w,tag_dim) f32`` numpy
    arrays.  Each person p has a centre and J joint positions; the heatmap is
    the per-joint max over persons of a Gaussian (sigma px, amplitude
    U(.5,1)) plus N(0, noise); the tag map holds 3*(p+1) (+ small noise) on
    the blob support.  A fraction ``drop`` of (person, joint) blobs is left
    out so that ``refine`` has missing joints to fill.
    """
    rng = np.random.default_rng(seed)            # layout stream
    nrng = np.random.default_rng(seed + 7919)    # noise stream (size-dependent)
    det = np.zeros((joints, h, w), np.float32)
    tag = np.zeros((joints, h, w, tag_dim), np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    for p in range(persons):
        cy = rng.uniform(0.15 * h, 0.85 * h)
        cx = rng.uniform(0.15 * w, 0.85 * w)
        tagv = (3.0 * (p + 1) + rng.normal(0, 0.05, tag_dim)).astype(np.float32)
        for j in range(joints):
            if rng.uniform() < drop:
                continue
            jy = float(np.clip(cy + rng.normal(0, 0.08 * h), 3, h - 4))
            jx = float(np.clip(cx + rng.normal(0, 0.08 * w), 3, w - 4))
            amp = np.float32(rng.uniform(0.5, 1.0))
            r = int(4 * sigma) + 1
            y0, y1 = max(0, int(jy) - r), min(h, int(jy) + r + 2)
            x0, x1 = max(0, int(jx) - r), min(w, int(jx) + r + 2)
            blob = amp * np.exp(-((yy[y0:y1, x0:x1] - np.float32(jy)) ** 2 +
                                  (xx[y0:y1, x0:x1] - np.float32(jx)) ** 2)
                                / np.float32(2 * sigma * sigma))
            blob = blob.astype(np.float32)
            win = det[j, y0:y1, x0:x1]
            sel = blob > win
            win[sel] = blob[sel]
            tsel = sel & (blob > 0.01)
            tag[j, y0:y1, x0:x1][tsel] = tagv + nrng.normal(
                0, tag_noise, (int(tsel.sum()), tag_dim)).astype(np.float32)
    det += nrng.normal(0, noise, det.shape).astype(np.float32)
    return det[None], tag[None]


def make_lowres_maps(persons, H=640, W=640, seed=0):
    """Network-shaped decode inputs: ``refined (1,17,H/2,W/2)`` heatmaps and
    ``tags (1,17,H/4,W/4)`` (what forward() returns as refined / preds[:,17:]),
    to be bilinearly upsampled to (H, W) as validate_hhrnet.py:94-98 does."""
    det, _ = make_decode_maps(persons, H // 2, W // 2, seed=seed, sigma=1.5)
    _, tag = make_decode_maps(persons, H // 4, W // 4, seed=seed, sigma=1.0)
    # the two calls share the seed, hence the same person / joint layout
    # (positions scale with h, w); tags come from the quarter-res call.
    return det, tag[..., 0]
