"""PyTorch-CPU restatement of the dual-head student of config 5 (oracle).

Test infrastructure (see oracle/__init__.py).  Follows /root/reference/rtpe/students.py:
``AttentionStudent.forward`` :724-771, ``ContextAwareModule.forward`` :180-201,
``SELayer.forward`` :137-142, ``StemHRNet.forward`` :242-255.  Evaluated straight from the
state dict of ``AttentionStudent`` (keys ``stem.1.*``, ``mid_stem.*``, ``att_*``, ``det_*``).
"""
import torch
import torch.nn.functional as F

from .hrnet_ref import BN_EPS, _Net


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.1, BN_EPS)


def _cbr(x, sd, p, dilation=1):
    w = sd[p + "0.weight"]
    pad = dilation * (w.shape[-1] // 2)
    return F.relu(_bn(F.conv2d(x, w, None, 1, pad, dilation), sd, p + "1."))


def _cam(x, sd, p):
    """ContextAwareModule.forward :180-201 (dilation of branch i is i+1, :649-704)"""
    res = _cbr(x, sd, p + "residual.")
    g = x.mean((2, 3))                                              # AdaptiveAvgPool2d(1).view(b, c)
    g = F.relu(F.linear(g, sd[p + "se.fc.0.weight"], sd[p + "se.fc.0.bias"]))
    g = torch.sigmoid(F.linear(g, sd[p + "se.fc.2.weight"], sd[p + "se.fc.2.bias"]))[:, :, None, None]
    n = 0
    while (p + "hdcs.%d.0.weight" % n) in sd:
        n += 1
    out = torch.cat([_cbr(x, sd, p + "hdcs.%d." % i, i + 1) for i in range(n)], 1)
    out = _cbr(out, sd, p + "hdc_top.")
    return F.relu(res + out * g.expand_as(out))


def _pool(x):
    return F.avg_pool2d(x, 3, 2, 1, count_include_pad=False)


@torch.no_grad()
def student_forward(sd, x, half_stem=True):
    """x (N,3,H,W) float32 -> (att (N,1,H/4,W/4), det (N,18,H/4,W/4)) float32"""
    sd = {k: v.detach() for k, v in sd.items()}
    stem = _Net({k[len("stem.1."):]: v for k, v in sd.items() if k.startswith("stem.1.")}, half_stem)
    s = stem.stem(x).float()                                        # tofp32 of the wrapped stem
    f = {k: v.float() for k, v in sd.items() if not k.startswith("stem.")}
    s = _cbr(s, f, "mid_stem.")                                     # mid_stem[0..2]
    s = F.relu(_bn(F.conv2d(s, f["mid_stem.3.weight"], None, 1, 1), f, "mid_stem.4."))
    hw = s.shape[-2:]
    hi = _cam(s, f, "att_hi.0.")
    mid = _cam(_pool(s), f, "att_mid.1.")
    lo = _cam(_pool(mid), f, "att_lo.1.")
    mid = F.interpolate(lo, hw, mode="nearest")                     # both from lo, :739-742
    lo = F.interpolate(lo, hw, mode="nearest")
    att = hi + mid + lo
    att = F.conv2d(att, f["att_top.0.weight"], f["att_top.0.bias"], 1, 1)
    att = torch.sigmoid(att / 20)
    s = s + att.expand(s.shape)
    hi = _cam(s, f, "det_hi.0.")
    lo = _cam(_pool(hi), f, "det_lo.1.")                            # det_hi feeds both, det_mid unused, :759-761
    up = F.interpolate(lo, hw, mode="nearest")
    det = hi + up + up
    det = F.conv2d(det, f["det_top.0.weight"], f["det_top.0.bias"], 1, 1)
    return att, det


# --------------------------------------------------------------------------- #
# row 8f-3: AttentionStudentSteps (students.py:786-1063) and the alt colour spaces
# --------------------------------------------------------------------------- #
@torch.no_grad()
def student_steps_forward(sd, x, alt, att_divisor=None, half_stem=True):
    """``AttentionStudentSteps.forward(x, alt=alt, att_divisor=...)`` :966-1063 -> (att, det)"""
    sd = {k: v.detach() for k, v in sd.items()}
    stem = _Net({k[len("stem.1."):]: v for k, v in sd.items() if k.startswith("stem.1.")}, half_stem)
    s = stem.stem(x).float()
    f = {k: v.float() for k, v in sd.items() if not k.startswith("stem.")}
    s = _cbr(s, f, "mid_stem.")
    s = F.relu(_bn(F.conv2d(s, f["mid_stem.3.weight"], None, 1, 1), f, "mid_stem.4."))
    a = F.relu(_bn(F.conv2d(alt, f["alt_img_stem.0.weight"], None, 2, 2), f, "alt_img_stem.1."))      # :984
    a = F.relu(_bn(F.conv2d(a, f["alt_img_stem.3.weight"], None, 2, 2), f, "alt_img_stem.4."))
    hw = s.shape[-2:]
    alt_s = F.interpolate(alt, hw, mode="bilinear")                                                  # :996-1000
    s = torch.cat((s, alt_s), dim=1)
    hi = _cam(s, f, "att_hi.0.")
    mid = _cam(_pool(s), f, "att_mid.1.")
    lo = _cam(_pool(mid), f, "att_lo.1.")
    mid = F.interpolate(lo, hw, mode="nearest")
    lo = F.interpolate(lo, hw, mode="nearest")
    att = hi + mid + lo
    att = F.conv2d(att, f["att_top.0.weight"], f["att_top.0.bias"], 1, 1)
    if att_divisor is not None:
        att = att / att_divisor
    att = torch.sigmoid(att)
    s = s * att.expand(s.shape)
    s = torch.cat((s, a), dim=1)
    for i in range(3):
        s = _cam(s, f, "steps.%d." % i)
    det = F.conv2d(s, f["steps.3.weight"], f["steps.3.bias"], 1, 1)
    return att, det


def rgb2lab(rgb):
    """scikit-image ``skimage.color.rgb2lab`` (illuminant D65, 2 degree observer) restated from its published
    algorithm - the package is not in the image (**parity unpinned**; call site rtpe/dataloaders.py:352-356).
    rgb: (..., 3) float in [0, 1] -> (..., 3) float64 (L in [0, 100])."""
    import numpy as np
    arr = np.asarray(rgb, np.float64).copy()
    mask = arr > 0.04045
    arr[mask] = np.power((arr[mask] + 0.055) / 1.055, 2.4)
    arr[~mask] /= 12.92
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    xyz = arr @ m.T
    xyz = xyz / np.array([0.95047, 1.0, 1.08883])
    mask = xyz > 0.008856
    xyz[mask] = np.cbrt(xyz[mask])
    xyz[~mask] = 7.787 * xyz[~mask] + 16.0 / 116.0
    x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
    return np.stack([116.0 * y - 16.0, 500.0 * (x - y), 200.0 * (y - z)], axis=-1)


def rgb2hsv(rgb):
    """``skimage.color.rgb2hsv`` restated (same status as rgb2lab): (..., 3) float in [0, 1] -> h, s, v in [0, 1]"""
    import numpy as np
    arr = np.asarray(rgb, np.float64)
    v = arr.max(-1)
    delta = arr.max(-1) - arr.min(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        s = np.where(delta == 0, 0.0, delta / v)
        r, g, b = arr[..., 0], arr[..., 1], arr[..., 2]
        h = np.where(v == r, (g - b) / delta, np.where(v == g, 2.0 + (b - r) / delta, 4.0 + (r - g) / delta))
        h = (h / 6.0) % 1.0
    h = np.where(delta == 0, 0.0, h)
    return np.stack([h, s, v], axis=-1)
