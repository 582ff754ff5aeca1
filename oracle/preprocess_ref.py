"""numpy restatement of the pre-processing step (oracle; test infrastructure only).

Geometry follows /root/reference/rtpe/third_party/transforms.py: ``get_multi_scale_size``
:155-176 and ``get_affine_transform`` :59-93 for rot = 0, where the three point pairs reduce to a
uniform scale about the image centre.  The image warp restates the convention of
csrc/preprocess.hip (fp32 bilinear weights, zero border, pixel centres at integers), NOT
cv2.warpAffine's fixed-point interpolation: cv2 is not installed, so that part is **parity
unpinned**  and the expected deviation is about one grey level.  What follows the warp is
the published arithmetic of the reference's data flow: the warp returns a uint8 image
(transforms.py:185-190; ``round_u8``), torchvision's ToTensor divides by 255 and Normalize computes
``(x - mean) / std`` with a true division (validate_hhrnet.py:63-67).
"""
import numpy as np


def multi_scale_size(h, w, input_size, current_scale=1, min_scale=1):
    """transforms.py:155-176 -> ((w_resized, h_resized), center, scale)"""
    center = np.array([int(w / 2.0 + 0.5), int(h / 2.0 + 0.5)])
    mis = int((min_scale * input_size + 63) // 64 * 64)
    if w < h:
        wr = int(mis * current_scale / min_scale)
        hr = int(int((mis / w * h + 63) // 64 * 64) * current_scale / min_scale)
        sw, sh = w / 200.0, hr / wr * w / 200.0
    else:
        hr = int(mis * current_scale / min_scale)
        wr = int(int((mis / h * w + 63) // 64 * 64) * current_scale / min_scale)
        sh, sw = h / 200.0, wr / hr * h / 200.0
    return (wr, hr), center, np.array([sw, sh])


def dst_to_src_matrix(center, scale, size_resized):
    """closed form of get_affine_transform(..., inv=1) for rot = 0 (:59-93): the point triples are
    (c, c + (0, -sw*100), perpendicular) -> (d, d + (0, -dw/2), perpendicular), i.e. an isotropic
    scale k = (sw * 200) / dw about the centres"""
    dw, dh = size_resized
    k = scale[0] * 200.0 / dw
    return np.array([[k, 0.0, center[0] - k * dw * 0.5], [0.0, k, center[1] - k * dh * 0.5]])


def warp_normalize(img, input_size, mean, std, current_scale=1, min_scale=1, round_u8=True):
    """(h, w, 3) uint8 -> (3, H, W) float32, same arithmetic order as the kernel"""
    h, w = img.shape[:2]
    (ow, oh), center, scale = multi_scale_size(h, w, input_size, current_scale, min_scale)
    m = dst_to_src_matrix(center, scale, (ow, oh)).astype(np.float32)
    f = np.float32
    xs = np.arange(ow, dtype=np.float32)[None, :]
    ys = np.arange(oh, dtype=np.float32)[:, None]

    def fma(a, b, c):       # float32 fused multiply-add through float64 (exact product, one rounding)
        return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)

    sx = fma(np.broadcast_to(m[0, 0], (oh, ow)), np.broadcast_to(xs, (oh, ow)),
             fma(np.broadcast_to(m[0, 1], (oh, ow)), np.broadcast_to(ys, (oh, ow)), np.broadcast_to(m[0, 2], (oh, ow))))
    sy = fma(np.broadcast_to(m[1, 0], (oh, ow)), np.broadcast_to(xs, (oh, ow)),
             fma(np.broadcast_to(m[1, 1], (oh, ow)), np.broadcast_to(ys, (oh, ow)), np.broadcast_to(m[1, 2], (oh, ow))))
    fx, fy = np.floor(sx), np.floor(sy)
    lx, ly = (sx - fx).astype(f), (sy - fy).astype(f)
    x0, y0 = fx.astype(np.int64), fy.astype(np.int64)
    acc = np.zeros((oh, ow, 3), np.float32)
    for dy in (0, 1):
        for dx in (0, 1):
            xx, yy = x0 + dx, y0 + dy
            wgt = ((lx if dx else f(1) - lx) * (ly if dy else f(1) - ly)).astype(f)
            ok = (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
            px = img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.float32)
            upd = fma(wgt[..., None], px, acc)
            acc = np.where(ok[..., None], upd, acc)
    if round_u8:
        acc = np.clip(np.floor(acc + f(0.5)), f(0), f(255)).astype(f)
    out = ((acc / f(255.0) - np.asarray(mean, f)) / np.asarray(std, f)).astype(f)
    return out.transpose(2, 0, 1).astype(f), center, scale
