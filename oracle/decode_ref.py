"""numpy / torch-CPU restatement of the heatmap -> keypoint decode (oracle).

Test infrastructure (see oracle/__init__.py).  Follows
/root/reference/rtpe/third_party/group.py (``HeatmapParser`` :125-287,
``match_by_tag`` :26-97, ``Params`` :100-110) and the two bilinear upsamples
of /root/reference/validate_hhrnet.py:94-98.  Written to make every rounding
and ordering decision explicit, because the HIP kernels and the C++ host
matcher have to reproduce them bit for bit:

* bilinear, align_corners=True, fp32, *as PyTorch-CPU evaluates it*:
  ``T = fma(v0, lx0, v1*lx1)`` per row, ``out = fma(T0, ly0, T1*ly1)``,
  ``l1 = clamp(scale*dst - i0, 0, 1)``, ``l0 = 1 - l1``,
  ``scale = float32(in-1) / float32(out-1)`` (found by experiment against
  ``F.interpolate`` in the build container, see ``bilinear_explicit``).
* numpy float32 means: 8-way pairwise over a contiguous 1-D run
  (``np.mean`` of (n,1) tags, ``ans[:,2].mean()``), plain sequential when the
  reduced axis is not the only one (D > 1 tags).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

from .hungarian_ref import munkres_compute

f32 = np.float32


class Params:
    """group.py:100-110 (joint_order is the identity, :110)"""

    def __init__(self, num_joints=17, max_num_people=30, detection_threshold=0.1,
                 tag_threshold=1.0, use_detection_val=True, ignore_too_much=False):
        self.num_joints = num_joints
        self.max_num_people = max_num_people
        self.detection_threshold = detection_threshold
        self.tag_threshold = tag_threshold
        self.use_detection_val = use_detection_val
        self.ignore_too_much = ignore_too_much


# --------------------------------------------------------------------------- #
# bilinear upsample (validate_hhrnet.py:94-98)
# --------------------------------------------------------------------------- #
def upsample_bilinear(x, h, w):
    """the stock op the reference calls"""
    return F.interpolate(x, (h, w), mode="bilinear", align_corners=True)


def _fma32(a, b, c):
    # exact product and sum in f64 (24+24 bit products are exact; the sum of an
    # exact product and an f32 rounds once when cast back) == fused multiply-add
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def bilinear_axis(n_in, n_out):
    """source indices and weights of one axis (align_corners=True)"""
    if n_in == n_out:
        i0 = np.arange(n_out)
        return i0, i0.copy(), np.ones(n_out, f32), np.zeros(n_out, f32)
    scale = f32(n_in - 1) / f32(n_out - 1) if n_out > 1 else f32(0)
    real = scale * np.arange(n_out).astype(f32)
    i0 = np.minimum(real.astype(np.int64), n_in - 1)
    i1 = i0 + (i0 < n_in - 1)
    l1 = np.clip(real - i0.astype(f32), 0, 1).astype(f32)
    return i0, i1, (f32(1) - l1).astype(f32), l1


def bilinear_explicit(x, h, w):
    """explicit formula, bit-equal to ``upsample_bilinear`` on the CPU
    (tests/test_oracle_golden.py::test_bilinear_formula)"""
    x = np.asarray(x, f32)
    y0, y1, ly0, ly1 = bilinear_axis(x.shape[-2], h)
    x0, x1, lx0, lx1 = bilinear_axis(x.shape[-1], w)
    r0, r1 = x[..., y0, :], x[..., y1, :]
    t0 = _fma32(r0[..., x0], lx0, r0[..., x1] * lx1)
    t1 = _fma32(r1[..., x0], lx0, r1[..., x1] * lx1)
    return _fma32(t0, ly0[:, None], t1 * ly1[:, None])


# --------------------------------------------------------------------------- #
# numpy reduction orders
# --------------------------------------------------------------------------- #
def pairwise8_sum_f32(a):
    """numpy's contiguous float32 add.reduce for n <= 128"""
    a = np.asarray(a, f32)
    n = a.shape[0]
    if n < 8:
        s = f32(0)
        for v in a:
            s = f32(s + v)
        return s
    r = [f32(a[i]) for i in range(8)]
    i = 8
    while i + 8 <= n:
        for j in range(8):
            r[j] = f32(r[j] + a[i + j])
        i += 8
    s = f32(f32(f32(r[0] + r[1]) + f32(r[2] + r[3])) + f32(f32(r[4] + r[5]) + f32(r[6] + r[7])))
    while i < n:
        s = f32(s + a[i])
        i += 1
    return s


# --------------------------------------------------------------------------- #
# HeatmapParser
# --------------------------------------------------------------------------- #
class HeatmapParserRef:
    def __init__(self, num_joints=17, max_num_people=30, detection_threshold=0.1,
                 tag_threshold=1.0, use_detection_val=True, ignore_too_much=False,
                 tag_per_joint=True, nms_ksize=5, nms_padding=2):
        self.params = Params(num_joints, max_num_people, detection_threshold,
                             tag_threshold, use_detection_val, ignore_too_much)
        self.tag_per_joint = tag_per_joint
        self.nms_ksize, self.nms_padding = nms_ksize, nms_padding

    def nms(self, det):
        """group.py:134-138: keep a pixel iff it equals its 5x5 window max"""
        m = F.max_pool2d(det, self.nms_ksize, 1, self.nms_padding)
        return det * torch.eq(m, det).float()

    def top_k(self, det, tag):
        """group.py:144-179.  det (N,J,h,w); tag (N,J,h,w,D) (or (N,1,h,w,D) if
        not tag_per_joint).  Order among equal values is whatever ATen gives."""
        det = self.nms(det)
        N, J, h, w = det.shape
        val_k, ind = det.reshape(N, J, -1).topk(self.params.max_num_people, dim=2)
        tag = tag.reshape(tag.shape[0], tag.shape[1], w * h, -1)
        if not self.tag_per_joint:
            tag = tag.expand(-1, self.params.num_joints, -1, -1)
        tag_k = torch.stack([torch.gather(tag[..., d], 2, ind) for d in range(tag.shape[3])], 3)
        x = ind % w
        y = (ind / w).long()          # true division then truncation, :168-169
        return {"tag_k": tag_k.numpy(), "loc_k": torch.stack((x, y), 3).numpy(),
                "val_k": val_k.numpy()}

    def match(self, tag_k, loc_k, val_k):
        return [match_by_tag(t, l, v, self.params) for t, l, v in zip(tag_k, loc_k, val_k)]

    def adjust(self, ans, det):
        """group.py:181-200 quarter-pixel shift towards the larger neighbour;
        columns are compared first, then rows; ties and borders go to -0.25"""
        det = np.asarray(det)
        for b, people in enumerate(ans):
            for p in range(len(people)):
                for j in range(people.shape[1]):
                    if people[p, j, 2] > 0:
                        cx, cy = people[p, j, 0], people[p, j, 1]       # float32
                        col, row = int(cx), int(cy)
                        m = det[b, j]
                        H, W = m.shape
                        cx = cx + f32(0.25) if m[row, min(col + 1, W - 1)] > m[row, max(col - 1, 0)] \
                            else cx - f32(0.25)
                        cy = cy + f32(0.25) if m[min(row + 1, H - 1), col] > m[max(row - 1, 0), col] \
                            else cy - f32(0.25)
                        people[p, j, 0] = cx + f32(0.5)
                        people[p, j, 1] = cy + f32(0.5)
        return ans

    def refine(self, det, tag, kp):
        """group.py:202-264 for one person.  det (J,h,w) f32, tag (J,h,w,D) f32,
        kp (J,3+D) f32 (modified in place and returned)."""
        if tag.ndim == 3:
            tag = tag[..., None]
        J, h, w = det.shape
        D = tag.shape[3]
        seen = [tag[j, int(kp[j, 1]), int(kp[j, 0])] for j in range(J) if kp[j, 2] > 0]
        seen = np.asarray(seen, f32)                       # (n, D)
        if D == 1:
            mean = np.array([pairwise8_sum_f32(seen[:, 0]) / f32(len(seen))], f32)
        else:
            s = np.zeros(D, f32)
            for t in seen:
                s = (s + t).astype(f32)
            mean = (s / f32(len(seen))).astype(f32)
        for j in range(J):
            d = tag[j] - mean[None, None, :]
            sq = d * d
            if D < 8:
                ss = sq[..., 0].copy()
                for k in range(1, D):
                    ss = ss + sq[..., k]
            else:
                ss = sq.sum(axis=2)                        # numpy pairwise, as the reference
            score = det[j] - np.rint(np.sqrt(ss))          # np.round == half-to-even
            flat = int(np.argmax(score))                   # first maximum
            y, x = divmod(flat, w)
            val = det[j, y, x]
            fx = x + 0.5 + (0.25 if det[j, y, min(x + 1, w - 1)] > det[j, y, max(x - 1, 0)] else -0.25)
            fy = y + 0.5 + (0.25 if det[j, min(y + 1, h - 1), x] > det[j, max(y - 1, 0), x] else -0.25)
            if val > 0 and kp[j, 2] == 0:                  # fill only undetected joints
                kp[j, 0], kp[j, 1], kp[j, 2] = fx, fy, val
        return kp

    def parse(self, det, tag, adjust=True, refine=True):
        """group.py:266-287.  det (N,J,h,w) torch f32; tag (N,J,h,w,D)."""
        ans = self.match(**self.top_k(det, tag))
        det_np = det.numpy()
        if adjust:
            ans = self.adjust(ans, det_np)
        scores = [pairwise8_sum_f32(np.ascontiguousarray(p[:, 2])) / f32(len(p[:, 2]))
                  for p in ans[0]]
        if refine:
            people = ans[0]
            tag_np = tag[0].numpy()
            if not self.tag_per_joint:
                tag_np = np.tile(tag_np, (self.params.num_joints, 1, 1, 1))
            for p in range(len(people)):
                people[p] = self.refine(det_np[0], tag_np, people[p])
            ans = [people]
        return ans, scores


def match_by_tag(tag_k, loc_k, val_k, P):
    """group.py:26-97 for one image.  tag_k (J,K,D) f32, loc_k (J,K,2) i64,
    val_k (J,K) f32 -> (P, J, 3+D) float32 (shape (0,) if nobody found)."""
    J, D = P.num_joints, tag_k.shape[2]
    people = OrderedDict()      # key = float value of tag[0] -> [rows (J,3+D) f64, [tags f32]]

    def put(key, j, row, t):
        # dict.setdefault semantics of :51-52 / :91-94: an existing key keeps
        # its rows but its tag list is reset
        if key not in people:
            people[key] = [np.zeros((J, 3 + D)), None]
        people[key][0][j] = row
        people[key][1] = [t]

    for j in range(J):
        rows = np.concatenate((loc_k[j].astype(np.float64),
                               val_k[j][:, None].astype(np.float64),
                               tag_k[j].astype(np.float64)), 1)
        keep = rows[:, 2] > P.detection_threshold
        rows, tags = rows[keep], tag_k[j][keep]
        if not len(rows):
            continue
        if j == 0 or not people:
            for t, r in zip(tags, rows):
                put(t[0], j, r, t)
            continue
        keys = list(people)[:P.max_num_people]
        if P.ignore_too_much and len(keys) == P.max_num_people:
            continue
        centres = np.stack([_mean_tags_f32(people[k][1]) for k in keys])     # (G, D) f32
        diff = rows[:, None, 3:] - centres[None].astype(np.float64)
        dist = np.sqrt(_sum_last_f64(diff * diff))                           # (A, G)
        cost = np.rint(dist) * 100 - rows[:, 2:3] if P.use_detection_val else dist.copy()
        A, G = dist.shape
        if A > G:
            cost = np.concatenate((cost, np.full((A, A - G), 1e10)), 1)
        for r, c in munkres_compute(cost):
            if r < A and c < G and dist[r, c] < P.tag_threshold:
                people[keys[c]][0][j] = rows[r]
                people[keys[c]][1].append(tags[r])
            else:
                put(tags[r][0], j, rows[r], tags[r])
    return np.array([v[0] for v in people.values()]).astype(f32)


def _mean_tags_f32(tags):
    """np.mean(list of (D,) f32, axis=0), group.py:55"""
    a = np.asarray(tags, f32)
    n = f32(len(a))
    if a.shape[1] == 1:
        return np.array([pairwise8_sum_f32(a[:, 0]) / n], f32)
    s = np.zeros(a.shape[1], f32)
    for t in a:
        s = (s + t).astype(f32)
    return (s / n).astype(f32)


def _sum_last_f64(a):
    """add.reduce over the contiguous last axis, float64 (np.linalg.norm :62)"""
    D = a.shape[-1]
    if D < 8:
        s = a[..., 0].copy()
        for k in range(1, D):
            s = s + a[..., k]
        return s
    return a.sum(axis=-1)
