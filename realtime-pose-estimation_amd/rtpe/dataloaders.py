"""The one piece of the reference's ``rtpe/dataloaders.py`` that sits next to the accelerated path: the
alternative colour space of ``CocoDistillationDatasetAugmented2`` (:314-375), which feeds ``AttentionStudentSteps``
its ``alt`` input.  The reference calls ``skimage.color.rgb2lab`` / ``rgb2hsv`` on the ToTensor'd image on the CPU;
here one HIP pass (``rtpe_rgb_to_alt``) does it for a batch on the GPU.  The datasets themselves (COCO I/O,
pycocotools scoring) are out of scope."""
import ctypes

import torch

from . import _native as nat


def _convert(img, mode):
    t = img if torch.is_tensor(img) else torch.as_tensor(img)
    nat.require_gpu(t, "rgb2lab / rgb2hsv")
    if t.dtype != torch.float32 or t.dim() not in (3, 4) or t.shape[-3] != 3:
        raise TypeError("expected a float32 (3,H,W) or (N,3,H,W) RGB tensor in [0, 1]")
    x = t.contiguous().view((-1, 3) + tuple(t.shape[-2:]))
    out = torch.empty_like(x)
    with nat.on_device(x):
        nat.check(nat.lib().rtpe_rgb_to_alt(ctypes.c_void_p(x.data_ptr()), x.shape[0], x.shape[2], x.shape[3], mode,
                                            ctypes.c_void_p(out.data_ptr()), nat.stream_ptr(x.device)))
    return out.view(t.shape)


def rgb2lab(img):
    """(3,H,W) / (N,3,H,W) float32 RGB in [0,1] on the GPU -> CIE-LAB (D65), L in [0,100]; what the reference gets
    from ``skimage.color.rgb2lab`` + ``to_tensor`` (dataloaders.py:352-356)"""
    return _convert(img, 0)


def rgb2hsv(img):
    """-> HSV, all three in [0,1] (``skimage.color.rgb2hsv``)"""
    return _convert(img, 1)


def alt_colorspace(img, alt_colorspace="LAB"):
    """``colorspace_fn`` of CocoDistillationDatasetAugmented2 (:337-343)"""
    if alt_colorspace == "LAB":
        return rgb2lab(img)
    if alt_colorspace == "HSV":
        return rgb2hsv(img)
    raise NotImplementedError("Unknown color space {}".format(alt_colorspace))
