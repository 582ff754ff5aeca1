"""Multi-scale / flip test inference.

The reference's upstream-faithful validation scripts (``legacy/valid_ae1dim.py:166-207``, ``legacy/valid_ae_avg.py``)
call ``get_multi_stage_outputs`` and ``aggregate_results`` of the upstream HigherHRNet code base
(``lib/core/inference.py`` of HRNet/HigherHRNet-Human-Pose-Estimation).  That module is NOT in the reference
repository; its published algorithm is restated here behind the same two function names, with the ``cfg`` object
replaced by keyword arguments that carry the ``cfg.TEST.* / cfg.DATASET.* / cfg.LOSS.*`` fields it reads.  All tensor
arithmetic runs in one HIP kernel (``rtpe_resize_combine``: ``dst = [dst +] resize(flip(src[:, perm])) [/ div]``,
bit-equal to ``F.interpolate(mode="bilinear", align_corners=False)`` / ``torch.flip`` / index / add / divide on the
CPU); PyTorch only allocates.  ``multi_scale_inference`` is the per-image body of valid_ae1dim.py:166-207 on top of
the accelerated teacher, warp and decode of this package.
"""
import ctypes

import torch

from . import _native as nat

# upstream dataset/transforms FLIP_CONFIG["COCO"]: left <-> right joints
FLIP_CONFIG = {"COCO": [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15],
               "COCO_WITH_CENTER": [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15, 17]}


def resize_combine(src, size, channels=None, flip=False, out=None, accumulate=False, div=1.0):
    """``out = [out +] interpolate(flip(src[:, channels]), size, bilinear, align_corners=False) [/ div]`` in one
    pass.  src (N,C,h,w) float32 on the GPU; ``channels``: list of source channels (default all); ``flip``:
    ``torch.flip(., [3])`` AFTER the resize (upstream order).  Returns ``out`` (allocated when None)."""
    nat.require_gpu(src, "resize_combine")
    if src.dtype != torch.float32 or src.dim() != 4:
        raise TypeError("resize_combine expects a float32 (N,C,h,w) tensor")
    src = src.contiguous()
    N, C, h, w = src.shape
    chans = list(range(C)) if channels is None else [int(c) for c in channels]
    oh, ow = int(size[0]), int(size[1])
    if out is None:
        if accumulate:
            raise ValueError("resize_combine: accumulate needs an existing output")
        out = torch.empty((N, len(chans), oh, ow), dtype=torch.float32, device=src.device)
    if tuple(out.shape) != (N, len(chans), oh, ow) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("resize_combine: output must be a contiguous float32 %s tensor" % ((N, len(chans), oh, ow),))
    nat.same_device("resize_combine", src, out)
    cmap = (ctypes.c_int32 * len(chans))(*chans)
    with nat.on_device(src):
        nat.check(nat.lib().rtpe_resize_combine(
            ctypes.c_void_p(src.data_ptr()), N, C, h, w, cmap, len(chans), int(bool(flip)),
            ctypes.c_void_p(out.data_ptr()), oh, ow, int(bool(accumulate)), float(div), nat.stream_ptr(src.device)))
    return out


def get_multi_stage_outputs(model, image, with_flip=False, project2image=False, size_projected=None,
                            num_joints=17, with_heatmaps=(True, True), with_heatmaps_loss=(True, True),
                            with_ae=(True, False), with_ae_loss=(True, False), tag_per_joint=True,
                            flip_index=None):
    """upstream ``get_multi_stage_outputs(cfg, model, image, with_flip, project2image, size_projected)``:
    ``model(image) -> [preds (N,34,h/4,w/4), refined (N,17,h/2,w/2)]``; every output but the last is resized to the
    last one's size (bilinear, align_corners=False); the heat maps of all stages are averaged, the tag channels
    collected; with ``with_flip`` the same for the mirrored image, mirrored back and with left / right joints
    swapped; with ``project2image`` everything is resized to ``size_projected = (w, h)``.
    Returns ``(outputs, heatmaps, tags)`` as upstream: ``heatmaps`` = [avg] or [avg, avg_flipped], ``tags`` = list of
    (N,17,h,w) maps."""
    flip_index = FLIP_CONFIG["COCO"] if flip_index is None else list(flip_index)
    joints = list(range(num_joints))
    outputs = list(model(image))
    heatmaps, tags = [], []

    def stage_maps(outs, mirrored):
        size = tuple(outs[-1].shape[2:])
        n_hm = sum(1 for i in range(len(outs)) if with_heatmaps_loss[i] and with_heatmaps[i])
        avg, seen = None, 0
        for i, o in enumerate(outs):
            o = o.float()
            offset = num_joints if with_heatmaps_loss[i] else 0
            if with_heatmaps_loss[i] and with_heatmaps[i]:
                seen += 1
                src = [flip_index[j] for j in joints] if mirrored else joints
                # heatmaps_avg += output[:, :J][:, flip_index]; the division by the count rides on the last term
                avg = resize_combine(o, size, src, flip=mirrored, out=avg, accumulate=avg is not None,
                                     div=float(n_hm) if seen == n_hm else 1.0)
            if with_ae_loss[i] and with_ae[i]:
                n_tag = o.shape[1] - offset
                src = [offset + (flip_index[j] if (mirrored and tag_per_joint) else j) for j in range(n_tag)]
                tags.append(resize_combine(o, size, src, flip=mirrored))
        if avg is not None:
            heatmaps.append(avg)

    stage_maps(outputs, False)
    if with_flip:
        outputs_flip = list(model(torch.flip(image, [3])))
        stage_maps(outputs_flip, True)
        outputs = outputs + outputs_flip        # upstream appends the mirrored-back outputs; callers ignore them
    if project2image and size_projected:
        size = (int(size_projected[1]), int(size_projected[0]))
        heatmaps = [resize_combine(h, size) for h in heatmaps]
        tags = [resize_combine(t, size) for t in tags]
    return outputs, heatmaps, tags


def aggregate_results(scale_factor, final_heatmaps, tags_list, heatmaps, tags, scale_factors=(1,),
                      flip_test=True, project2image=True):
    """upstream ``aggregate_results(cfg, scale_factor, final_heatmaps, tags_list, heatmaps, tags)``: tags are kept
    for scale 1 only, the (flip-averaged) heat maps of every scale are summed at the size of the first scale"""
    if scale_factor == 1 or len(scale_factors) == 1:
        if final_heatmaps is not None and not project2image:
            tags = [resize_combine(t, final_heatmaps.shape[2:]) for t in tags]
        for t in tags:
            tags_list.append(t.unsqueeze(4))
    if flip_test:
        avg = resize_combine(heatmaps[0], heatmaps[0].shape[2:])                       # a copy
        avg = resize_combine(heatmaps[1], avg.shape[2:], out=avg, accumulate=True, div=2.0)
    else:
        avg = heatmaps[0]
    if final_heatmaps is None:
        final_heatmaps = avg
    else:       # `+=` (PROJECT2IMAGE: same size) or `+= interpolate(avg, size of final)`: the same kernel call
        resize_combine(avg, final_heatmaps.shape[2:], out=final_heatmaps, accumulate=True)
    return final_heatmaps, tags_list


def multi_scale_inference(model, parser, image, input_size=640, scale_factors=(1,), flip_test=True,
                          project2image=True, adjust=True, refine=True, device="cuda", ags=False, **stage_kw):
    """The per-image body of legacy/valid_ae1dim.py:166-207: for every test scale (largest first) warp the image,
    run the teacher (and its mirror image), aggregate; average the heat maps over the scales, concatenate the tag
    maps, group with ``parser.parse`` and map the keypoints back to image coordinates with ``get_final_preds``.
    image: (h, w, 3) uint8.  Returns ``(final_results, scores, final_heatmaps, tags)``.

    ``ags=True`` takes the branch the reference script actually runs (valid_ae1dim.py:177, :191-199, ``AGS = True``):
    the grouping sees ONE tag map for all joints - channel 0 of the first (un-mirrored) tag map of the LAST scale of
    the loop, a (1,1,h,w,1) tensor - with ``parser.tag_per_joint = False`` (the script sets the attribute on the
    caller's parser and leaves it set; so does this) and ``adjust = refine = True`` whatever the arguments say.
    ``tags`` in the result is then that tensor."""
    from .third_party import transforms
    scale_factors = list(scale_factors)
    base_size, center, scale = transforms.get_multi_scale_size(image, input_size, 1.0, min(scale_factors))
    final_heatmaps, tags_list = None, []
    with torch.no_grad():
        for s in sorted(scale_factors, reverse=True):
            t, center, scale = transforms.warp_normalize(image, input_size, s, min(scale_factors), device=device)
            _, heatmaps, tags = get_multi_stage_outputs(model, t, flip_test, project2image, base_size, **stage_kw)
            ags_map = tags[0][:, 0].unsqueeze(-1).unsqueeze(0)          # valid_ae1dim.py:177
            final_heatmaps, tags_list = aggregate_results(s, final_heatmaps, tags_list, heatmaps, tags, scale_factors,
                                                          flip_test, project2image)
        if len(scale_factors) != 1:
            final_heatmaps = resize_combine(final_heatmaps, final_heatmaps.shape[2:], div=float(len(scale_factors)))
        tags = torch.cat(tags_list, dim=4)
        if ags:
            parser.tag_per_joint = False                                # valid_ae1dim.py:196
            tags = ags_map.contiguous()
            grouped, scores = parser.parse(final_heatmaps, tags, True, True)
        else:
            grouped, scores = parser.parse(final_heatmaps, tags, adjust, refine)
    final_results = transforms.get_final_preds(grouped, center, scale,
                                               [final_heatmaps.size(3), final_heatmaps.size(2)])
    return final_results, scores, final_heatmaps, tags
