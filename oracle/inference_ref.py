"""Torch-CPU restatement of the multi-scale / flip test aggregation (oracle; test infrastructure only).

The functions restated here - ``get_multi_stage_outputs`` and ``aggregate_results`` - belong to the upstream HigherHRNet
code base (HRNet/HigherHRNet-Human-Pose-Estimation, ``lib/core/inference.py``), which the reference imports in
``legacy/valid_ae1dim.py:36`` / ``legacy/valid_ae_avg.py`` but does NOT contain (the legacy scripts "import
upstream modules not present in this repo"), so there is nothing in /root/reference to run or to take vectors from:
**parity unpinned** against upstream; anchored on the reference's call sites (valid_ae1dim.py:166-207) and on the
published algorithm: every tensor op below is the stock torch op upstream calls (``F.interpolate(mode="bilinear",
align_corners=False)``, ``torch.flip``, fancy channel indexing, ``+``, ``/``), so the product's HIP kernel is checked
bit for bit against these ops on the same inputs.
"""
import torch
import torch.nn.functional as F

FLIP_COCO = [0, 2, 1, 4, 3, 6, 5, 8, 7, 10, 9, 12, 11, 14, 13, 16, 15]


def get_multi_stage_outputs(model, image, with_flip=False, project2image=False, size_projected=None, num_joints=17,
                            with_heatmaps=(True, True), with_heatmaps_loss=(True, True), with_ae=(True, False),
                            with_ae_loss=(True, False), tag_per_joint=True, flip_index=FLIP_COCO):
    heatmaps_avg, num_heatmaps, heatmaps, tags = 0, 0, [], []
    outputs = list(model(image))
    for i, output in enumerate(outputs):
        if len(outputs) > 1 and i != len(outputs) - 1:
            output = F.interpolate(output, size=(outputs[-1].size(2), outputs[-1].size(3)), mode="bilinear",
                                   align_corners=False)
        offset_feat = num_joints if with_heatmaps_loss[i] else 0
        if with_heatmaps_loss[i] and with_heatmaps[i]:
            heatmaps_avg += output[:, :num_joints]
            num_heatmaps += 1
        if with_ae_loss[i] and with_ae[i]:
            tags.append(output[:, offset_feat:])
    if num_heatmaps > 0:
        heatmaps.append(heatmaps_avg / num_heatmaps)
    if with_flip:
        heatmaps_avg, num_heatmaps = 0, 0
        outputs_flip = list(model(torch.flip(image, [3])))
        for i in range(len(outputs_flip)):
            output = outputs_flip[i]
            if len(outputs_flip) > 1 and i != len(outputs_flip) - 1:
                output = F.interpolate(output, size=(outputs_flip[-1].size(2), outputs_flip[-1].size(3)), mode="bilinear",
                                       align_corners=False)
            output = torch.flip(output, [3])
            outputs.append(output)
            offset_feat = num_joints if with_heatmaps_loss[i] else 0
            if with_heatmaps_loss[i] and with_heatmaps[i]:
                heatmaps_avg += output[:, :num_joints][:, flip_index, :, :]
                num_heatmaps += 1
            if with_ae_loss[i] and with_ae[i]:
                tags.append(output[:, offset_feat:])
                if tag_per_joint:
                    tags[-1] = tags[-1][:, flip_index, :, :]
        heatmaps.append(heatmaps_avg / num_heatmaps)
    if project2image and size_projected:
        heatmaps = [F.interpolate(h, size=(size_projected[1], size_projected[0]), mode="bilinear", align_corners=False)
                    for h in heatmaps]
        tags = [F.interpolate(t, size=(size_projected[1], size_projected[0]), mode="bilinear", align_corners=False)
                for t in tags]
    return outputs, heatmaps, tags


def aggregate_results(scale_factor, final_heatmaps, tags_list, heatmaps, tags, scale_factors=(1,), flip_test=True,
                      project2image=True):
    if scale_factor == 1 or len(scale_factors) == 1:
        if final_heatmaps is not None and not project2image:
            tags = [F.interpolate(t, size=(final_heatmaps.size(2), final_heatmaps.size(3)), mode="bilinear",
                                  align_corners=False) for t in tags]
        for t in tags:
            tags_list.append(torch.unsqueeze(t, dim=4))
    heatmaps_avg = (heatmaps[0] + heatmaps[1]) / 2.0 if flip_test else heatmaps[0]
    if final_heatmaps is None:
        final_heatmaps = heatmaps_avg
    elif project2image:
        final_heatmaps += heatmaps_avg
    else:
        final_heatmaps += F.interpolate(heatmaps_avg, size=(final_heatmaps.size(2), final_heatmaps.size(3)),
                                        mode="bilinear", align_corners=False)
    return final_heatmaps, tags_list


def multi_scale_maps(model, inputs, scale_factors, base_size, flip_test=True, project2image=True):
    """the tensor part of legacy/valid_ae1dim.py:166-195 for pre-warped network inputs ``inputs[s]`` (one per scale)"""
    final_heatmaps, tags_list = None, []
    for s in sorted(scale_factors, reverse=True):
        _, heatmaps, tags = get_multi_stage_outputs(model, inputs[s], flip_test, project2image, base_size)
        final_heatmaps, tags_list = aggregate_results(s, final_heatmaps, tags_list, heatmaps, tags, scale_factors,
                                                      flip_test, project2image)
    final_heatmaps = final_heatmaps / float(len(scale_factors))
    return final_heatmaps, torch.cat(tags_list, dim=4)
