"""Model factory of the hot path (drop-in for ``rtpe/helpers.py:32-73`` of the
reference; the logging / plotting helpers of that file are out of scope)."""
import torch

from .third_party.fp16_utils.fp16util import network_to_half
from .third_party.pose_higher_hrnet import PoseHigherResolutionNet

W48_KWARGS = dict(
    num_joints=17, tag_per_joint=True, final_conv_ksize=1, pretrained_layers=["*"], inplanes=64,
    s2_modules=1, s2_branches=2, s2_block_type="BASIC", s2_blocks=[4, 4], s2_chans=[48, 96],
    s3_modules=4, s3_branches=3, s3_block_type="BASIC", s3_blocks=[4, 4, 4], s3_chans=[48, 96, 192],
    s4_modules=3, s4_branches=4, s4_block_type="BASIC", s4_blocks=[4, 4, 4, 4],
    s4_chans=[48, 96, 192, 384],
    deconvs=1, deconv_chans=[48], deconv_ksize=[4], deconv_num_blocks=4, deconv_cat=[True],
    with_ae_loss=(True, False))


def build_hrnet_w48_teacher(state_dict=None):
    """HigherHRNet-w48 under the half wrapper; ``state_dict`` uses the
    checkpoint's ``1.``-prefixed keys (the wrapper is a Sequential)."""
    model = network_to_half(PoseHigherResolutionNet(**W48_KWARGS))
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model


def get_hrnet_w48_teacher(w48_statedict_path):
    """Instantiate HigherHRNet_w48, strictly load the upstream state dict and
    return it in eval mode (reference helpers.py:32-73).  Call ``.to("cuda")``
    on the result as the reference's scripts do."""
    return build_hrnet_w48_teacher(torch.load(w48_statedict_path, map_location="cpu"))
