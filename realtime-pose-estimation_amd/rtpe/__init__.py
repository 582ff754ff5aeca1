"""MI355X-native inference path for the HigherHRNet-w48 teacher of
andres-fr/realtime-pose-estimation (forward pass + heatmap->keypoint decode).

Same import paths as the reference's ``rtpe`` package for the hot path:
``rtpe.helpers.get_hrnet_w48_teacher``, ``rtpe.engine.eval_student``,
``rtpe.third_party.pose_higher_hrnet``, ``rtpe.third_party.group``,
``rtpe.third_party.fp16_utils.fp16util``.
"""
