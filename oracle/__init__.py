"""CPU oracle for the HigherHRNet-w48 forward + heatmap->keypoint decode path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product (``realtime-pose-estimation_amd/rtpe``) never imports it and raises
if its HIP library is missing.

What is in here
---------------
* ``hrnet_ref``  - functional PyTorch-CPU restatement of the reference network
  (rtpe/third_party/pose_higher_hrnet.py) incl. the half wrapper of
  rtpe/third_party/fp16_utils/fp16util.py:40-91.
* ``decode_ref`` - numpy / torch-CPU restatement of
  rtpe/third_party/group.py:19-287 and of the two bilinear upsamples of
  validate_hhrnet.py:94-98.
* ``hungarian_ref`` - restatement of the Kuhn-Munkres assignment that
  group.py:19-23 obtains from the third-party PyPI package ``munkres``
  (NOT vendored in the reference, no version pin anywhere in it, not installed
  in this image).
* ``synth`` - the seeded synthetic weights / images / decode maps of
  SURVEY.md section 8(d).

Parity pin
----------
The restatement is pinned against outputs of the reference itself, imported
from /root/reference in the build container by ``tools/gen_golden.py``; the
vectors live in ``tests/golden/*.npz`` (``tests/test_oracle_golden.py``).
The one unpinned spot is the Hungarian tie-break: the reference's ``munkres``
dependency cannot be run here, so the fixtures were produced with an optimal
assignment from scipy in its place ("parity unpinned" for equal-cost
assignments only).
"""
