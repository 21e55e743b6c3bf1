"""CPU oracle for the 3D-box keypoint-regression hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`3d-object-detection.pytorch_amd/`) imports this directory; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and
only as the checker / the reported CPU baseline.

The oracle restates, in plain torch-CPU fp32 functional ops (numpy fp64 for the
geometry), the arithmetic of the reference `sovrasov/3d-object-detection.pytorch`
on the path SURVEY.md section 8 names.  Every function cites the reference
file:line it follows.

Pinning status
  * mobilenetv3_large / mobilenetv3_small + ModelWrapper, all losses, LossManager
    (incl. ALWA), ADD/SADD/accuracy, lift_2d:  PINNED -- checked against
    golden vectors in tests/golden/*.npz that `oracle/gen_golden.py` produced
    by importing the real reference from /root/reference (third-party imports
    stubbed) in the build container.
  * mobilenetv2 backbone: the reference has no MobileNetV2 regression model
     -- PARITY UNPINNED w.r.t. the reference; it is the
    standard Sandler et al. (t,c,n,s) table run through the same (pinned) block
    arithmetic.
  * 3-D IoU (`objectron.dataset.box/iou`): third-party submodule absent from
    /root/reference and un-pinned -- PARITY UNPINNED numerically; restated from
    the published algorithm and pinned only by analytic known answers and the
    reference's own threshold test (tests/test_geometry.py:31-40).
"""
