"""Stand-in for the ``import smplx`` at the top of the reference's demos.py (:3).  demo_full (demos.py:10-45, 296-331) creates an
SMPL body model -- a licensed asset this image does not have -- writes a template, a pose and an orientation into it and reads
posed vertices back.  By default the stand-in refuses (SMPL posing is outside the hot path); with CAPE_STUB_SMPL=identity
(tests/test_reference_entry_script.py, the main.py run) ``create`` returns a model whose "posing" is the identity: exactly the
attributes and the call demo_full touches, so that the reference's train / test pipeline can be executed to its end.
TEST INFRASTRUCTURE ONLY."""
import os


class _Posed(object):
    def __init__(self, vertices):
        self.vertices = vertices


class _IdentitySMPL(object):
    def __init__(self):
        import numpy as np
        import torch
        import trimesh
        here = os.getcwd()                               # the scripts run from the checkout root (data/template_mesh.obj)
        m = trimesh.load(os.path.join(here, "data", "template_mesh.obj"), process=False)
        self.faces = np.asarray(m.faces)
        self.v_template = torch.zeros((len(m.vertices), 3), dtype=torch.float64)
        self.body_pose = torch.zeros((1, 69), dtype=torch.float64)
        self.global_orient = torch.zeros((1, 3), dtype=torch.float64)

    def __call__(self):
        return _Posed(self.v_template.clone()[None])


class body_models(object):
    @staticmethod
    def create(*a, **kw):
        if os.environ.get("CAPE_STUB_SMPL") == "identity":
            return _IdentitySMPL()
        raise NotImplementedError("smplx stand-in: SMPL posing (demos.py demo_full) is outside the hot path")
