"""Stand-in for the ``import smplx`` at the top of the reference's demos.py (:3); only demo_full (SMPL posing, out of scope)
calls into it.  TEST INFRASTRUCTURE ONLY."""


class body_models(object):
    @staticmethod
    def create(*a, **kw):
        raise NotImplementedError("smplx stand-in: SMPL posing (demos.py demo_full) is outside the hot path")
