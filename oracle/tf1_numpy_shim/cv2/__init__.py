"""Stub: the reference's lib/utils.py imports cv2 at module top (only cv2.Rodrigues is used, in data
preparation that is off the hot path).  TEST INFRASTRUCTURE ONLY."""


def Rodrigues(*a, **k):
    raise NotImplementedError("cv2 stub")
