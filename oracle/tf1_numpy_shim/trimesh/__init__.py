"""Stub of the one trimesh call the reference's model makes (lib/models.py:44:
``trimesh.load(obj, process=False).vertices``).  TEST INFRASTRUCTURE ONLY."""
import numpy as np


class _Mesh(object):
    def __init__(self, vertices):
        self.vertices = vertices


def load(path, process=False):
    verts = []
    with open(path) as fh:
        for line in fh:
            if line.startswith('v '):
                verts.append([float(t) for t in line.split()[1:4]])
    return _Mesh(np.asarray(verts, dtype=np.float64))
