"""Stand-in for the two trimesh calls of the reference's demos.py (:352 ``trimesh.load(obj, process=False)`` -> ``.vertices`` /
``.faces``; :406-407 ``trimesh.Trimesh(vertices=, faces=).export(path)``) and the import in lib/models.py:8.  Wavefront OBJ
in / out.  TEST INFRASTRUCTURE ONLY (tests/test_reference_entry_script.py)."""
import numpy as np


class Trimesh(object):
    def __init__(self, vertices=None, faces=None, process=False):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)

    def export(self, path):
        with open(path, "w") as f:
            for v in self.vertices:
                f.write("v %.8f %.8f %.8f\n" % tuple(v))
            for t in self.faces:
                f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))
        return path


def load(path, process=False):
    v, t = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                v.append([float(x) for x in p[1:4]])
            elif p[0] == "f":
                t.append([int(x.split("/")[0]) - 1 for x in p[1:4]])
    return Trimesh(vertices=np.array(v), faces=np.array(t))
