"""Stand-in for psbody.mesh.Mesh as the reference's entry scripts construct it (``Mesh(filename=obj)`` -> ``.v`` / ``.f``;
``Mesh(v=, f=)``).  TEST INFRASTRUCTURE ONLY."""
import numpy as np


class Mesh(object):
    def __init__(self, v=None, f=None, filename=None):
        if filename is not None:
            import trimesh
            m = trimesh.load(filename, process=False)
            v, f = m.vertices, m.faces
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f, dtype=np.int64)

    def write_obj(self, path):
        import trimesh
        trimesh.Trimesh(vertices=self.v, faces=self.f).export(path)
