"""Stand-in for psbody.mesh.Mesh as the reference's entry scripts and lib/mesh_sampling.py use it: ``Mesh(filename=obj)`` ->
``.v`` / ``.f``; ``Mesh(v=, f=)``; ``write_obj``; ``compute_aabb_tree().nearest(points, True)`` (lib/mesh_sampling.py:73, the
closest-point query behind the up-sampling matrices -- answered by cape_amd.mesh_operators.closest_points_on_mesh).
TEST INFRASTRUCTURE ONLY."""
import numpy as np


class _AabbTree(object):
    def __init__(self, mesh):
        self.mesh = mesh

    def nearest(self, points, nearest_part=False):
        from cape_amd.mesh_operators import closest_points_on_mesh
        f, part, pt = closest_points_on_mesh(self.mesh.v, self.mesh.f, points)
        if nearest_part:
            return f[None].astype(np.uint32), part[None].astype(np.uint32), pt
        return f[None].astype(np.uint32), pt


class Mesh(object):
    def __init__(self, v=None, f=None, filename=None):
        if filename is not None:
            import trimesh
            m = trimesh.load(filename, process=False)
            v, f = m.vertices, m.faces
        self.v = np.asarray(v, dtype=np.float64)
        self.f = np.asarray(f, dtype=np.int64)

    def compute_aabb_tree(self):
        return _AabbTree(self)

    def write_obj(self, path):
        import trimesh
        trimesh.Trimesh(vertices=self.v, faces=self.f).export(path)


class MeshViewers(object):
    """demos.py imports the name next to Mesh (:254, :296); the interactive viewer itself is never opened here (--vis_demo 0)."""

    def __init__(self, *a, **kw):
        raise NotImplementedError("psbody stand-in: no interactive viewer")
