#!/usr/bin/env python3
"""Build cape_amd/data/smpl_mesh_pack.npz (and tests/golden/template_faces.npy) from the reference's shipped data assets.

Runs ONLY in the build container (needs /root/reference, which does not exist on
the GPU box).  The pack is a plain-array re-encoding (no pickles) of

  * data/transform_matrices/{for_demo,ds2}/{A,D,U}.npy   (scipy csc, loaded the way
    lib/load_data.py:7-32 loads them -- np.load(..., encoding='latin1'))
  * data/edges_smpl.npy                                   (lib/models.py:45)
  * the 'v' lines of data/template_mesh.obj               (lib/models.py:44); its 'f' lines (triangles, 0-based) go to
    the separate tests/golden/template_faces.npy -- the input of the operator generation (main.py:31-39)
  * data/demo_data/demo_pose_params.npz 'rot'             (demos.py:367-376)

Every matrix is stored in the exact CSC form the reference ships (indptr, indices,
data with the shipped dtype), so `cape_amd.load_data.load_graph_mtx` can rebuild
bit-identical scipy matrices without /root/reference.  No reference *source* is
copied; these are data assets used as test/bench fixtures.
"""
import os
import sys
import numpy as np

REF = os.environ.get("CAPE_REFERENCE", "/root/reference")
_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(_ROOT, "cape_amd", "data", "smpl_mesh_pack.npz")          # product data (operators the package loads)
FACES = os.path.join(_ROOT, "tests", "golden", "template_faces.npy")           # test fixture


def main():
    pack = {}
    for hier in ("for_demo", "ds2"):
        for name in ("A", "D", "U"):
            mats = list(np.load(os.path.join(REF, "data", "transform_matrices", hier, name + ".npy"),
                                encoding="latin1", allow_pickle=True))
            pack["%s_%s_count" % (hier, name)] = np.int64(len(mats))
            for i, m in enumerate(mats):
                m = m.tocsc()
                key = "%s_%s_%d" % (hier, name, i)
                pack[key + "_shape"] = np.asarray(m.shape, dtype=np.int64)
                pack[key + "_indptr"] = m.indptr.astype(np.int32)
                pack[key + "_indices"] = m.indices.astype(np.int32)
                pack[key + "_data"] = m.data  # dtype preserved (f32 for_demo, f64 ds2)
    pack["edges_smpl"] = np.load(os.path.join(REF, "data", "edges_smpl.npy")).astype(np.int32)
    verts, faces = [], []
    with open(os.path.join(REF, "data", "template_mesh.obj")) as fh:
        for line in fh:
            if line.startswith("v "):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith("f "):
                faces.append([int(t.split("/")[0]) - 1 for t in line.split()[1:4]])
    faces = np.asarray(faces, dtype=np.int32)
    assert faces.shape == (13776, 3) and faces.min() == 0 and faces.max() == 6889
    if "--faces-only" in sys.argv:
        np.save(FACES, faces)
        print("wrote template_faces.npy", faces.shape)
        return
    pack["template_verts"] = np.asarray(verts, dtype=np.float64)
    assert pack["template_verts"].shape == (6890, 3)
    pose = np.load(os.path.join(REF, "data", "demo_data", "demo_pose_params.npz"))
    pack["demo_rot"] = pose["rot"].astype(np.float64)
    np.savez_compressed(OUT, **pack)
    np.save(FACES, faces)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(pack), "arrays")


if __name__ == "__main__":
    sys.exit(main())
