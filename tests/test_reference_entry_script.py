"""The reference's own entry scripts, executed UNMODIFIED against cape_amd (SURVEY 7.3 / INTEGRATION.md section 1).

`run_simple_demo.py` (with `demos.py`, `config_parser.py`, `lib/load_data.py`, `lib/utils.py`, `lib/mesh_sampling.py`, the YAML
configs and the shipped data) is copied from /root/reference into a scratch checkout at TEST TIME (never into this repository),
`lib/models.py` there is the one-file shim INTEGRATION.md prescribes, and the script runs through `runpy` as `__main__` with
the shipped affine-nz64 configuration.  Packages the image lacks are the stand-ins of tests/dropin_stubs
(configargparse / trimesh / smplx / psbody) and the numpy TF1 shim (lib/utils.py imports tensorflow and cv2 at module top).

This container has no GPU and the GPU box has no /root/reference, so the two halves meet through a committed fixture:

  * HERE (CPU, this file): the script runs for real up to the three compute entry points, which are replaced by recorders that
    return arrays of the documented shapes; everything else -- argument parsing, the reference's own `load_graph_mtx` on the
    shipped operators, `models.CAPE(**params)` with cape_amd's REAL constructor, `demo_simple`, the .obj export -- is live.
    The recorded call trace (constructor keywords, method calls, every array the script passed in) must equal
    tests/golden/run_simple_demo_trace.npz (`CAPE_WRITE_TRACE=1` regenerates it).
  * ON THE GPU (tests/test_gpu_dropin_api.py::test_reference_demo_trace_on_device): the same trace is replayed against the
    real model: same keywords, same calls, same arrays, the demo's own post-processing, .obj files written and finite.

`main.py --mode train` (the training entry point: BodyData -> generate_transform_matrices -> CAPE(**params) -> build_graph('train')
-> fit -> build_graph('demo') -> demo_full.test_model -> demo_full.run) runs the same way in `test_main_train_unmodified`, with the
reference's OWN lib/mesh_sampling.py (its psbody calls answered by tests/dropin_stubs/psbody), a synthetic dataset in the file
layout BodyData reads (tests/entry_synth.py: the real one is licensed), and an identity stand-in for the SMPL posing step
(tests/dropin_stubs/smplx).  Its trace is tests/golden/main_train_trace.npz; the device half is
tests/test_gpu_dropin_api.py::test_reference_main_train_trace_on_device, which trains for real.
"""
import json
import os
import runpy
import shutil
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("CAPE_REFERENCE", "/root/reference")
TRACE = os.path.join(ROOT, "tests", "golden", "run_simple_demo_trace.npz")
CONFIG = "configs/CAPE-affineconv_nz64_pose32_clotype32_male.yaml"

SHIM = '''# <CAPE checkout>/lib/models.py   (replaces the TF1 file; nothing else in the checkout changes)
from cape_amd.models import CAPE, base_model          # noqa: F401
'''


def _scratch_checkout(dst, script="run_simple_demo.py"):
    for fn in (script, "demos.py", "config_parser.py"):
        shutil.copy(os.path.join(REF, fn), os.path.join(dst, fn))
    os.makedirs(os.path.join(dst, "lib"))
    for fn in ("__init__.py", "load_data.py", "utils.py", "mesh_sampling.py"):
        shutil.copy(os.path.join(REF, "lib", fn), os.path.join(dst, "lib", fn))
    with open(os.path.join(dst, "lib", "models.py"), "w") as f:
        f.write(SHIM)
    shutil.copytree(os.path.join(REF, "configs"), os.path.join(dst, "configs"))
    os.makedirs(os.path.join(dst, "data"))
    for fn in ("template_mesh.obj", "clothing_verts_idx.npy"):
        shutil.copy(os.path.join(REF, "data", fn), os.path.join(dst, "data", fn))
    for d in ("demo_data", "transform_matrices"):
        shutil.copytree(os.path.join(REF, "data", d), os.path.join(dst, "data", d))


def _jsonable(v):
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    if isinstance(v, (np.integer, np.floating, np.bool_)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [_jsonable(x) for x in v]
    raise TypeError(type(v))


_SCRIPT_MODULES = ("demos", "config_parser", "trimesh", "tensorflow", "cv2", "smplx", "psbody")


def _forget_script_modules(drop):
    for mod in [m for m in sys.modules if m.split(".")[0] in ("lib",) + _SCRIPT_MODULES]:
        drop(mod)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on this machine (GPU box): the committed "
                                                   "trace is replayed there instead")
def test_run_simple_demo_unmodified(tmp_path, monkeypatch):
    import cape_amd.models as cm
    dst = str(tmp_path / "checkout")
    os.makedirs(dst)
    _scratch_checkout(dst)

    trace = {"calls": [], "arrays": {}}

    def keep(name, a):
        trace["arrays"][name] = np.array(a)
        return name

    # ---- recorders in place of the three entry points that need the device (no GPU in this container) -----------------
    real_init = cm.CAPE.__init__

    def init(self, *a, **kw):
        assert not a, "the entry scripts pass everything by keyword (run_simple_demo.py:45)"
        ops = {k: kw[k] for k in ("L", "D", "U", "L_d", "D_d")}
        trace["ctor"] = {k: _jsonable(v) for k, v in kw.items() if k not in ops}
        trace["operator_shapes"] = {k: [list(m.shape) for m in v] for k, v in ops.items()}
        trace["operator_nnz"] = {k: [int(m.nnz) for m in v] for k, v in ops.items()}
        real_init(self, **kw)                                      # cape_amd's REAL constructor (host side: no device needed)

    def build_graph(self, input_num_verts, nn_input_channel, phase='train'):
        trace["calls"].append(["build_graph", int(input_num_verts), int(nn_input_channel), phase])

    def encode_only_condition(self, cond, cond2):
        trace["calls"].append(["encode_only_condition", keep("eoc_cond", cond), keep("eoc_cond2", cond2)])
        rng = np.random.default_rng(5)
        return (rng.standard_normal((len(cond), self.nz_cond)).astype(np.float32),
                rng.standard_normal((len(cond2), self.nz_cond2)).astype(np.float32))

    def decode(self, z_total, cond, cond2):
        i = sum(1 for c in trace["calls"] if c[0] == "decode")
        trace["calls"].append(["decode", keep("dec%d_z" % i, z_total), keep("dec%d_cond" % i, cond), keep("dec%d_cond2" % i, cond2)])
        return np.random.default_rng(9 + i).standard_normal((len(z_total), self.input_num_verts, 3)).astype(np.float32) * 1e-2

    monkeypatch.setattr(cm.CAPE, "__init__", init)
    monkeypatch.setattr(cm.CAPE, "build_graph", build_graph)
    monkeypatch.setattr(cm.CAPE, "encode_only_condition", encode_only_condition)
    monkeypatch.setattr(cm.CAPE, "decode", decode)

    # ---- environment of the run: scratch checkout first, stand-ins for absent packages, numpy >= 1.16.3 pickle default -------
    for p in (os.path.join(ROOT, "oracle", "tf1_numpy_shim"), os.path.join(ROOT, "tests", "dropin_stubs"), dst):
        monkeypatch.syspath_prepend(p)
    _forget_script_modules(lambda m: monkeypatch.delitem(sys.modules, m))
    monkeypatch.chdir(dst)
    monkeypatch.setattr(sys, "argv", ["run_simple_demo.py", "--config", CONFIG, "--name", "entry_script_test"])
    np_load = np.load
    monkeypatch.setattr(np, "load", lambda *a, **k: np_load(*a, **dict(k, allow_pickle=True)))   # the shipped A/D/U are pickled lists
    try:
        runpy.run_path(os.path.join(dst, "run_simple_demo.py"), run_name="__main__")
    finally:
        _forget_script_modules(lambda m: sys.modules.pop(m, None))

    # ---- what the script did ------------------------------------------------------------------------------------------------
    assert trace["calls"][0] == ["build_graph", 6890, 3, "demo"]
    assert [c[0] for c in trace["calls"]] == ["build_graph", "encode_only_condition"] + ["decode"] * 4
    assert trace["arrays"]["eoc_cond"].shape == (4, 126) and trace["arrays"]["eoc_cond2"].shape == (4, 4)
    assert all(trace["arrays"]["dec%d_z" % i].shape == (3, 64 + 32 + 32) and trace["arrays"]["dec%d_cond" % i].shape == (1, 32) for i in range(4))
    assert trace["ctor"]["F"] == [64, 64, 128, 128, 256, 256, 512, 512] and trace["ctor"]["K"] == [2] * 8 and trace["ctor"]["affine"] is True
    assert trace["operator_shapes"]["L"][0] == [6890, 6890] and trace["operator_shapes"]["L_d"][-1] == [431, 431]
    objs = sorted(os.listdir(os.path.join(dst, "results", "demo_results")))
    assert len(objs) == 12 and objs[0].endswith(".obj")           # 4 clothing types x 3 samples (demos.py:372-407)
    first = [l for l in open(os.path.join(dst, "results", "demo_results", objs[0])) if l.startswith("v ")]
    assert len(first) == 6890

    # ---- pinned against the committed fixture (what the GPU side replays) ---------------------------------------------------------
    stats = np_load(os.path.join(dst, "data", "demo_data", "trainset_stats.npz"))
    payload = dict(meta=np.array(json.dumps(dict(ctor=trace["ctor"], calls=trace["calls"], operator_shapes=trace["operator_shapes"],
                                                 operator_nnz=trace["operator_nnz"], config=CONFIG), sort_keys=True)),
                   train_mean=stats["mean"].astype(np.float32), train_std=stats["std"].astype(np.float32),
                   **{k: v.astype(np.float32) for k, v in trace["arrays"].items()})
    if os.environ.get("CAPE_WRITE_TRACE") == "1":
        np.savez_compressed(TRACE, **payload)
    g = np.load(TRACE)
    assert json.loads(str(g["meta"])) == json.loads(str(payload["meta"]))
    for k, v in payload.items():
        if k != "meta":
            assert np.array_equal(g[k], v), k


MAIN_TRACE = os.path.join(ROOT, "tests", "golden", "main_train_trace.npz")
MAIN_ARGV = ["main.py", "--config", CONFIG, "--name", "entry_main_test", "--mode", "train", "--dataset", "synthetic_entry",
             "--num_epochs", "1", "--demo_n_sample", "2", "--vis_demo", "0"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout is not on this machine (GPU box): the committed "
                                                   "trace is replayed there instead")
def test_main_train_unmodified(tmp_path, monkeypatch):
    """main.py --mode train, unmodified, from BodyData to the last .obj of demo_full.run (module docstring)."""
    import cape_amd.models as cm
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import entry_synth
    finally:
        sys.path.pop(0)
    dst = str(tmp_path / "checkout")
    os.makedirs(dst)
    _scratch_checkout(dst, script="main.py")
    shutil.copy(os.path.join(REF, "data", "edges_smpl.npy"), os.path.join(dst, "data", "edges_smpl.npy"))      # demos.py:31
    # the dataset files main.py:22-29 names, in BodyData's layout (lib/load_data.py:58-84)
    for split, names in (("train", ("train_disp", "train_rot", "train_clo_label")), ("test", ("test_disp", "test_rot", "test_clo_label"))):
        os.makedirs(os.path.join(dst, "data", "datasets", "synthetic_entry", split))
    for k, v in entry_synth.arrays().items():
        np.save(os.path.join(dst, "data", "datasets", "synthetic_entry", k.split("_")[0], k + ".npy"), v)

    trace = {"calls": [], "arrays": {}, "summaries": {}}

    def keep(name, a):
        trace["arrays"][name] = np.array(a)
        return name

    def digest(name, a):
        trace["summaries"][name] = entry_synth.summary(a)
        return name

    real_init = cm.CAPE.__init__

    def init(self, *a, **kw):
        assert not a, "main.py:87 passes everything by keyword"
        ops = {k: kw[k] for k in ("L", "D", "U", "L_d", "D_d")}
        trace["ctor"] = {k: _jsonable(v) for k, v in kw.items() if k not in ops}
        trace["operator_shapes"] = {k: [list(m.shape) for m in v] for k, v in ops.items()}
        trace["operator_nnz"] = {k: [int(m.nnz) for m in v] for k, v in ops.items()}
        trace["operators"] = ops
        real_init(self, **kw)

    def build_graph(self, input_num_verts, nn_input_channel, phase='train'):
        trace["calls"].append(["build_graph", int(input_num_verts), int(nn_input_channel), phase])

    def fit(self, data_wrapper):
        want = entry_synth.Wrapper()
        for k in entry_synth.Wrapper.FIELDS:                      # the reference's BodyData, field by field, bit for bit
            assert np.array_equal(getattr(data_wrapper, k), getattr(want, k)), k
            assert getattr(data_wrapper, k).dtype == getattr(want, k).dtype, k
            digest("bodydata." + k, getattr(data_wrapper, k))
        trace["calls"].append(["fit", "bodydata"])
        return [0.25], 0.5                                         # (validation losses per epoch, seconds per step)  main.py:92

    def predict(self, data, cond, cond2, labels=None, sess=None, phase='train'):
        assert sess is None and labels is data                     # demos.py:63-67
        trace["calls"].append(["predict", digest("predict.data", data), digest("predict.cond", cond), digest("predict.cond2", cond2), phase])
        return np.random.default_rng(3).standard_normal(data.shape).astype(np.float32) * 0.1, 0.1, 0.2, 0.3

    def encode_only_condition(self, cond, cond2):
        i = sum(1 for c in trace["calls"] if c[0] == "encode_only_condition")
        trace["calls"].append(["encode_only_condition", keep("eoc%d_cond" % i, cond), keep("eoc%d_cond2" % i, cond2)])
        rng = np.random.default_rng(5 + i)
        return (rng.standard_normal((len(cond), self.nz_cond)).astype(np.float32),
                rng.standard_normal((len(cond2), self.nz_cond2)).astype(np.float32))

    def decode(self, z_total, cond, cond2):
        i = sum(1 for c in trace["calls"] if c[0] == "decode")
        trace["calls"].append(["decode", keep("dec%d_z" % i, z_total), keep("dec%d_cond" % i, cond), keep("dec%d_cond2" % i, cond2)])
        return np.random.default_rng(9 + i).standard_normal((len(z_total), self.input_num_verts, 3)).astype(np.float32) * 1e-2

    for name, fn in (("__init__", init), ("build_graph", build_graph), ("fit", fit), ("predict", predict),
                     ("encode_only_condition", encode_only_condition), ("decode", decode)):
        monkeypatch.setattr(cm.CAPE, name, fn)

    for p in (os.path.join(ROOT, "oracle", "tf1_numpy_shim"), os.path.join(ROOT, "tests", "dropin_stubs"), dst):
        monkeypatch.syspath_prepend(p)
    _forget_script_modules(lambda m: monkeypatch.delitem(sys.modules, m))
    monkeypatch.chdir(dst)
    monkeypatch.setenv("CAPE_STUB_SMPL", "identity")
    monkeypatch.setattr(sys, "argv", list(MAIN_ARGV))
    np_load = np.load
    monkeypatch.setattr(np, "load", lambda *a, **k: np_load(*a, **dict(k, allow_pickle=True)))
    try:
        runpy.run_path(os.path.join(dst, "main.py"), run_name="__main__")
    finally:
        _forget_script_modules(lambda m: sys.modules.pop(m, None))

    # ---- what the script did ------------------------------------------------------------------------------------------------
    names = [c[0] for c in trace["calls"]]
    n_pose = len(np_load(os.path.join(dst, "data", "demo_data", "demo_pose_params.npz"))["rot"])
    assert names == (["build_graph", "fit", "build_graph", "predict", "encode_only_condition"] + ["decode"] * n_pose
                     + ["encode_only_condition"] + ["decode"] * 4)                       # main.py:91-101, demos.py:47-219,329-331
    assert trace["calls"][0] == ["build_graph", 6890, 3, "train"] and trace["calls"][2] == ["build_graph", 6890, 3, "demo"]
    assert trace["calls"][3][-1] == "test"
    c = trace["ctor"]
    assert c["F"] == [64, 64, 128, 128, 256, 256, 512, 512] and c["K"] == [2] * 8 and c["affine"] is True and c["num_epochs"] == 1
    assert c["decay_steps"] == 2 * entry_synth.N_TRAIN / 16 and c["cond_dim"] == 126 and c["batch_size"] == 16   # main.py:69-70
    assert c["p"] == [6890, 6890, 3445, 3445, 1723, 1723, 862, 862, 862]
    # the operators the reference's OWN generate_transform_matrices (lib/mesh_sampling.py:243-263, psbody stand-ins) produced
    # are the ones cape_amd ships in its pack (and so the ones the device half builds the model on)
    from cape_amd.load_data import load_graph_mtx
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)
    for k, mats in dict(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2).items():
        for a, b in zip(trace["operators"][k], mats):
            assert a.shape == b.shape and abs(a - b).max() <= 1e-6, k
    res = os.path.join(dst, "results", "entry_main_test")
    assert len(os.listdir(os.path.join(res, "test_reconstruction_objs_synthetic_entry"))) == 2      # every 2nd of 4 test examples
    assert len(os.listdir(os.path.join(res, "sample_vary_pose"))) == 2 * n_pose
    assert len(os.listdir(os.path.join(res, "sample_vary_clotype"))) == 2 * 4
    assert "Eucledian err mean" in open(os.path.join(res, "test_results_synthetic_entry.txt")).read()

    payload = dict(meta=np.array(json.dumps(dict(ctor=trace["ctor"], calls=trace["calls"], operator_shapes=trace["operator_shapes"],
                                                 operator_nnz=trace["operator_nnz"], summaries=trace["summaries"], argv=MAIN_ARGV,
                                                 n_pose=int(n_pose)), sort_keys=True)),
                   **{k: v.astype(np.float32) for k, v in trace["arrays"].items()})
    if os.environ.get("CAPE_WRITE_TRACE") == "1":
        np.savez_compressed(MAIN_TRACE, **payload)
    g = np.load(MAIN_TRACE)
    assert json.loads(str(g["meta"])) == json.loads(str(payload["meta"]))
    for k, v in payload.items():
        if k != "meta":
            assert np.array_equal(g[k], v), k
