#!/usr/bin/env python3
"""Generate tests/golden/ref_*.npz by executing the REFERENCE's own lib/models.py (graph assembly,
layer order, indices, weight layouts, variable names) on the numpy TF1 shim in
oracle/tf1_numpy_shim.  Runs only where /root/reference exists (the build container); the committed
.npz files are what the tests and the GPU box use.  TEST INFRASTRUCTURE ONLY.

What this pins: everything the reference does in Python.  What it cannot pin: TensorFlow 1.13's C++
kernels, which the shim replaces by numpy ops with the documented semantics (float64 compute).
"""
import copy
import os
import sys
import zlib

sys.dont_write_bytecode = True            # never write __pycache__ into the read-only reference tree
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CAPE_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "tf1_numpy_shim"))
sys.path.insert(1, REF)
sys.path.insert(2, os.path.dirname(HERE))

import numpy as np                         # noqa: E402
import tensorflow as tf                    # noqa: E402  (the shim)
from lib import models                     # noqa: E402  (the reference's model code)
from lib.load_data import load_graph_mtx   # noqa: E402  (the reference's loader)
from lib.utils import filter_cloth_pose    # noqa: E402

from oracle.configs import cape_params     # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def range_setup():
    """(input field, decoder field) of the "range" profile from the reference's own template and row selections."""
    from oracle.golden_inputs import range_fields
    verts = np.array([[float(t) for t in l.split()[1:4]] for l in open(os.path.join(REF, "data", "template_mesh.obj")) if l.startswith("v ")])
    _np_load = np.load
    np.load = lambda *a, **k: _np_load(*a, **dict(k, allow_pickle=True))
    try:
        D = load_graph_mtx(REF, load_for_demo=True)[1]
    finally:
        np.load = _np_load
    return range_fields(verts, D)


def inputs(N, nz, seed, in_field=None):
    from oracle.golden_inputs import golden_inputs
    rot = np.load(os.path.join(REF, "data", "demo_data", "demo_pose_params.npz"))["rot"]
    d = golden_inputs(N, nz, seed, rot, in_field=in_field)
    # cross-check our filter_cloth_pose restatement against the reference's (lib/utils.py:38-62)
    assert np.array_equal(d["cond"], np.tile(filter_cloth_pose(rot), (N // 6 + 1, 1))[:N].astype(np.float32))
    return d


def run_chunked(tag, cfg, overrides, N, seed, chunk):
    """A batch the lazy numpy graph cannot hold in this container's memory (the GraphCMR generator at batch 32 needs > 54 GB
    on the float64 shim): the reference's graph code is run on ``N / chunk`` consecutive slices of the batch-N inputs at
    static batch ``chunk`` and the results are assembled.  Every op of the path acts per sample (lib/models.py:81-83,
    147-151; group-norm statistics per sample, :698-699, for channel counts divisible by the group count) and every loss
    is a batch mean, so per-sample outputs concatenate and scalar losses average (equal slice sizes) to exactly the values
    of the batch-N graph; the variable inventory does not depend on the batch.  The stored config says so."""
    assert N % chunk == 0
    full = inputs(N, cape_params(cfg, N)["nz"] if not (overrides or {}).get("nz") else overrides["nz"], seed)
    parts = []
    for k in range(N // chunk):
        sl = slice(k * chunk, (k + 1) * chunk)
        parts.append(run(None, cfg, overrides, chunk, seed, given={key: v[sl] for key, v in full.items()}))
        print("  slice", k, "done")
    out = {}
    for key in parts[0]:
        vals = [p_[key] for p_ in parts]
        if key.startswith("out_"):
            v0 = np.asarray(vals[0])
            out[key] = np.concatenate(vals, 0) if v0.ndim >= 1 and v0.shape[0] == chunk else np.mean(vals, 0)
        elif key.startswith("var_"):
            assert all(np.array_equal(v, vals[0]) for v in vals), key
            out[key] = vals[0]
    out["out_op_prediction"] = out["out_op_prediction"].astype(np.float32)
    out.pop("out_op_decoder", None)
    out["meta_N"], out["meta_nz"], out["meta_seed"] = np.int64(N), parts[0]["meta_nz"], np.int64(seed)
    out["config"] = np.array(repr(dict(cfg=cfg, overrides=overrides, N=N, seed=seed, assembled_from_static_batch=chunk)))
    fn = os.path.join(OUT, "ref_%s.npz" % tag)
    np.savez_compressed(fn, **out)
    print(tag, "->", fn, os.path.getsize(fn), "bytes;", len(out["var_names"]), "variables; prediction mean|.|",
          float(np.abs(out["out_op_prediction"]).mean()))


def run(tag, cfg, overrides, N, seed, given=None, profile=None):
    if profile is not None:
        # operand-range cases (oracle/weights.py "range" profile): zero conv biases, decoder dense kernel scaled per vertex,
        # displacements scaled per vertex over 22 binades with an exactly-zero region
        from oracle import weights
        assert profile == "range" and given is None
        f_in, f_dec = range_setup()
        nz = dict(cape_params(cfg, N), **(overrides or {}))["nz"]
        with weights.profile("range", f_dec):
            return run(tag, cfg, overrides, N, seed, given=inputs(N, nz, seed, in_field=f_in), profile=None)
    tf.shim_reset()
    # the reference targets numpy < 1.16.3 where np.load unpickled object arrays by default
    _np_load = np.load
    np.load = lambda *a, **k: _np_load(*a, **dict(k, allow_pickle=True))
    try:
        L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(REF, load_for_demo=True)  # run_simple_demo.py:14
    finally:
        np.load = _np_load
    params = cape_params(cfg, N)
    params.update(overrides or {})
    params["p"] = p
    nz = params["nz"]
    inp = given if given is not None else inputs(N, nz, seed)
    tf.shim_configure(seed=params["seed"], compute_dtype=np.float64, eps=inp["eps"])
    ref_params = copy.deepcopy(params)
    for k in ("lr", "num_epochs", "decay_rate", "decay_steps", "momentum", "optimizer"):
        ref_params.setdefault(k, None)
    model = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, **ref_params)          # run_simple_demo.py:45
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='demo')  # run_simple_demo.py:47
    sess = tf.Session(graph=model.graph)
    feed = {model.ph_data_g: inp["x"], model.ph_cond_g: inp["cond"], model.ph_cond2_g: inp["clo"],
            model.ph_gt: inp["gt"], model.ph_data_d: inp["xd"], model.ph_cond_d: inp["cond_d"],
            model.ph_cond2_d: inp["clo_d"], model.is_train: False}
    names = ["op_prediction", "z_mean", "z_logvar", "recon_loss", "latent_loss", "edge_loss", "loss_g", "loss_d",
             "op_loss_g", "op_loss_d", "fc_regularization_g", "op_vae_mean", "op_vae_var", "op_cond_latent",
             "op_cond2_latent", "y_latent_g", "y2_latent_g"]
    vals = sess.run([getattr(model, n) for n in names], feed)
    out = {"out_" + n: np.asarray(v, dtype=np.float64) for n, v in zip(names, vals)}
    out64 = dict(out)
    # decoder-only path used by demos.py:392-395 (model.decode)
    z_total = np.concatenate([out["out_op_vae_mean"], out["out_op_cond_latent"], out["out_op_cond2_latent"]], 1)
    out["out_op_decoder"] = np.asarray(sess.run(model.op_decoder, {
        model.ph_z_total: z_total, model.ph_y_latent: out["out_op_cond_latent"],
        model.ph_y2_latent: out["out_op_cond2_latent"], model.is_train: False}), dtype=np.float64)
    # big tensors are stored as float32 (6e-8 resolution); inputs are regenerated from the seed
    for k in ("out_op_prediction", "out_op_decoder"):
        out[k] = out[k].astype(np.float32)
    if N > 4:
        del out["out_op_decoder"]          # same decoder graph on the same z: not worth another 1.3 MB at batch 16
    out["meta_N"], out["meta_nz"], out["meta_seed"] = np.int64(N), np.int64(nz), np.int64(seed)
    # variable inventory: names, shapes and checksums of the weights the reference graph created
    var = tf.shim_variables()
    vn = sorted(var)
    out["var_names"] = np.array(vn)
    out["var_shapes"] = np.array([",".join(str(s) for s in var[n].shape) for n in vn])
    out["var_sums"] = np.array([float(np.asarray(var[n], np.float64).sum()) for n in vn])
    out["var_crc"] = np.array([zlib.crc32(np.ascontiguousarray(var[n], dtype=np.float32).tobytes()) for n in vn], dtype=np.int64)
    from oracle import weights as _w
    meta = dict(cfg=cfg, overrides=overrides, N=N, seed=seed)
    if _w.PROFILE is not None:
        meta["profile"] = _w.PROFILE["kind"]
    out["config"] = np.array(repr(meta))
    if tag is None:
        return out
    fn = os.path.join(OUT, "ref_%s.npz" % tag)
    np.savez_compressed(fn, **out)
    print(tag, "->", fn, os.path.getsize(fn), "bytes;", len(vn), "variables; prediction mean|.|",
          float(np.abs(out["out_op_prediction"]).mean()))


CASES = [
    ("affine_nz64", "affine_nz64", None, 2, 11),
    ("cmr_nz18", "cmr_nz18", None, 2, 12),
    ("resblock_udn_tanh", "affine_nz18", dict(use_res_block=True, use_res_block_dec=False, cond_encoder=True,
                                             activation='b1tanh', F=[16, 16, 32, 32, 64, 64, 128, 128],
                                             reduce_dim=32, loss='l2'), 2, 13),
    # BASELINE configs[2] at its stated batch (static batch 16, reference lib/models.py:272-282, config_parser.py:33):
    # the shapes for which the HIP library selects its large-tile kernels
    ("affine_nz64_b16", "affine_nz64", None, 16, 21),
    # polynomial order 6 in every generator layer: the explicit Chebyshev recurrence of lib/models.py:88-96 (K > 2) and
    # the [M, Fin, K] -> [M*N.., Fin*K] reshuffle of :97-102 with K = 6 (BASELINE configs[1] is one such layer)
    # option switches no shipped YAML uses: two-layer clothing-type network, ReLU, discriminator order 2, condition
    # networks excluded from the optimiser, conditioned plain encoder with the udn decoder
    ("switches_relu", "affine_nz18", dict(n_layer_cond=2, activation='b1relu', Kd=2, optim_condnet=False, use_res_block=False,
                                          use_res_block_dec=False, cond_encoder=True, F=[16, 16, 32, 32, 64, 64, 128, 128],
                                          reduce_dim=16), 2, 15),
    # mixed polynomial orders with the affine res-blocks: the precomposed K = 3 operators and the recurrence (K = 4, 5, 6)
    # next to each other, the affine block in both its fused and its composed form
    ("affine_mixed_k", "affine_nz18", dict(K=[3, 4, 3, 4, 2, 5, 3, 6], F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=16), 2, 16),
    # Huber reconstruction loss (lib/models.py:360-363) with b1relu and the res-block encoder on the affine decoder
    ("huber_res_affine", "affine_nz18", dict(loss='huber', activation='b1relu', use_res_block=True, cond_encoder=True,
                                             F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=16), 2, 17),
    # GraphCMR / group-norm decoder with polynomial order 3, res-block encoder and conditioned encoder
    ("cmr_k3_res", "cmr_nz18", dict(K=[3] * 8, use_res_block=True, cond_encoder=True,
                                    F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=16), 2, 18),
    # no channel reduction before / after the dense layers (reduce_dim = 0 skips the 1x1 convolutions, :175-180, :590-597)
    ("reduce0", "affine_nz18", dict(reduce_dim=0, F=[16, 16, 32, 32, 32, 32, 32, 32]), 2, 31),
    # b2relu: one bias per vertex and channel (:148-152) in every conv layer, udn decoder
    ("b2relu_udn", "affine_nz18", dict(activation='b2relu', use_res_block_dec=False, F=[16, 16, 32, 32, 64, 64, 128, 128],
                                       reduce_dim=16), 2, 32),
    # three-layer clothing-type network, other embedding sizes (:479-511)
    ("cond3", "affine_nz18", dict(n_layer_cond=3, nz_cond=16, nz_cond2=4, F=[16, 16, 32, 32, 64, 64, 128, 128],
                                  reduce_dim=16), 2, 33),
    # BASELINE configs[3] at its stated batch: the GraphCMR / group-norm generator + discriminator at static batch 32
    ("cmr_nz18_b32", "cmr_nz18", None, 32, 41, 4),          # assembled from eight static-batch-4 runs, see run_chunked
    ("cheb_k6", "affine_nz18", dict(use_res_block=True, use_res_block_dec=False, cond_encoder=True, K=[6] * 8,
                                    F=[16, 16, 32, 32, 32, 32, 64, 64], reduce_dim=16), 2, 14),
]

# operand-range cases of the fp16 two-piece contractions at model level (see run(profile=...)): rows spanning 20+ binades, zero
# rows; the second one with tanh activations, which saturate on the large rows and are linear on the small ones
RANGE_CASES = [
    ("range_affine", "affine_nz64", None, 2, 51),
    ("range_tanh", "affine_nz18", dict(use_res_block=True, use_res_block_dec=False, cond_encoder=True, activation='b1tanh',
                                       F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=32, loss='l2'), 2, 52),
]


def main():
    only = set(sys.argv[1:])               # optional: tags to (re)generate
    for case in CASES:
        if not only or case[0] in only:
            (run_chunked if len(case) == 6 else run)(*case)
    for case in RANGE_CASES:
        if not only or case[0] in only:
            run(*case, profile="range")


if __name__ == "__main__":
    import threading
    sys.setrecursionlimit(100000)          # the lazy graph is evaluated recursively (GN decoder is deep)
    threading.stack_size(512 * 1024 * 1024)
    t = threading.Thread(target=main)
    t.start()
    t.join()
