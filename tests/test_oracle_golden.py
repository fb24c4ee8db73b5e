"""The CPU oracle (oracle/cape_oracle.py, numpy fp64 restatement) must reproduce the golden vectors
recorded by oracle/make_golden.py, i.e. the outputs of the REFERENCE's own lib/models.py graph-assembly
code executed on the numpy TF1 shim -- same variable names and shapes, same weights (name-keyed
initialisers), same forward values.  This is what pins the oracle; the HIP path is then compared with
the oracle (tests/test_gpu_*.py) and with these vectors directly."""
import ast
import contextlib
import os
import zlib

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["affine_nz64", "cmr_nz18", "resblock_udn_tanh", "affine_nz64_b16", "cmr_nz18_b32", "cheb_k6", "switches_relu", "affine_mixed_k", "huber_res_affine", "cmr_k3_res", "reduce0", "b2relu_udn", "cond3",
         # operand-range cases (oracle/weights.py "range" profile): rows spanning 20+ binades, exactly-zero rows, saturating tanh
         "range_affine", "range_tanh"]


def load_case(tag):
    g = np.load(os.path.join(GOLD, "ref_%s.npz" % tag))
    meta = ast.literal_eval(str(g["config"]))
    return g, meta


@contextlib.contextmanager
def case_profile(meta, mesh_ops):
    """The weight profile a golden case was recorded under (must stay active while the oracle / twin materialises its
    variables, i.e. around the forward pass); yields the per-vertex input field for golden_inputs (None: plain case)."""
    if meta.get("profile") == "range":
        from oracle import weights
        from oracle.golden_inputs import range_fields
        f_in, f_dec = range_fields(mesh_ops["pack"]["template_verts"], mesh_ops["D"])
        with weights.profile("range", f_dec):
            yield f_in
    else:
        assert meta.get("profile") is None, meta
        yield None


def build_oracle(meta, mesh_ops, dtype=np.float64):
    from oracle.cape_oracle import OracleCAPE
    from oracle.configs import cape_params
    P = cape_params(meta["cfg"], meta["N"])
    P.update(meta["overrides"] or {})
    m = mesh_ops
    return P, OracleCAPE(m["L"], m["D"], m["U"], m["L_d"], m["D_d"], p=m["p"], dtype=dtype,
                         verts_ref=m["pack"]["template_verts"], vpe=m["pack"]["edges_smpl"], **P)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_reproduces_reference_graph(tag, mesh_ops):
    from oracle.golden_inputs import golden_inputs
    g, meta = load_case(tag)
    with case_profile(meta, mesh_ops) as in_field:
        P, orc = build_oracle(meta, mesh_ops)
        inp = golden_inputs(meta["N"], P["nz"], meta["seed"], mesh_ops["pack"]["demo_rot"], in_field=in_field)
        y, y2 = orc.cond_embeddings(inp["cond"], inp["clo"])
        xh, zm, zl = orc.generator(inp["x"], y, y2, inp["eps"])
        yd, y2d = orc.cond_embeddings(inp["cond_d"], inp["clo_d"])
        d_fake = orc.discriminator(xh, y, y2)
        d_real = orc.discriminator(inp["xd"], yd, y2d)
    ls = orc.losses(xh, inp["gt"], zm, zl, d_real, d_fake)

    # identical variable inventory (names, shapes, values) to what the reference graph created
    names = [str(n) for n in g["var_names"]]
    assert sorted(orc.vs.vars) == names
    for n, shp, crc in zip(names, g["var_shapes"], g["var_crc"]):
        v = orc.vs.vars[n]
        assert ",".join(str(s) for s in v.shape) == str(shp), n
        assert zlib.crc32(np.ascontiguousarray(v, dtype=np.float32).tobytes()) == int(crc), n

    rel = lambda a, b: np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(xh, g["out_op_prediction"].astype(np.float64)) < 5e-7        # golden stored as float32
    assert rel(zm, g["out_z_mean"]) < 1e-10 and rel(zl, g["out_z_logvar"]) < 1e-10
    assert rel(y, g["out_y_latent_g"]) < 1e-12 and rel(y2, g["out_y2_latent_g"]) < 1e-12
    for key, val in (("recon_loss", ls["recon"]), ("latent_loss", ls["latent"]), ("edge_loss", ls["edge"]),
                     ("loss_g", ls["gan_g"]), ("loss_d", ls["gan_d"]), ("op_loss_g", ls["loss_g"]),
                     ("op_loss_d", ls["loss_d"]), ("fc_regularization_g", ls["fc_reg_g"])):
        assert abs(float(val) - float(g["out_" + key])) <= 1e-9 * max(1.0, abs(float(g["out_" + key]))), key
    # demo-phase ops: encoder outputs and the decoder-only path (model.decode)
    assert rel(zm, g["out_op_vae_mean"]) < 1e-10
    if "out_op_decoder" in g.files:          # (not stored for the batch-16 case: same decoder graph, 1.3 MB)
        zt = np.concatenate([g["out_op_vae_mean"], g["out_op_cond_latent"], g["out_op_cond2_latent"]], 1)
        dec = orc.decoder_cond_vert(zt, g["out_op_cond_latent"], g["out_op_cond2_latent"])
        assert rel(dec, g["out_op_decoder"].astype(np.float64)) < 5e-7


def test_fp32_tier_close_to_fp64(mesh_ops):
    """oracle tier 2 (fp32, reference op order) vs tier 1 (fp64): the noise floor the GPU is judged by."""
    from oracle.golden_inputs import golden_inputs
    g, meta = load_case("affine_nz64")
    P, orc32 = build_oracle(meta, mesh_ops, dtype=np.float32)
    inp = golden_inputs(meta["N"], P["nz"], meta["seed"], mesh_ops["pack"]["demo_rot"])
    y, y2 = orc32.cond_embeddings(inp["cond"], inp["clo"])
    xh, _, _ = orc32.generator(inp["x"], y, y2, inp["eps"])
    ref = g["out_op_prediction"].astype(np.float64)
    err = np.sqrt(((xh.astype(np.float64) - ref) ** 2).sum(-1)).max() / np.sqrt((ref ** 2).sum(-1)).max()
    assert xh.dtype == np.float32 and err < 2e-5
