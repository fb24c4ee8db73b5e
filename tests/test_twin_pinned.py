"""Pins the GRADIENT oracle itself (SURVEY 8c tier 3; VERDICT r03 "missing #4").  Every backward number of the HIP path is
compared with oracle/torch_twin.py (torch-CPU autograd, fp64), because the reference obtains its gradients from
``tf.gradients`` (lib/models.py:447-467), which cannot run here.  This file holds the twin to something tested on its own:

(a) forward: ``TwinCAPE`` reproduces the thirteen golden vectors recorded from the reference's own lib/models.py (the same
    bar tests/test_oracle_golden.py sets for the numpy oracle), so the function being differentiated IS the reference graph;
(b) backward: ``torch.autograd.gradcheck`` (fp64, central differences) of the twin's operators -- chebyshev5, poolwT,
    group_norm, forced_act, edge_loss_calc -- on a 40-vertex graph, and a finite-difference check of d loss_g / d variable for
    variables of every kind (conv weights, dense kernel, bias, group-norm gamma) of small full models, incl. the adversarial term.
"""
import ast
import os

import numpy as np
import pytest
import scipy.sparse as sp
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["affine_nz64", "cmr_nz18", "resblock_udn_tanh", "affine_nz64_b16", "cmr_nz18_b32", "cheb_k6", "switches_relu", "affine_mixed_k",
         "huber_res_affine", "cmr_k3_res", "reduce0", "b2relu_udn", "cond3"]


def _twin(meta_cfg, N, overrides, mesh_ops):
    from oracle.configs import cape_params
    from oracle.torch_twin import TwinCAPE
    P = cape_params(meta_cfg, N)
    P.update(overrides or {})
    m = mesh_ops
    return P, TwinCAPE(m["L"], m["D"], m["U"], m["L_d"], m["D_d"], p=m["p"], dtype=np.float64, tdtype=torch.float64,
                       verts_ref=m["pack"]["template_verts"], vpe=m["pack"]["edges_smpl"], **P)


@pytest.mark.parametrize("tag", CASES)
def test_twin_forward_reproduces_reference_graph(tag, mesh_ops):
    from oracle.golden_inputs import golden_inputs
    g = np.load(os.path.join(GOLD, "ref_%s.npz" % tag))
    meta = ast.literal_eval(str(g["config"]))
    N = meta["N"]
    if N > 4:
        # every op of the path is per sample and every loss a batch mean: the first samples of the static batch are pinned
        # (the whole batch at fp64 on the CPU costs minutes; the numpy oracle covers the full batch)
        N = 2
    P, twin = _twin(meta["cfg"], N, meta["overrides"], mesh_ops)
    full = golden_inputs(meta["N"], P["nz"], meta["seed"], mesh_ops["pack"]["demo_rot"])
    inp = {k: v[:N] for k, v in full.items()}
    with torch.no_grad():
        y, y2 = twin.cond_embeddings(inp["cond"], inp["clo"])
        xh, zm, zl = twin.generator(inp["x"], y, y2, inp["eps"])
        d_fake = twin.discriminator(xh, y, y2)
        yd, y2d = twin.cond_embeddings(inp["cond_d"], inp["clo_d"])
        d_real = twin.discriminator(inp["xd"], yd, y2d)
        ls = twin.losses(xh, inp["gt"], zm, zl, d_real, d_fake)
    rel = lambda a, b: np.abs(a.numpy() - b).max() / max(np.abs(b).max(), 1e-30)
    assert sorted(twin.vs.vars) == [str(n) for n in g["var_names"]]
    assert rel(xh, g["out_op_prediction"][:N].astype(np.float64)) < 5e-7            # golden stored as float32
    assert rel(zm, g["out_z_mean"][:N]) < 1e-10 and rel(zl, g["out_z_logvar"][:N]) < 1e-10
    assert rel(y, g["out_y_latent_g"][:N]) < 1e-12 and rel(y2, g["out_y2_latent_g"][:N]) < 1e-12
    if N == meta["N"]:                                                              # batch means: whole batch only
        for key, val in (("recon_loss", ls["recon"]), ("latent_loss", ls["latent"]), ("edge_loss", ls["edge"]),
                         ("loss_g", ls["gan_g"]), ("loss_d", ls["gan_d"]), ("op_loss_g", ls["loss_g"]),
                         ("op_loss_d", ls["loss_d"]), ("fc_regularization_g", ls["fc_reg_g"])):
            assert abs(float(val) - float(g["out_" + key])) <= 1e-9 * max(1.0, abs(float(g["out_" + key]))), key


# ------------------------------------------------------------------------------------------------------------------
# (b) the twin's backward against finite differences
# ------------------------------------------------------------------------------------------------------------------
def _ring_graph(M=40, seed=0):
    """Symmetric adjacency of a ring with chords (every vertex 4-6 neighbours), like a small closed mesh."""
    rng = np.random.default_rng(seed)
    A = sp.lil_matrix((M, M))
    for i in range(M):
        for d in (1, 2):
            A[i, (i + d) % M] = A[(i + d) % M, i] = 2.0                             # adjacency values are 2.0
    for i in rng.choice(M, M // 4, replace=False):
        j = (i + M // 2 + int(rng.integers(0, 3))) % M
        A[i, j] = A[j, i] = 2.0
    from oracle.cape_oracle import laplacian
    return laplacian(sp.csr_matrix(A, dtype=np.float64))


def _gc(fn, *inputs):
    torch.manual_seed(0)
    assert torch.autograd.gradcheck(fn, inputs, eps=1e-6, atol=1e-7, rtol=1e-5, nondet_tol=0.0)


@pytest.mark.parametrize("K", [1, 2, 3, 6])
def test_gradcheck_chebyshev5(K):
    from oracle import torch_twin as tt
    L = _ring_graph()
    rng = np.random.default_rng(K)
    x = torch.tensor(rng.standard_normal((2, 40, 3)), dtype=torch.float64, requires_grad=True)
    W = torch.tensor(0.3 * rng.standard_normal((3 * K, 4)), dtype=torch.float64, requires_grad=True)
    _gc(lambda x_, W_: tt.chebyshev5(x_, L, W_, K), x, W)


def test_gradcheck_pool_groupnorm_forced_act_edge_loss():
    from oracle import torch_twin as tt
    rng = np.random.default_rng(7)
    M, Mo = 40, 20
    # down-sampling = row selection; up-sampling = three weights per row (Appendix D)
    D = sp.csr_matrix((np.ones(Mo), (np.arange(Mo), np.sort(rng.choice(M, Mo, replace=False)))), shape=(Mo, M))
    U = sp.lil_matrix((M, Mo))
    for r in range(M):
        U[r, rng.choice(Mo, 3, replace=False)] = rng.uniform(-0.6, 1.5, 3)
    x = torch.tensor(rng.standard_normal((2, M, 5)), dtype=torch.float64, requires_grad=True)
    xc = torch.tensor(rng.standard_normal((2, Mo, 5)), dtype=torch.float64, requires_grad=True)
    _gc(lambda t: tt.poolwT(t, D), x)
    _gc(lambda t: tt.poolwT(t, sp.csr_matrix(U)), xc)
    # group norm: G divides C (64 -> 32 groups of 2), and the reference's free-dimension grouping (48 channels -> groups of 1)
    for C in (64, 48, 6):
        xg = torch.tensor(rng.standard_normal((2, 12, C)) * 2 + 0.5, dtype=torch.float64, requires_grad=True)
        ga = torch.tensor(rng.uniform(0.5, 1.5, C), dtype=torch.float64, requires_grad=True)
        be = torch.tensor(rng.standard_normal(C), dtype=torch.float64, requires_grad=True)
        _gc(lambda a, b, c: tt.group_norm(a, b, c), xg, ga, be)
    # forced_act: the branch pattern is DATA, the function is linear per unit
    z = torch.tensor(rng.standard_normal((2, M, 5)), dtype=torch.float64, requires_grad=True)
    sign = rng.random((2, M, 5)) > 0.5
    _gc(lambda t: tt.forced_act(t, 0.2, sign), z)
    rows = np.sort(rng.choice(M, Mo, replace=False))
    _gc(lambda t: tt.forced_act(t, 0.0, sign[:, :Mo], rows=rows), z)
    # edge loss (lib/losses.py:9-25): sqrt of a sum of squares per edge -- keep the edge differences away from 0
    edges = np.stack([np.arange(M), (np.arange(M) + 1) % M], 1)
    pred = torch.tensor(rng.standard_normal((2, M, 3)), dtype=torch.float64, requires_grad=True)
    gt = torch.tensor(rng.standard_normal((2, M, 3)), dtype=torch.float64)
    _gc(lambda t: tt.edge_loss_calc(t, gt, edges), pred)


def _fd_check(twin, inp, names, with_d, h=1e-5, n_dir=2):
    """Directional central differences of loss_g (and loss_d) along random directions of the named variables against the
    twin's autograd gradient.  The (leaky-)ReLU pattern and the L1 sign are FROZEN at the base point (the same mechanism the
    GPU gradient tests use), so the function is smooth along the line and the comparison is at fp64 arithmetic accuracy."""
    import collections

    def evaluate(signs, l1):
        twin.forced_signs = None if signs is None else collections.deque(signs)
        twin.forced_l1_sign = l1
        twin.flip_log = []
        y, y2 = twin.cond_embeddings(inp["cond"], inp["clo"])
        xh, zm, zl = twin.generator(inp["x"], y, y2, inp["eps"])
        d_fake = d_real = None
        if with_d:
            d_fake = twin.discriminator(xh, y, y2)
            yd, y2d = twin.cond_embeddings(inp["cond_d"], inp["clo_d"])
            d_real = twin.discriminator(inp["xd"], yd, y2d)
        assert not twin.forced_signs
        ls = twin.losses(xh, inp["gt"], zm, zl, d_real, d_fake)
        return ls, xh

    twin.sign_log = []
    ls0, xh0 = evaluate(None, None)
    signs = [s for s, _pool in twin.sign_log]
    twin.sign_log = None
    l1 = torch.sign((xh0 - twin._t(inp["gt"])).detach()).numpy() if twin.which_loss == 'l1' else None
    ls, _ = evaluate(signs, l1)
    assert abs(float(ls["loss_g"].detach()) - float(ls0["loss_g"].detach())) < 1e-12 * max(1.0, abs(float(ls0["loss_g"].detach())))
    targets = [("loss_g", ls["loss_g"])] + ([("loss_d", ls["loss_d"])] if with_d else [])
    rng = np.random.default_rng(11)
    worst = 0.0
    # all analytic gradients first: the perturbations below modify the leaves in place
    all_grads = {key: [None if g is None else g.detach().clone() for g in
                       torch.autograd.grad(val, [twin.params[n] for n in names], retain_graph=True, allow_unused=True)]
                 for key, val in targets}
    for key, _val in targets:
        for n, g in zip(names, all_grads[key]):
            if g is None:
                continue
            p = twin.params[n]
            for k in range(n_dir):
                # direction 0: along the gradient itself (the derivative is then |g|: a strong signal); then random ones
                d = g.detach().clone() if k == 0 else torch.tensor(rng.standard_normal(tuple(p.shape)), dtype=torch.float64)
                if float(d.norm()) == 0.0:
                    continue
                d /= d.norm()
                base = p.detach().clone()
                with torch.no_grad():
                    p.copy_(base + h * d)
                    fp = float(evaluate(signs, l1)[0][key])
                    p.copy_(base - h * d)
                    fm = float(evaluate(signs, l1)[0][key])
                    p.copy_(base)
                fd = (fp - fm) / (2 * h)
                an = float((g * d).sum())
                scale = max(abs(an), float(g.norm()) * 1e-3, 1e-12)
                worst = max(worst, abs(fd - an) / scale)
                # noise floor of the difference quotient: rounding of f (|f| * 2^-52 / h) and the h^2 truncation term
                assert abs(fd - an) <= 2e-6 * scale + 2e-9 * max(1.0, abs(fp)), (key, n, fd, an)
    return worst


def test_finite_differences_of_loss_g_affine_model(mesh_ops):
    from oracle.golden_inputs import golden_inputs
    P, twin = _twin("affine_nz18", 1, dict(F=[4, 4, 4, 4, 8, 8, 8, 8], reduce_dim=4), mesh_ops)
    inp = golden_inputs(1, P["nz"], 5, mesh_ops["pack"]["demo_rot"])
    names = ["generator/encoder/encoder_conv2/weights", "generator/encoder/encoder_conv2/bias", "generator/encoder/fc_mean/dense/kernel",
             "generator/decoder/decoder_resblock_affine3/graph_conv/weights", "generator/decoder/decoder_resblock_affine3/affine/weights",
             "generator/decoder/outputs/bias", "condition_pose/fc1/dense/kernel", "discriminator/shared/conv2/weights",
             "discriminator/prediction_map/weights"]
    with torch.no_grad():                       # create the variables
        y, y2 = twin.cond_embeddings(inp["cond"], inp["clo"])
        xh, _, _ = twin.generator(inp["x"], y, y2, inp["eps"])
        twin.discriminator(xh, y, y2)
    missing = [n for n in names if n not in twin.params]
    assert not missing, (missing, sorted(twin.params))
    _fd_check(twin, inp, names, with_d=True)


def test_finite_differences_of_loss_g_groupnorm_model(mesh_ops):
    from oracle.golden_inputs import golden_inputs
    # channel counts the reference's group-norm reshape accepts at batch 1: 32 + 32 condition channels, 16 inside the block
    P, twin = _twin("cmr_nz18", 1, dict(F=[32] * 8, reduce_dim=4), mesh_ops)
    inp = golden_inputs(1, P["nz"], 6, mesh_ops["pack"]["demo_rot"])
    with torch.no_grad():
        y, y2 = twin.cond_embeddings(inp["cond"], inp["clo"])
        twin.generator(inp["x"], y, y2, inp["eps"])
    names = [n for n in sorted(twin.params) if "decoder_resblock_cmr4/" in n] + ["generator/decoder/fc1/dense/kernel"]
    assert any(n.endswith("gamma") for n in names) and any(n.endswith("graph_linear_input/weights") for n in names)
    _fd_check(twin, inp, names, with_d=False)
