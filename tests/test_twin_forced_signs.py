"""The gradient-parity tests replay the DEVICE's (leaky-)ReLU branch pattern in the fp64 twin (oracle.torch_twin:
``forced_signs`` / ``forced_act``).  This CPU test pins that mechanism on the twin alone: replaying the twin's own pattern
changes nothing; a pattern given only on the kept rows of a row-selection pool is expanded correctly; flipping the branch
of one unit changes exactly the gradients behind it."""
import collections

import numpy as np
import scipy.sparse as sp
import torch


def _small_twin(mesh_ops):
    from oracle.configs import cape_params
    from oracle.torch_twin import TwinCAPE
    P = cape_params("affine_nz18", 1)
    P.update(F=[8, 8, 8, 8, 16, 16, 16, 16], reduce_dim=8)
    m = mesh_ops
    return P, TwinCAPE(m["L"], m["D"], m["U"], m["L_d"], m["D_d"], p=m["p"], dtype=np.float64, tdtype=torch.float64,
                       verts_ref=m["pack"]["template_verts"], vpe=m["pack"]["edges_smpl"], **P)


def _run(twin, inp, signs=None):
    twin.forced_signs = None if signs is None else collections.deque(signs)
    twin.flip_log = []
    y, y2 = twin.cond_embeddings(inp["cond"], inp["clo"])
    xh, zm, zl = twin.generator(inp["x"], y, y2, inp["eps"])
    d_fake = twin.discriminator(xh, y, y2)
    assert not twin.forced_signs
    twin.forced_signs = None
    ls = twin.losses(xh, inp["gt"], zm, zl, None, d_fake)
    names = sorted(twin.params)
    grads = torch.autograd.grad(ls["loss_g"], [twin.params[n] for n in names], allow_unused=True)
    return float(ls["loss_g"]), {n: (None if g is None else g.numpy().copy()) for n, g in zip(names, grads)}


def test_replaying_a_branch_pattern(mesh_ops):
    from oracle.golden_inputs import golden_inputs
    P, twin = _small_twin(mesh_ops)
    inp = golden_inputs(1, P["nz"], 5, mesh_ops["pack"]["demo_rot"])
    twin.sign_log = []
    loss0, g0 = _run(twin, inp)
    log, twin.sign_log = twin.sign_log, None
    assert len(log) == 1 + 8 + 1 + 8 + 4          # pose MLP, encoder convs, decoder fc1, affine blocks, discriminator convs

    # (1) own pattern, given -- like the device does for layers evaluated on the kept vertices only -- on the pooled rows
    signs, pooled = [], 0
    for s, pool in log:
        if pool is not None and sp.csr_matrix(pool).shape[0] != s.shape[1]:
            s = s[:, torch.as_tensor(sp.csr_matrix(pool).indices, dtype=torch.long)]
            pooled += 1
        signs.append(s)
    assert pooled == 3 + 4                        # three down-sampling encoder layers, four discriminator layers
    loss1, g1 = _run(twin, inp, signs)
    assert sum(twin.flip_log) == 0 and loss1 == loss0
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None)
        if g0[n] is not None:
            assert np.array_equal(g0[n], g1[n]), n

    # (2) one unit of the last encoder layer takes the other branch: gradients behind it move, the decoder's do not
    site = 1 + 7
    flipped = [s.clone() for s in signs]
    flipped[site][0, 5, 3] = ~flipped[site][0, 5, 3]
    loss2, g2 = _run(twin, inp, flipped)
    assert sum(twin.flip_log) == 1
    moved = {n for n in g0 if g0[n] is not None and not np.array_equal(g0[n], g2[n])}
    assert "generator/encoder/encoder_conv8/weights" in moved and "generator/encoder/encoder_conv1/weights" in moved
    # the decoder sees a (slightly) different latent code, so its gradients move as well -- but only by the size of the
    # perturbation, while the flipped unit's own bias gradient changes by O(1) of that channel's entry
    db0, db2 = g0["generator/encoder/encoder_conv8/bias"].ravel(), g2["generator/encoder/encoder_conv8/bias"].ravel()
    assert np.argmax(np.abs(db2 - db0)) == 3
