"""bf16 activation storage (BASELINE configs[4]; the *_bf16 entry points of include/cape_hip.h): the mesh activations and
their gradients live in HBM as bf16, contractions use bf16 operands with fp32 accumulation, variables and every reduction
stay fp32.  Parity bar: <= 2e-2 relative against the fp32 / fp64 evaluation of the same graph --
per-vertex L2 for tensors, max-norm for weight gradients -- for single operators and for the full CAPE-affineconv_nz64
model; the kernels the library selects are asserted too (the matrix-pipe kernel with ONE bf16 product for eligible
launches, the generic gather kernel elsewhere)."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-2


def vertex_err(a, ref):
    a = np.asarray(a, dtype=np.float64).reshape(-1, ref.shape[-1])
    r = np.asarray(ref, dtype=np.float64).reshape(-1, ref.shape[-1])
    return np.sqrt(((a - r) ** 2).sum(-1)).max() / max(np.sqrt((r * r).sum(-1)).max(), 1e-30)


def mat_err(a, ref):
    a, r = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)


CASES = [
    # name,          level, N, Cin, Fout, K, act,           bias,      pool, unpool, Cc(in), affine, expected fwd family
    ("enc_conv2",        0, 4, 64, 64, 2, "b1leakyrelu", "channel", 1, None, 0, False, 2),
    ("enc_conv1_in3",    0, 3, 3, 64, 2, "b1leakyrelu", "channel", 0, None, 0, False, 0),
    ("enc_conv5",        4, 2, 128, 256, 2, "b1relu", "channel", 4, None, 0, False, 2),
    ("onebyone",         8, 3, 64, 512, 1, None, None, None, None, 0, False, 2),
    ("affine_blk7_cin",  1, 2, 64, 32, 2, None, None, None, 1, 64, True, None),
    ("affine_blk2_cin",  6, 2, 256, 256, 2, None, None, None, 6, 64, True, 2),
    ("affine_blk3_cin",  5, 2, 256, 128, 2, None, None, None, 5, 64, True, 2),      # up-sampling block (coarse form)
    ("out_conv_cin",     0, 2, 32, 3, 2, None, "vertex", None, None, 64, False, 0),
    ("disc_conv1_cin",  "d0", 2, 3, 64, 3, "b1leakyrelu", "channel", "d0", None, 64, False, 0),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_cheb_conv_bf16(case, mesh_ops):
    from cape_amd import ops
    from cape_amd.graph import ConvOperators
    from test_gpu_ops import _twin_conv
    name, level, N, Cin, Fout, K, act, bias_kind, pool_i, unpool_i, Cci, affine, want_family = case
    dev = torch.device("cuda:0")
    L = mesh_ops["L_d"][int(level[1:])] if isinstance(level, str) else mesh_ops["L"][level]
    pool = None
    if pool_i is not None:
        pool = mesh_ops["D_d"][int(pool_i[1:])] if isinstance(pool_i, str) else mesh_ops["D"][pool_i]
    unpool = mesh_ops["U"][unpool_i] if unpool_i is not None else None
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    Mi = unpool.shape[1] if unpool is not None else L.shape[0]
    Mo = pool.shape[0] if pool is not None else L.shape[0]
    x = rng.standard_normal((N, Mi, Cin))
    W = 0.1 * rng.standard_normal(((Cin + Cci) * K, Fout))
    W_aff = 0.1 * rng.standard_normal((Cin + Cci, Fout)) if affine else None
    cond_in = rng.standard_normal((N, Cci)) if Cci else None
    b = None
    if bias_kind == "channel":
        b = 0.1 * rng.standard_normal((1, 1, Fout))
    elif bias_kind == "vertex":
        b = 0.1 * rng.standard_normal((1, L.shape[0], Fout))
    gy = rng.standard_normal((N, Mo, Fout))

    t = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tx, tW, tWa, tb, tci = t(x), t(W), t(W_aff), t(b), t(cond_in)
    ty = _twin_conv(tx, L, tW, K, tb, act, pool=pool, unpool=unpool, W_aff=tWa, cond_in=tci)
    ty.backward(torch.tensor(gy, dtype=torch.float64))

    g = lambda a: None if a is None else torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=True)
    hx32, hW, hWa, hb, hci = g(x), g(W), g(W_aff), g(b), g(cond_in)
    hx = hx32.to(torch.bfloat16)                     # the cast is differentiable: hx32.grad arrives as fp32
    dops = ops.DeviceConvOps(ConvOperators(L, K, unpool=unpool, pool=pool), dev)
    ops.PLAN_LOG = set()
    try:
        hy = ops.chebyshev5(hx, hW, dops, bias=hb, activation=act, W_affine=hWa, cond_in=hci)
        assert hy.dtype == torch.bfloat16 and tuple(hy.shape) == (N, Mo, Fout)
        hy.backward(torch.tensor(gy, dtype=torch.float32, device=dev).to(torch.bfloat16))
        torch.cuda.synchronize()
        plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    assert all(p[-1] == "bf16" for p in plans), plans
    if want_family is not None:
        assert any(p[0] == "fwd" and p[1] == want_family for p in plans), (want_family, plans)

    errs = dict(fwd=vertex_err(hy.detach().float().cpu().numpy(), ty.detach().numpy()),
                dx=vertex_err(hx32.grad.cpu().numpy(), tx.grad.numpy()),
                dW=mat_err(hW.grad.cpu().numpy(), tW.grad.numpy()))
    if affine:
        errs["dWa"] = mat_err(hWa.grad.cpu().numpy(), tWa.grad.numpy())
    if b is not None:
        errs["db"] = mat_err(hb.grad.cpu().numpy(), tb.grad.numpy())
    if Cci:
        errs["dcond"] = mat_err(hci.grad.cpu().numpy(), tci.grad.numpy())
    print(name, {k: "%.2e" % v for k, v in errs.items()}, sorted(plans))
    for k, v in errs.items():
        assert v < TOL, (k, v)


def test_spmm_bf16(mesh_ops):
    import scipy.sparse as sp
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for P, C in ((mesh_ops["U"][1], 24), (mesh_ops["D"][3], 7), (mesh_ops["U_d"][3], 64)):
        P64 = sp.csr_matrix(P, dtype=np.float64)
        x = rng.standard_normal((3, P.shape[1], C))
        fwd, bwd = ops.DeviceCSR(HostCSR(P64), dev), ops.DeviceCSR(HostCSR(P64.T), dev)
        hx32 = torch.tensor(x, dtype=torch.float32, device=dev, requires_grad=True)
        hy = ops.poolwT(hx32.to(torch.bfloat16), fwd, bwd)
        assert hy.dtype == torch.bfloat16
        gy = rng.standard_normal(tuple(hy.shape))
        hy.backward(torch.tensor(gy, dtype=torch.float32, device=dev).to(torch.bfloat16))
        ref = np.stack([P64 @ x[n] for n in range(3)])
        refg = np.stack([P64.T @ gy[n] for n in range(3)])
        assert vertex_err(hy.detach().float().cpu().numpy(), ref) < TOL
        assert vertex_err(hx32.grad.cpu().numpy(), refg) < TOL


def test_full_model_bf16_storage(mesh_ops):
    """CAPE-affineconv_nz64 + discriminator with bf16 activation storage against the fp64 twin: forward within 2e-2
    (per-vertex / max-norm), every loss within 2e-2, gradients: global relative L2 error and per-variable max-norm
    bounded (bf16 rounding of ~35 stacked layers; reported)."""
    import test_gpu_model as T
    N = 2
    P, twin, model = T._build("affine_nz64", mesh_ops, N, dict(act_dtype='bf16'))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = T._inputs(N, P["nz"])
    xh, zm, zl, d_real, d_fake, ls = T._run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps)
    model.load_variables(twin.vs.vars)
    dev = model.device
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    from cape_amd import ops
    ops.PLAN_LOG = set()
    try:
        out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
        g_names, d_names = model._g_names, model._d_names
        hg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in g_names], retain_graph=True, allow_unused=True)
        hd = torch.autograd.grad(out['loss_d'], [model._vars[n] for n in d_names], allow_unused=True)
        torch.cuda.synchronize()
        plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    assert plans and all(p[-1] == "bf16" for p in plans), plans
    assert any(p[0] == "fwd" and p[1] == 2 for p in plans) and any(p[0] == "dw" and p[1] == 3 for p in plans), plans
    e_pred = T.vertex_err(out['prediction'].detach().cpu().numpy(), xh.detach().numpy())
    e_zm = T.rel_err(out['z_mean'].detach().cpu().numpy(), zm.detach().numpy())
    e_zl = T.rel_err(out['z_logvar'].detach().cpu().numpy(), zl.detach().numpy())
    print("bf16 storage forward: prediction %.2e  z_mean %.2e  z_logvar %.2e" % (e_pred, e_zm, e_zl))
    assert out['prediction'].dtype == torch.float32 and e_pred < TOL and e_zm < TOL and e_zl < TOL
    for k in ('recon', 'latent', 'edge', 'gan_g', 'gan_d', 'loss_g', 'loss_d'):
        assert abs(float(out[k]) - float(ls[k])) < TOL * max(abs(float(ls[k])), 1e-3), (k, float(out[k]), float(ls[k]))
    tg = torch.autograd.grad(ls['loss_g'], [twin.params[n] for n in g_names], retain_graph=True, allow_unused=True)
    td = torch.autograd.grad(ls['loss_d'], [twin.params[n] for n in d_names], allow_unused=True)
    num = den = 0.0
    worst = ("", 0.0)
    for names, tgr, hgr in ((g_names, tg, hg), (d_names, td, hd)):
        for n, a, b in zip(names, tgr, hgr):
            if a is None:
                continue
            a64, b64 = a.numpy(), b.cpu().numpy().astype(np.float64)
            num += ((b64 - a64) ** 2).sum()
            den += (a64 ** 2).sum()
            e = T.rel_err(b64, a64)
            if e > worst[1]:
                worst = (n, e)
    gl = np.sqrt(num / den)
    print("bf16 storage gradients: global relative L2 error %.2e; worst variable (max-norm) %s %.2e" % (gl, worst[0], worst[1]))
    assert gl < 3e-2, gl
    assert worst[1] < 0.15, worst


def test_bf16_train_steps_reduce_the_loss(mesh_ops):
    """Three captured training steps in bf16 storage: finite losses, the generator loss goes down on a fixed batch."""
    import test_gpu_model as T
    from cape_amd.runtime import GraphedTrainStep
    N = 2
    P, twin, model = T._build("affine_nz64", mesh_ops, N, dict(act_dtype='bf16', lr_warmup=False, decay_steps=1000, lr=2e-3))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = T._inputs(N, P["nz"])
    runner = GraphedTrainStep(model, with_gan=False)
    runner.load_batch(data_g=x, cond_g=cond, cond2_g=clo, gt=gt, data_d=xd, cond_d=cond_d, cond2_d=clo_d, eps=eps)
    runner.capture(preserve_state=True)
    losses = []
    for _ in range(4):
        runner.step()
        losses.append(float(runner.losses['loss_g']))
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
