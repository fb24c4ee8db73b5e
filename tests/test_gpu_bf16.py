"""bf16 activation storage (BASELINE configs[4]; the *_bf16 entry points of include/cape_hip.h): the mesh activations and
their gradients live in HBM as bf16, contractions use bf16 operands with fp32 accumulation, variables and every reduction
stay fp32.  Parity bar: <= 2e-2 relative against the fp32 / fp64 evaluation of the same graph --
per-vertex L2 / max-norm for the forward pass of single operators and of the full CAPE-affineconv_nz64 model; the
backward kernels exactly (<= 1e-2) against torch on identical bf16 inputs, and against the fp64 graph within the
activation-flip noise bf16 rounding causes (documented at the assertion); the kernels the library selects are asserted too (the matrix-pipe kernel with ONE bf16 product for eligible
launches, the generic gather kernel elsewhere)."""
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-2
BF16_GRAD_TOL = 1e-2      # gradients of the bf16 network against fp64 on the same activation pattern (relative L2; measured 2.5e-3 global, 8.8e-3 worst variable)


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

    # Forward: per-vertex, 2e-2.  Gradients: a (leaky-)ReLU unit whose pre-activation lies within the bf16 rounding error
    # of the forward pass (~0.3 % of its scale: ~0.25 % of all units) takes the other branch than in the fp64 evaluation and
    # its gradient changes by 80-100 % -- the backward pass is exact for the activation pattern the bf16 forward produced
    # (tests below: every backward kernel against torch on identical bf16 inputs), but against the fp64 pattern that is a
    # ~4 % relative L2 deviation.  Layers with a sign-dependent derivative are therefore held to 8e-2 in relative L2 over
    # the whole tensor, smooth layers (no activation here) to 2e-2 per vertex / max-norm.
    l2 = lambda a, r: float(np.sqrt(((np.asarray(a, np.float64) - r) ** 2).sum() / max((np.asarray(r, np.float64) ** 2).sum(), 1e-300)))
    sign_dep = affine or act is not None
    errs = dict(fwd=vertex_err(hy.detach().float().cpu().numpy(), ty.detach().numpy()))
    pairs = [("dx", hx32.grad, tx.grad), ("dW", hW.grad, tW.grad)]
    if affine:
        pairs.append(("dWa", hWa.grad, tWa.grad))
    if b is not None:
        pairs.append(("db", hb.grad, tb.grad))
    if Cci:
        pairs.append(("dcond", hci.grad, tci.grad))
    for k, h, r in pairs:
        h, r = h.cpu().numpy(), r.numpy()
        errs[k] = l2(h, r) if sign_dep else (vertex_err(h, r) if k == "dx" else mat_err(h, r))
    print(name, "sign-dependent" if sign_dep else "smooth", {k: "%.2e" % v for k, v in errs.items()}, sorted(plans))
    assert errs["fwd"] < TOL, errs
    for k, v in errs.items():
        assert v < (8e-2 if (sign_dep and k != "fwd") else TOL), (k, v)


def test_bf16_backward_kernels_on_identical_inputs(mesh_ops):
    """Every kernel of a conv layer's backward pass in bf16 storage against torch fp32 arithmetic on the SAME bf16 tensors
    (no activation-pattern ambiguity): backward-prep (dz, bias gradient), the all-orders data-gradient contraction with
    the de-interleaving epilogue, the summed / separate operator applications, the weight gradient.  <= 1e-2 max-norm
    (one bf16 rounding of the outputs)."""
    from cape_amd import ops
    from cape_amd.graph import ConvOperators
    dev = torch.device("cuda:0")
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    rel = lambda a, r: float((a.float() - r.float()).abs().max() / r.float().abs().max().clamp_min(1e-30))
    for (lvl, pool_i, N, Ch, Fout, K) in ((0, 1, 4, 64, 64, 2), (4, 4, 16, 128, 256, 2)):
        dops = ops.DeviceConvOps(ConvOperators(mesh_ops["L"][lvl], K, pool=mesh_ops["D"][pool_i]), dev)
        Mo, Mi = dops.Mo, dops.Mi
        mk = lambda *s: torch.randn(s, generator=g).to(dev).to(bf)
        gy, y, x = mk(N, Mo, Fout), mk(N, Mo, Fout), mk(N, Mo, Ch)
        W = (0.1 * torch.randn((Ch * K, Fout), generator=g)).to(dev)
        act_ = lambda t: ops.alloc_act(t.shape[0], t.shape[1], t.shape[2], dev, dtype=bf).copy_(t)
        dz, dbv, _, _ = ops.bwd_prep(act_(gy), y=act_(y), act="leaky", want_bias=True)
        ref_dz = gy.float() * torch.where(y.float() > 0, 1.0, 0.2)
        assert dz.dtype == bf and rel(dz, ref_dz) < 1e-2 and rel(dbv, ref_dz.sum((0, 1))) < 1e-3
        ChP = (Ch + 3) // 4 * 4
        Gall = ops.alloc_act(N, Mo, K * ChP, dev, dtype=bf)
        ops.gconv_fwd([dict(x=dz, csr=None, w=(W, 0, 1, Fout))], Gall, deinterleave=K, F=K * Ch)
        Gref = dz.float() @ W.t()                          # (weights enter as two bf16 planes hi + mid: 16 significant bits)
        Gs = [Gall[:, :, k * ChP:k * ChP + Ch] for k in range(K)]
        for k in range(K):
            assert rel(Gs[k], Gref[:, :, k::K]) < 1e-2, k
        dx = ops.spmm_multi(Gs, [dops.bwd[k] for k in range(K)], sum=True)
        ref = torch.zeros((N, Mi, Ch), device=dev)
        for k in range(K):
            h = dops.host.bwd[k]
            if h.identity:
                ref += Gs[k].float()
                continue
            S = torch.sparse_csr_tensor(torch.from_numpy(h.rowptr.astype(np.int64)), torch.from_numpy(h.colidx.astype(np.int64)),
                                        torch.from_numpy(h.vals), size=h.shape).to(dev)
            for n in range(N):
                ref[n] += S @ Gs[k][n].float()
        assert dx.dtype == bf and tuple(dx.shape) == (N, Mi, Ch) and rel(dx, ref) < 1e-2
        dW = torch.empty((Ch, Fout), device=dev)
        ops.gconv_dw([dict(x=act_(x), csr=None, w=(dW, 0, Fout, 1))], dz)
        assert rel(dW, torch.einsum('nrc,nrf->cf', x.float(), dz.float())) < 1e-3


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
    pred, ref = out['prediction'].detach().cpu().numpy().astype(np.float64), xh.detach().numpy()
    e_pred = T.vertex_err(pred, ref)                                       # worst vertex of 2 x 6890
    e_pred_l2 = float(np.sqrt(((pred - ref) ** 2).sum() / (ref ** 2).sum()))
    e_zm = T.rel_err(out['z_mean'].detach().cpu().numpy(), zm.detach().numpy())
    e_zl = T.rel_err(out['z_logvar'].detach().cpu().numpy(), zl.detach().numpy())
    print("bf16 storage forward: prediction relative L2 %.2e, worst vertex %.2e;  z_mean %.2e  z_logvar %.2e" % (e_pred_l2, e_pred, e_zm, e_zl))
    # SURVEY 8c asks <= 2e-2 relative vs the fp32 path.  With the fp32 master weights rounded to ONE bf16 plane the
    # reconstruction after the whole chain (35 stacked layers) measured 2.0e-2 relative L2 / 2.35e-2 at the worst vertex on
    # these inputs -- the weight rounding, identical for every vertex and sample, was the dominant and coherent part
    # (tools/diag_bf16_error.py: 6.3e-3 with exact weights).  The contractions now take the weights as two bf16 planes.
    assert out['prediction'].dtype == torch.float32 and e_pred_l2 < TOL and e_pred < TOL and e_zm < TOL and e_zl < TOL
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
    # the gradient of the bf16 network for ITS activation pattern, compared here with the fp64 pattern (units within the bf16
    # forward error of zero take the other branch); measured 7.4e-3 with the two-plane weights (0.2 when the weights were
    # rounded to one bf16 plane).  The pattern-pinned comparison is test_bf16_batch16_parity_covers_every_bench_kernel.
    assert gl < 3e-2, gl


def _grad_errors(model, twin, out, ls):
    g_names, d_names = model._g_names, model._d_names
    tg = torch.autograd.grad(ls['loss_g'], [twin.params[n] for n in g_names], retain_graph=True, allow_unused=True)
    td = torch.autograd.grad(ls['loss_d'], [twin.params[n] for n in d_names], allow_unused=True)
    hg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in g_names], retain_graph=True, allow_unused=True)
    hd = torch.autograd.grad(out['loss_d'], [model._vars[n] for n in d_names], allow_unused=True)
    num = den = 0.0
    rows = []
    for names, tgr, hgr in ((g_names, tg, hg), (d_names, td, hd)):
        for n, a, b in zip(names, tgr, hgr):
            if a is None:
                continue
            a64, b64 = a.numpy(), b.cpu().numpy().astype(np.float64)
            e2, r2 = ((b64 - a64) ** 2).sum(), (a64 ** 2).sum()
            rows.append((n, float(np.sqrt(e2 / max(r2, 1e-300))), r2))
            num += e2
            den += r2
    return float(np.sqrt(num / den)), rows, den


def test_bf16_batch16_parity_covers_every_bench_kernel(mesh_ops):
    """BASELINE configs[4]'s per-GPU shard AT its size (16 meshes, bf16 activation storage): the library selects its
    128 x 128 one-product tiles (gemm_split_kernel<128,128,*,unsigned short>, dw_split_kernel<128,128,unsigned short>) only
    there.  (1) forward against the golden vectors the reference's own lib/models.py produced at batch 16 and against the
    fp64 twin: <= 2e-2, per-vertex and in relative L2, latent heads and every loss; (2) gradients against the
    fp64 twin evaluated on the activation pattern the bf16 forward took (no branch ambiguity, see tests/test_gpu_model.py):
    every variable and the whole bucket in relative L2; (3) every kernel instantiation of the step ``bench.py --dtype bf16``
    times must have been launched by (1) and (2)."""
    import test_gpu_model as T
    from cape_amd import ops
    g, meta, N, inputs = T._golden_batch_inputs("affine_nz64_b16", mesh_ops)
    assert N == 16
    P, twin, model = T._build("affine_nz64", mesh_ops, N, dict(act_dtype='bf16'))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = inputs
    xh, zm, zl, d_real, d_fake, ls = T._run_twin(twin, *inputs)
    model.load_variables(twin.vs.vars)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
    ops.PLAN_LOG, ops.ACT_TRACE, ops.L1_SIGN_TRACE = set(), [], []
    try:
        out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
        signs, l1 = list(ops.ACT_TRACE), list(ops.L1_SIGN_TRACE)
        ops.ACT_TRACE = ops.L1_SIGN_TRACE = None
        _, _, _, _, _, lsm = T._run_twin(twin, *inputs, signs=signs, l1_sign=l1[0].numpy() if l1 else None)
        gl, rows, den = _grad_errors(model, twin, out, lsm)
        torch.cuda.synchronize()
        parity_plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = ops.ACT_TRACE = ops.L1_SIGN_TRACE = None
    assert parity_plans and all(p[-1] == "bf16" for p in parity_plans), parity_plans
    pred = out['prediction'].detach().cpu().numpy().astype(np.float64)
    for name, ref in (("reference golden (batch 16)", g["out_op_prediction"].astype(np.float64)), ("fp64 twin", xh.detach().numpy())):
        e_v = T.vertex_err(pred, ref)
        e_l2 = float(np.sqrt(((pred - ref) ** 2).sum() / (ref ** 2).sum()))
        print("bf16 storage, batch 16, prediction vs %s: worst vertex %.2e, relative L2 %.2e" % (name, e_v, e_l2))
        assert e_v < TOL and e_l2 < TOL, (name, e_v, e_l2)
    assert T.rel_err(out['z_mean'].detach().cpu().numpy(), g["out_z_mean"]) < TOL
    assert T.rel_err(out['z_logvar'].detach().cpu().numpy(), g["out_z_logvar"]) < TOL
    for key, name in (("recon", "recon_loss"), ("latent", "latent_loss"), ("edge", "edge_loss"),
                      ("gan_g", "loss_g"), ("gan_d", "loss_d"), ("loss_g", "op_loss_g"), ("loss_d", "op_loss_d")):
        assert abs(float(out[key]) - float(g["out_" + name])) < TOL * max(abs(float(g["out_" + name])), 1e-3), key
    worst = sorted(rows, key=lambda r: -r[1])[:6]
    print("bf16 storage, batch 16, gradients on the device's activation pattern (%d of %d units differ from fp64): global "
          "relative L2 %.2e; worst variables %s" % (sum(twin.flip_log), sum(int(s_.numel()) for s_ in signs), gl,
                                                     ", ".join("%s %.2e" % (n.split('/', 1)[-1], e) for n, e, _ in worst)))
    assert gl < BF16_GRAD_TOL, gl
    for n, e, r2 in rows:
        if r2 > 1e-8 * den:
            assert e < 2 * BF16_GRAD_TOL, (n, e)
    del out
    bench_plans = T._bench_step_plans(model, inputs, (False,))
    missing = bench_plans - parity_plans
    assert not missing, "kernels launched by the benchmarked bf16 step without a parity case: %s" % T._plan_names(missing)
    for need in (("fwd", 2, 128, 128, 0, 0, "bf16"), ("fwd", 2, 128, 128, 1, 0, "bf16"), ("dw", 3, 128, 128, "bf16")):
        assert need in bench_plans, (need, sorted(bench_plans))
    print("kernel instantiations of the benchmarked bf16 step, all covered at batch 16:", T._plan_names(bench_plans))


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
