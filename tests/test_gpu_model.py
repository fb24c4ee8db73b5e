"""Full-network GPU parity: cape_amd.models.CAPE (HIP kernels through the C-ABI) against the CPU
oracle (numpy fp64 restatement of reference lib/models.py + torch autograd twin) with identical,
name-keyed weights and inputs.

Tolerances (fp32 path; SURVEY section 8c, literally): every error against the fp64 oracle <= 4 x the error the fp32
restatement of the reference's op order makes on the same inputs and measure (the same twin in float32 on the CPU;
tests/parity_bar.py records the measured ratios), and in absolute terms per-vertex L2 error of the reconstruction <= 1e-4 x
the largest per-vertex L2 norm of the fp64 oracle output; latent codes / logits 1e-4 relative (max-norm).

Parameter gradients are compared with the fp64 twin evaluated ON THE DEVICE'S ACTIVATION PATTERN: a (leaky-)ReLU unit
whose pre-activation lies within fp32 rounding of zero takes the other branch in a different fp32 evaluation and moves
whole gradients by 1e-4..1e-2 -- noise that says nothing about the kernels.  The device forward records the branch every
unit took (cape_amd.ops.ACT_TRACE) and the twin replays it (oracle.torch_twin.forced_act), so both sides differentiate the
same piecewise-linear function: every variable's gradient must then agree to GRAD_TOL in relative L2 and the bucket as a
whole likewise (no flip-noise allowance; the count of replayed flips is printed).
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def vertex_err(a, ref):
    a = np.asarray(a, np.float64).reshape(-1, ref.shape[-1])
    r = np.asarray(ref, np.float64).reshape(-1, ref.shape[-1])
    return np.sqrt(((a - r) ** 2).sum(-1)).max() / max(np.sqrt((r * r).sum(-1)).max(), 1e-30)


def rel_err(a, ref):
    a, r = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)


def _inputs(N, nz, seed=0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, 6890, 3))
    gt = x + 0.1 * rng.standard_normal((N, 6890, 3))
    xd = rng.standard_normal((N, 6890, 3))
    cond = 0.5 * rng.standard_normal((N, 126))
    cond_d = 0.5 * rng.standard_normal((N, 126))
    clo = np.eye(4)[np.arange(N) % 4]
    clo_d = np.eye(4)[(np.arange(N) + 1) % 4]
    eps = rng.standard_normal((N, nz))
    return x, gt, xd, cond, cond_d, clo, clo_d, eps


def _twin(cfg, mesh_ops, N, overrides=None, tdtype=torch.float64):
    from oracle.configs import cape_params
    from oracle.torch_twin import TwinCAPE
    P = cape_params(cfg, N)
    P.update(overrides or {})
    m = mesh_ops
    pack = m["pack"]
    return P, TwinCAPE(m["L"], m["D"], m["U"], m["L_d"], m["D_d"], p=m["p"], dtype=np.float64, tdtype=tdtype,
                       verts_ref=pack["template_verts"], vpe=pack["edges_smpl"], **P)


def _build(cfg, mesh_ops, N, overrides=None):
    from cape_amd.models import CAPE
    P, twin = _twin(cfg, mesh_ops, N, overrides)
    m = mesh_ops
    model = CAPE(L=m["L"], D=m["D"], U=m["U"], L_d=m["L_d"], D_d=m["D_d"], p=m["p"], **P)
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='train')
    return P, twin, model


def _run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps, signs=None, l1_sign=None):
    """``signs``: branch patterns recorded by the device forward (ops.ACT_TRACE), replayed site by site -- the sub-networks
    run in the order cape_amd.models.CAPE.forward_losses evaluates them."""
    import collections
    merged = False
    if signs is not None:
        # the device evaluates D(generated) and D(real) as one pass over the concatenated batch (cape_amd.ops.MERGED_D_PASS):
        # its discriminator sites carry 2N rows -- first half the generated samples, second half the real ones
        nb = x.shape[0]
        merged = any(m.shape[0] == 2 * nb for m in signs)
        if merged:
            dsites = [m for m in signs if m.shape[0] == 2 * nb]
            signs = [m for m in signs if m.shape[0] != 2 * nb] + [m[:nb] for m in dsites] + [m[nb:] for m in dsites]
    twin.forced_signs = None if signs is None else collections.deque(signs)
    twin.flip_log = []
    try:
        y, y2 = twin.cond_embeddings(cond, clo)
        xh, zm, zl = twin.generator(x, y, y2, eps)
        if merged:                              # device order: both condition networks, then the discriminator
            yd, y2d = twin.cond_embeddings(cond_d, clo_d)
            d_fake = twin.discriminator(xh, y, y2)
        else:
            d_fake = twin.discriminator(xh, y, y2)
            yd, y2d = twin.cond_embeddings(cond_d, clo_d)
        d_real = twin.discriminator(xd, yd, y2d)
        assert not twin.forced_signs, "%d recorded activation sites were not consumed" % len(twin.forced_signs)
    finally:
        twin.forced_signs = None
    twin.forced_l1_sign = l1_sign              # the L1 loss' sign(pred - gt), the graph's one other branch point
    try:
        ls = twin.losses(xh, gt, zm, zl, d_real, d_fake)
    finally:
        twin.forced_l1_sign = None
    return xh, zm, zl, d_real, d_fake, ls


GRAD_TOL = 2e-5          # relative L2, per variable and over the whole bucket, on the device's activation pattern


CONFIGS = [
    ("affine_nz64", None),
    ("cmr_nz18", None),
    # encoder res_block (:715-741) + plain udn decoder (:173-191) + conditioned encoder, tanh
    ("affine_nz18", dict(use_res_block=True, use_res_block_dec=False, cond_encoder=True, activation='b1tanh',
                         F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=32, loss='l2')),
    # polynomial orders 2-6 mixed over the layers, affine res-blocks: precomposed operators (K <= 3) next to the explicit
    # recurrence (:88-96) with materialised condition channels, the affine block fused and composed
    ("affine_nz18", dict(K=[3, 4, 3, 4, 2, 5, 3, 6], F=[16, 16, 32, 32, 64, 64, 128, 128], reduce_dim=16)),
]


@pytest.mark.parametrize("cfg,overrides", CONFIGS, ids=["affine_nz64", "cmr_nz18", "resblock_udn_tanh", "affine_mixed_k"])
def test_full_model_forward_backward(cfg, overrides, mesh_ops):
    _full_model_parity(cfg, overrides, mesh_ops, N=2)


def _full_model_parity(cfg, overrides, mesh_ops, N, inputs=None):
    """Forward values, all losses and every parameter gradient of the HIP model against the fp64 twin (same named
    weights, same inputs).  Returns the model (variables loaded) for further use."""
    from cape_amd import ops
    P, twin, model = _build(cfg, mesh_ops, N, overrides)
    x, gt, xd, cond, cond_d, clo, clo_d, eps = inputs if inputs is not None else _inputs(N, P["nz"])
    xh, zm, zl, d_real, d_fake, ls = _run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps)
    # same variable set (names + shapes) as the restated reference graph, then identical values
    assert set(model._vars) == set(twin.vs.vars), set(model._vars) ^ set(twin.vs.vars)
    model.load_variables(twin.vs.vars)

    dev = model.device
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    ops.ACT_TRACE, ops.L1_SIGN_TRACE = [], []
    try:
        out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
        signs, l1 = list(ops.ACT_TRACE), list(ops.L1_SIGN_TRACE)
    finally:
        ops.ACT_TRACE = ops.L1_SIGN_TRACE = None
    l1_sign = l1[0].numpy() if l1 else None
    # the fp32 restatement (same twin in float32, same named weights) on the device's activation pattern: the noise floor of
    # SURVEY 8(c)'s bar, forward values and gradients from one evaluation
    from parity_bar import check
    tag = "model[%s%s,N=%d]" % (cfg, "" if not overrides else "+" + ",".join(sorted(overrides)), N)
    P32, twin32 = _twin(cfg, mesh_ops, N, overrides, tdtype=torch.float32)
    xh32, zm32, zl32, _, _, ls32 = _run_twin(twin32, x, gt, xd, cond, cond_d, clo, clo_d, eps, signs=signs, l1_sign=l1_sign)
    assert all(np.array_equal(twin32.vs.vars[n], twin.vs.vars[n]) for n in twin.vs.vars), "fp32 twin drew other weights"
    n64 = lambda v: v.detach().cpu().numpy().astype(np.float64)
    check(tag, "prediction", vertex_err(n64(out['prediction']), n64(xh)), vertex_err(n64(xh32), n64(xh)), 1e-4)
    check(tag, "z_mean", rel_err(n64(out['z_mean']), n64(zm)), rel_err(n64(zm32), n64(zm)), 1e-4)
    check(tag, "z_logvar", rel_err(n64(out['z_logvar']), n64(zl)), rel_err(n64(zl32), n64(zl)), 1e-4)
    for k in ('recon', 'latent', 'edge', 'gan_g', 'gan_d', 'loss_g', 'loss_d'):
        assert abs(float(out[k]) - float(ls[k])) < 1e-4 * max(abs(float(ls[k])), 1e-3), k

    # gradients: loss_g w.r.t. generator+condition variables, loss_d w.r.t. discriminator variables -- the twin on the
    # activation pattern the device forward took (see the module docstring)
    _, _, _, _, _, lsm = _run_twin(twin, x, gt, xd, cond, cond_d, clo, clo_d, eps, signs=signs, l1_sign=l1_sign)
    nflip, nunits = sum(twin.flip_log), sum(int(s_.numel()) for s_ in signs)
    for k in ('loss_g', 'loss_d'):          # a flipped unit has |z| ~ 1e-7: the forward value does not move
        assert abs(float(lsm[k]) - float(ls[k])) < 1e-6 * max(abs(float(ls[k])), 1e-3), k
    g_names, d_names = model._g_names, model._d_names
    tg = torch.autograd.grad(lsm['loss_g'], [twin.params[n] for n in g_names], retain_graph=True, allow_unused=True)
    td = torch.autograd.grad(lsm['loss_d'], [twin.params[n] for n in d_names], allow_unused=True)
    hg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in g_names], retain_graph=True, allow_unused=True)
    hd = torch.autograd.grad(out['loss_d'], [model._vars[n] for n in d_names], allow_unused=True)
    fg = torch.autograd.grad(ls32['loss_g'], [twin32.params[n] for n in g_names], retain_graph=True, allow_unused=True)
    fd = torch.autograd.grad(ls32['loss_d'], [twin32.params[n] for n in d_names], allow_unused=True)
    rows, num, den, num32, worst32 = [], 0.0, 0.0, 0.0, 0.0
    for names, tgr, hgr, fgr in ((g_names, tg, hg, fg), (d_names, td, hd, fd)):
        for n, a, b, c in zip(names, tgr, hgr, fgr):
            if a is None:
                assert b is None or float(b.abs().max()) == 0.0, n
                continue
            a64, b64, c64 = a.numpy(), b.cpu().numpy().astype(np.float64), c.numpy().astype(np.float64)
            e2, r2, f2 = ((b64 - a64) ** 2).sum(), (a64 ** 2).sum(), ((c64 - a64) ** 2).sum()
            rows.append((n, np.sqrt(e2 / max(r2, 1e-300)), rel_err(b64, a64), e2, r2, np.sqrt(f2 / max(r2, 1e-300))))
            num += e2
            den += r2
            num32 += f2
    gl = np.sqrt(num / den)
    judged = [r for r in rows if r[4] > 1e-16 * den]
    check(tag, "gradient, whole bucket (rel L2)", gl, np.sqrt(num32 / den), GRAD_TOL)
    check(tag, "gradient, worst variable (rel L2)", max(r[1] for r in judged), max(r[5] for r in judged), GRAD_TOL)
    print("gradient error on the device's activation pattern (%d of %d units differ from the fp64 pattern, %d sites): "
          "worst variable %.3g relative L2, global %.3g" % (nflip, nunits, len(signs), max(r[1] for r in rows), gl))
    for r in sorted(rows, key=lambda r: -r[1])[:6]:
        print("   %-58s rel L2 %.2e  max-norm %.2e  share of global err^2 %.2f" % (r[0], r[1], r[2], r[3] / max(num, 1e-300)))
    # variables whose gradient is (numerically) zero against the bucket are judged by the global figure only
    for n, e, em, e2, r2, e32 in rows:
        if r2 > 1e-16 * den:
            assert e < GRAD_TOL, (n, e, em)
    assert gl < GRAD_TOL, gl
    return model, out


def _golden_batch_inputs(tag, mesh_ops):
    from oracle.golden_inputs import golden_inputs
    from test_oracle_golden import load_case
    g, meta = load_case(tag)
    N = int(meta["N"])
    from oracle.configs import cape_params
    nz = cape_params(meta["cfg"], N)["nz"]
    in_field = None
    if meta.get("profile") == "range":
        from oracle.golden_inputs import range_fields
        in_field = range_fields(mesh_ops["pack"]["template_verts"], mesh_ops["D"])[0]
    inp = golden_inputs(N, nz, meta["seed"], mesh_ops["pack"]["demo_rot"], in_field=in_field)
    inputs = tuple(np.asarray(inp[k], np.float64) for k in ("x", "gt", "xd", "cond", "cond_d", "clo", "clo_d", "eps"))
    return g, meta, N, inputs


def _assert_matches_golden(out, g, tol=1e-4):
    assert vertex_err(out['prediction'].detach().cpu().numpy(), g["out_op_prediction"].astype(np.float64)) < tol
    assert rel_err(out['z_mean'].detach().cpu().numpy(), g["out_z_mean"]) < tol
    assert rel_err(out['z_logvar'].detach().cpu().numpy(), g["out_z_logvar"]) < tol
    for key, name in (("recon", "recon_loss"), ("latent", "latent_loss"), ("edge", "edge_loss"),
                      ("gan_g", "loss_g"), ("gan_d", "loss_d"), ("loss_g", "op_loss_g"), ("loss_d", "op_loss_d")):
        assert abs(float(out[key]) - float(g["out_" + name])) < tol * max(abs(float(g["out_" + name])), 1e-3), key


def _bench_step_plans(model, inputs, gans):
    """Kernel instantiations (as the library reports them) of the step bench.py times: eager pass of the graph runner's body."""
    from cape_amd import ops
    from cape_amd.runtime import GraphedTrainStep
    plans = set()
    for gan in gans:
        runner = GraphedTrainStep(model, with_gan=gan, use_graph=False)
        runner.load_batch(data_g=inputs[0], gt=inputs[1], data_d=inputs[2], cond_g=inputs[3], cond_d=inputs[4],
                          cond2_g=inputs[5], cond2_d=inputs[6], eps=inputs[7])
        ops.PLAN_LOG = set()
        try:
            runner._fwd_bwd()
            torch.cuda.synchronize()
            plans |= ops.PLAN_LOG
        finally:
            ops.PLAN_LOG = None
    return plans


def _plan_names(plans):
    from cape_amd import ops
    bf = lambda p: p[-1] == "bf16"
    core = lambda p: p[1:-1] if bf(p) else p[1:]
    return sorted(ops.fwd_kernel_name(*core(p), bf16=bf(p)) if p[0] == "fwd" else ops.dw_kernel_name(*core(p)[:3], bf16=bf(p))
                  for p in plans)


def test_batch16_parity_covers_every_bench_kernel(mesh_ops):
    """BASELINE configs[2] AT its stated batch: static batch 16 is part of the reference contract
    (lib/models.py:272-282, config_parser.py:33) and the library picks its large-tile kernels only at that size
    (gemm_h2_kernel<128,128>, the DUAL gemm_h2_kernel<128,64>, dw_split_kernel<128,128>).  (1) full CAPE-affineconv_nz64 + discriminator at N = 16:
    forward, losses and all gradients against the fp64 twin, forward also against the golden vectors the reference's own
    lib/models.py produced at batch 16 (oracle/make_golden.py, case affine_nz64_b16); (2) every kernel instantiation the
    library reports (cape_gconv_fwd_plan / cape_gconv_dw_plan) for the step bench.py times -- CVAE step and adversarial
    step, through the same graph runner -- must have been launched by (1)."""
    from cape_amd import ops
    g, meta, N, inputs = _golden_batch_inputs("affine_nz64_b16", mesh_ops)
    assert N == 16
    ops.PLAN_LOG = set()
    try:
        model, out = _full_model_parity("affine_nz64", None, mesh_ops, N=N, inputs=inputs)
        parity_plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    # the reference's own graph code at batch 16 (numpy TF1 shim), same weights by name (checked by CRC in
    # tests/test_oracle_golden.py), same inputs
    _assert_matches_golden(out, g)
    del out
    bench_plans = _bench_step_plans(model, inputs, (False, True))
    missing = bench_plans - parity_plans
    assert not missing, "kernels launched by the benchmarked step without a parity case: %s" % _plan_names(missing)
    # the instantiations the headline rests on: the fp16 two-piece contraction in its tile forms (family 3: the wide 128 x 256
    # tile of round 6 -- gemm_h2x_kernel --, 64 x 64, the DUAL 128 x 64 of the affine blocks; one weight layout -- the piece planes
    # are contraction-contiguous), the 128 x 128 weight-gradient kernel on the same arithmetic (family 4: dw_h2_kernel)
    from cape_amd import ops as _ops
    if _ops.H2:                             # (CAPE_H2=0, the six-product reference leg of tests/test_gpu_knobs.py, has no such launches)
        wide = os.environ.get("CAPE_H2X", "1") != "0"
        for need in (("fwd", 3, 128, 256 if wide else 128, 1, 0), ("fwd", 3, 64, 64, 1, 0), ("fwd", 3, 128, 64, 1, 1), ("dw", 4, 128, 128)):
            assert need in bench_plans, (need, sorted(bench_plans))
    print("kernel instantiations of the benchmarked step, all covered at batch 16:", _plan_names(bench_plans))


def test_nz18_gan_batch32_parity_covers_every_bench_kernel(mesh_ops):
    """BASELINE configs[3] AT its stated batch: CAPE_nz18_pose24_clotype8 (GraphCMR / group-norm decoder, reference
    lib/models.py:744-774, 681-712) + mesh-patch discriminator (:648-678) at static batch 32.  Forward, losses and every
    gradient against the fp64 twin, forward also against the golden vectors of the reference's own graph code at batch 32
    (oracle/make_golden.py, case cmr_nz18_b32); every kernel instantiation of the adversarial step
    ``bench.py --config CAPE_nz18_pose24_clotype8_male --gan --batch 32`` times must have been launched by that parity run."""
    from cape_amd import ops
    g, meta, N, inputs = _golden_batch_inputs("cmr_nz18_b32", mesh_ops)
    assert N == 32
    ops.PLAN_LOG = set()
    try:
        model, out = _full_model_parity("cmr_nz18", None, mesh_ops, N=N, inputs=inputs)
        parity_plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    _assert_matches_golden(out, g)
    del out
    bench_plans = _bench_step_plans(model, inputs, (True,))
    missing = bench_plans - parity_plans
    assert not missing, "kernels launched by the benchmarked step without a parity case: %s" % _plan_names(missing)
    print("kernel instantiations of the adversarial nz18 step at batch 32, all covered:", _plan_names(bench_plans))


def test_encode_decode_api_padding(mesh_ops):
    """numpy-in / numpy-out drivers with zero padding to batch_size (reference :931-1174)."""
    N = 4
    P, twin, model = _build("affine_nz64", mesh_ops, N)
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(6, P["nz"], seed=3)     # 6 = 4 + 2 (ragged last batch)
    _run_twin(twin, x[:1], gt[:1], xd[:1], cond[:1], cond_d[:1], clo[:1], clo_d[:1], eps[:1])   # materialise all variables
    y, y2 = twin.cond_embeddings(cond, clo)
    zm, zl = twin.encoder(x, y, y2)
    model.load_variables(twin.vs.vars)
    hz, hl, hy, hy2 = model.encode(x, cond, clo)
    assert hz.shape == (6, P["nz"]) and hy.shape == (6, P["nz_cond"]) and hy2.shape == (6, P["nz_cond2"])
    assert rel_err(hz, zm.detach().numpy()) < 1e-4 and rel_err(hl, zl.detach().numpy()) < 1e-4
    assert rel_err(hy, y.detach().numpy()) < 1e-5
    c1, c2 = model.encode_only_condition(cond, clo)
    assert np.allclose(c1, hy) and np.allclose(c2, hy2)
    # decode with ONE condition row broadcast over several z (demos.py:392-395)
    z_total = np.concatenate([zm.detach().numpy()[:5], np.repeat(hy[:1], 5, 0), np.repeat(hy2[:1], 5, 0)], 1)
    rec = model.decode(z_total, cond=hy[:1], cond2=hy2[:1])
    ref = twin.decoder_cond_vert(z_total, np.repeat(hy[:1], 5, 0), np.repeat(hy2[:1], 5, 0)).detach().numpy()
    assert rec.shape == (5, 6890, 3) and vertex_err(rec, ref) < 1e-4
    # predict with labels: loss averaging over the zero-padded last batch (:1083-1086)
    preds, lr_, ll_, le_ = model.predict(x, cond, clo, labels=gt)
    assert preds.shape == (6, 6890, 3) and np.isfinite([lr_, ll_, le_]).all()
    assert model.predict(x[:3], cond[:3], clo[:3]).shape == (3, 6890, 3)


def row_rel_err(a, ref):
    """Largest per-vertex L2 error relative to THAT vertex's own norm (rows of the reference that are exactly zero must be
    exactly zero): the measure that sees a small row next to large ones."""
    a = np.asarray(a, np.float64).reshape(-1, ref.shape[-1])
    r = np.asarray(ref, np.float64).reshape(-1, ref.shape[-1])
    n = np.sqrt((r * r).sum(-1))
    e = np.sqrt(((a - r) ** 2).sum(-1))
    assert (e[n == 0] == 0).all(), "a row that is exactly zero in the reference is not zero"
    return (e[n > 0] / n[n > 0]).max()


GOLDEN_TAGS = ["affine_nz64", "cmr_nz18", "resblock_udn_tanh", "cheb_k6", "switches_relu", "affine_mixed_k", "huber_res_affine",
               "cmr_k3_res", "reduce0", "b2relu_udn", "cond3",
               # operand range of the fp16 two-piece contractions at model level: activation rows spanning 23-39 binades in the
               # encoder and 21 at the decoder's first layers, exactly-zero rows, tanh saturated on the large rows
               "range_affine", "range_tanh"]


@pytest.mark.parametrize("tag", GOLDEN_TAGS)
def test_model_matches_reference_golden(tag, mesh_ops):
    """HIP path vs the golden vectors produced by the reference's own lib/models.py code (run on the
    numpy TF1 shim by oracle/make_golden.py): same named weights, same inputs."""
    from test_oracle_golden import load_case, build_oracle, case_profile
    from oracle.golden_inputs import golden_inputs
    from cape_amd.models import CAPE
    g, meta = load_case(tag)
    with case_profile(meta, mesh_ops) as in_field:
        P, orc = build_oracle(meta, mesh_ops)
        inp = golden_inputs(meta["N"], P["nz"], meta["seed"], mesh_ops["pack"]["demo_rot"], in_field=in_field)
        y, y2 = orc.cond_embeddings(inp["cond"], inp["clo"])            # materialises the name-keyed weights
        xh, _, _ = orc.generator(inp["x"], y, y2, inp["eps"])
        orc.discriminator(xh, y, y2)
        _, orc32 = build_oracle(meta, mesh_ops, dtype=np.float32)
        y32, y232 = orc32.cond_embeddings(inp["cond"], inp["clo"])
        xh32, zm32, zl32 = orc32.generator(inp["x"], y32, y232, inp["eps"])
    m = mesh_ops
    model = CAPE(L=m["L"], D=m["D"], U=m["U"], L_d=m["L_d"], D_d=m["D_d"], p=m["p"], **P)
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='demo')
    assert sorted(model._vars) == [str(n) for n in g["var_names"]]
    model.load_variables(orc.vs.vars)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
    with torch.no_grad():
        out = model.forward_losses(t(inp["x"]), t(inp["cond"]), t(inp["clo"]), t(inp["gt"]), t(inp["xd"]),
                                   t(inp["cond_d"]), t(inp["clo_d"]), eps=t(inp["eps"]))
    assert all(bool(torch.isfinite(out[k]).all()) for k in ('prediction', 'z_mean', 'z_logvar', 'loss_g', 'loss_d'))
    # at most 4 x the error of the fp32 restatement in the reference's op order (numpy float32 oracle, tier 2)
    # against the same golden vectors; 1e-4 in absolute terms
    from parity_bar import check
    gp = g["out_op_prediction"].astype(np.float64)
    if meta.get("profile") == "range":
        check("golden[%s]" % tag, "prediction, row-relative", row_rel_err(out['prediction'].cpu().numpy(), gp), row_rel_err(xh32, gp))
    check("golden[%s]" % tag, "prediction", vertex_err(out['prediction'].cpu().numpy(), gp), vertex_err(xh32, gp), 1e-4)
    check("golden[%s]" % tag, "z_mean", rel_err(out['z_mean'].cpu().numpy(), g["out_z_mean"]), rel_err(zm32, g["out_z_mean"]), 1e-4)
    check("golden[%s]" % tag, "z_logvar", rel_err(out['z_logvar'].cpu().numpy(), g["out_z_logvar"]), rel_err(zl32, g["out_z_logvar"]), 1e-4)
    for key, name in (("recon", "recon_loss"), ("latent", "latent_loss"), ("edge", "edge_loss"),
                      ("gan_g", "loss_g"), ("gan_d", "loss_d"), ("loss_g", "op_loss_g"), ("loss_d", "op_loss_d")):
        assert abs(float(out[key]) - float(g["out_" + name])) < 1e-4 * max(abs(float(g["out_" + name])), 1e-3), key
    # decode() driver (numpy in/out) vs the reference's op_decoder
    zt = np.concatenate([g["out_op_vae_mean"], g["out_op_cond_latent"], g["out_op_cond2_latent"]], 1)
    rec = model.decode(zt, cond=g["out_op_cond_latent"], cond2=g["out_op_cond2_latent"])
    assert vertex_err(rec, g["out_op_decoder"].astype(np.float64)) < 1e-4


def test_operand_range_forward_and_gradients(mesh_ops):
    """The "range" profile (rows spanning 23-39 binades through the encoder, exactly-zero rows, zero biases; oracle/weights.py)
    through the whole parity harness: forward values, losses and every parameter gradient against the fp64 twin, the 4x bar
    against the fp32 twin -- the fp16 two-piece contractions take their power-of-two scales from row bounds the producing
    kernels write, and an under-estimated bound would overflow to inf without any other symptom."""
    from test_oracle_golden import case_profile
    g, meta, N, inputs = _golden_batch_inputs("range_affine", mesh_ops)
    with case_profile(meta, mesh_ops):
        model, out = _full_model_parity(meta["cfg"], meta["overrides"], mesh_ops, N, inputs=inputs)
    _assert_matches_golden(out, g)


@pytest.mark.parametrize("cfg", ["affine_nz64", "cmr_nz18"])
def test_train_step_matches_manual_update(cfg, mesh_ops):
    """train_step (flat buckets, fused sampling/KL op, regularisation gradient added in the bucket, queued reductions of the
    weight / bias / group-norm parameter gradients, clip + momentum) equals a manual update computed from autograd gradients
    of the same losses."""
    N = 2
    P, twin, model = _build(cfg, mesh_ops, N, dict(regularization=0.5, lr_warmup=False, decay_steps=1000))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    dev = model.device
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    args = (t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d))
    out = model.forward_losses(*args, eps=t(eps))                       # reg inside the loss (autograd path)
    names = model._g_names + model._d_names
    params = [model._vars[n] for n in names]
    # the reference differentiates loss_g w.r.t. the generator/condition variables and loss_d w.r.t. the discriminator
    # variables (lib/models.py:447-467); D(fake) is one shared sub-graph, so the two must be taken separately
    ng = len(model._g_names)
    grads = (torch.autograd.grad(out['loss_g'], params[:ng], retain_graph=True, allow_unused=True)
             + torch.autograd.grad(out['loss_d'], params[ng:], allow_unused=True))
    before = {n: p.detach().clone() for n, p in zip(names, params)}
    expect = {}
    for grp, gn, lr in (('g', model._g_names, model.lr_g), ('d', model._d_names, model.lr_d)):
        gs = [g if g is not None else torch.zeros_like(p) for g, p, n in zip(grads, params, names) if n in gn]
        norm = torch.sqrt(sum((g.double() ** 2).sum() for g in gs))
        scale = float(5.0 / max(float(norm), 5.0))
        for n, g in zip(gn, gs):
            expect[n] = before[n] - lr * (g * scale)                    # first momentum step: accum = g
    model.train_step(*args, eps=t(eps))
    worst = 0.0
    for n in names:
        d = (model._vars[n].detach() - expect[n]).abs().max().item()
        ref = (before[n] - expect[n]).abs().max().item()
        worst = max(worst, d / max(ref, 1e-12))
        # one fp32 ulp of the weight itself is the floor for any evaluation order of w - lr*(scale*g)
        ulp = 1.2e-7 * before[n].abs().max().item()
        assert d <= 2e-3 * max(ref, 1e-12) + 1e-9 + ulp, (n, d, ref)
    assert model.global_step == 2


@pytest.mark.parametrize("gan", [False, True])
def test_two_phase_backward_equals_single_sweep(gan, mesh_ops):
    """The data-parallel step runner differentiates in two phases (decoder + dense layers first, encoder convolutions
    and condition nets second, graph cut below the dense layers) so that the early 96 % of the bucket can be exchanged
    while phase 2 runs.  Same kernels, same order inside each layer: the updated variables must equal the single-sweep
    step's, with and without HIP-graph capture."""
    from cape_amd.runtime import GraphedTrainStep
    N = 2
    finals = {}
    for split, graph in ((False, False), (True, False), (True, True)):
        P, twin, model = _build("affine_nz64", mesh_ops, N, dict(regularization=0.5, lr_warmup=False, decay_steps=1000))
        if not finals:
            x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
        assert model._opt_state['g']['split_off'] < model._opt_state['g']['flat'].numel()
        runner = GraphedTrainStep(model, with_gan=gan, use_graph=graph, split=split)
        runner.load_batch(data_g=x, cond_g=cond, cond2_g=clo, gt=gt, data_d=xd, cond_d=cond_d, cond2_d=clo_d, eps=eps)
        # one eager pass first (loads every kernel before capture; the learning-rate scalars are still 0, so it only
        # fills the momentum buffers -- identically in all three variants), then capture without further warm-up
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # (a side stream, as graph capture requires of its warm-up)
            if split:
                runner._split_step_eager()
            else:
                runner._fwd_bwd()
                runner._update()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        runner.capture(warmup=0)
        for _ in range(2):
            runner.step()
        torch.cuda.synchronize()
        finals[(split, graph)] = {g: model._opt_state[g]['flat'].detach().clone() for g in (('g', 'd') if gan else ('g',))}
    ref = finals[(False, False)]
    for key in ((True, False), (True, True)):
        for g, v in finals[key].items():
            d = (v - ref[g]).abs().max().item()
            assert d <= 1e-6 * max(ref[g].abs().max().item(), 1.0), (key, g, d)


@pytest.mark.parametrize("cfg,gan", [("affine_nz64", True), ("cmr_nz18", False)], ids=["affine_nz64_gan", "cmr_nz18"])
def test_training_step_is_bitwise_reproducible(cfg, gan, mesh_ops):
    """No kernel on the path orders a floating-point sum by atomics or by arrival (split-K slabs, column sums, group-norm
    partials and the weight-gradient reductions all add in a fixed order): three HIP-graph replays from the same state on
    the same batch must leave bit-identical variables and momentum buffers, run after run."""
    from cape_amd.runtime import GraphedTrainStep
    N = 2
    P, twin, model = _build(cfg, mesh_ops, N, dict(regularization=0.5, lr_warmup=False, decay_steps=1000))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    runner = GraphedTrainStep(model, with_gan=gan)
    runner.load_batch(data_g=x, cond_g=cond, cond2_g=clo, gt=gt, data_d=xd, cond_d=cond_d, cond2_d=clo_d, eps=eps)
    runner.capture(preserve_state=True)
    groups = ('g', 'd') if gan else ('g',)
    start = {g: {k: model._opt_state[g][k].detach().clone() for k in ('flat', 'm')} for g in groups}
    step0 = model.global_step
    runs = []
    for rep in range(2):
        with torch.no_grad():
            for g in groups:
                for k in ('flat', 'm'):
                    model._opt_state[g][k].copy_(start[g][k])
        model.global_step = step0
        for _ in range(3):
            runner.step()
        torch.cuda.synchronize()
        runs.append({(g, k): model._opt_state[g][k].detach().clone() for g in groups for k in ('flat', 'm')})
    moved = False
    for key in runs[0]:
        assert torch.equal(runs[0][key], runs[1][key]), key
        moved = moved or not torch.equal(runs[0][key], start[key[0]][key[1]])
    assert moved                                         # the steps did update something


def test_piece_planes_follow_the_weights_through_graph_replays(mesh_ops):
    """A replayed step rewrites the variables without Python noticing (no version bump, no apply_updates call), and the piece
    planes of the fp16 two-piece contractions are refreshed at the START of the captured step: after a replay they are one
    update behind.  Every pass outside the graph -- also the SECOND evaluation between replays, when no flag is left over from
    the capture -- must see planes of the current weights: the generator's output has to equal that of the same weights with
    freshly built planes."""
    from cape_amd.runtime import GraphedTrainStep
    N = 2
    P, twin, model = _build("affine_nz64", mesh_ops, N, dict(regularization=0.5, lr_warmup=False, decay_steps=1000))
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
    runner = GraphedTrainStep(model, with_gan=False)
    runner.load_batch(data_g=x, cond_g=cond, cond2_g=clo, gt=gt, data_d=xd, cond_d=cond_d, cond2_d=clo_d, eps=eps)
    runner.capture(preserve_state=True)

    def evaluate():
        with torch.no_grad():
            y, y2 = model._conditions(t(cond), t(clo))
            return model.generator(t(x), y, y2, eps=t(eps))[0].float().clone()

    for rep in range(2):                                  # replay, evaluate, replay, evaluate
        runner.step()
        runner.step()
        got = evaluate()
        model._pieces_dirty = True                        # the reference: planes rebuilt from the weights as they are now
        want = evaluate()
        assert torch.equal(got, want), (rep, float((got - want).abs().max()))


def test_merged_discriminator_pass_equals_two_passes(mesh_ops):
    """D(generated) and D(real) as one pass over the concatenated batch (ops.MERGED_D_PASS, the default up to 16 + 16 meshes)
    against the two passes the reference builds (lib/models.py:299-302): same losses, same gradients of both groups."""
    from cape_amd import ops
    N = 2
    P, twin, model = _build("affine_nz64", mesh_ops, N)
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
    args = (t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d))
    res, saved = {}, ops.MERGED_D_PASS
    try:
        for mode in ("1", "0"):
            ops.MERGED_D_PASS = mode
            out = model.forward_losses(*args, eps=t(eps))
            gg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in model._g_names], retain_graph=True, allow_unused=True)
            gd = torch.autograd.grad(out['loss_d'], [model._vars[n] for n in model._d_names], allow_unused=True)
            res[mode] = ({k: float(out[k]) for k in ('loss_g', 'loss_d', 'gan_g', 'gan_d')}, gg + gd)
    finally:
        ops.MERGED_D_PASS = saved
    for k, v in res["1"][0].items():
        assert abs(v - res["0"][0][k]) <= 1e-6 * max(abs(v), 1e-3), k
    for n, a, b in zip(model._g_names + model._d_names, res["1"][1], res["0"][1]):
        if a is None or b is None:
            assert a is None and b is None, n
            continue
        den = float(b.norm())
        assert float((a - b).norm()) <= 2e-6 * max(den, 1e-12) + 1e-12, (n, float((a - b).norm()), den)
