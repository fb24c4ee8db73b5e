"""The fp16 two-piece contractions (cape_amd/csrc/gemm_h2.h, pieces.hip) through the C-ABI, against float64 numpy:

* gemm_h2_kernel in every tile form (128 x 128, 64 x 64, DUAL 128 x 64; the library must report plan family 3): ragged row / column
  counts, several sources, forward and data-gradient plane layouts, the de-interleaving epilogue, rank-1 terms, rows spanning
  40 binades incl. a zero row, a 1e-30 row and a 6e4 row -- every row judged on its OWN norm (the scales are per row);
* the row bounds every producer writes (GEMM epilogues, sparse kernels, backward-prep, standalone pass): they must bound the
  true maxima and stay within the documented slack;
* the weight piece planes against their numpy restatement (tests/test_h2_numerics.py);
* dw_h2_kernel (plan family 4) on ragged shapes.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.test_h2_numerics import scale_of, split2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _act(a):
    from cape_amd import ops
    t = ops.alloc_act(a.shape[0], a.shape[1], a.shape[2], torch.device(DEV))
    t.copy_(torch.tensor(a, dtype=torch.float32))
    return t


def _row_scales(N, M, rng):
    s = 2.0 ** (-rng.integers(0, 24, (N, M))).astype(np.float64)
    s[0, min(3, M - 1)] = 0.0
    s[0, min(5, M - 1)] = 1e-30
    s[-1, min(7, M - 1)] = 6e4
    return s


def _per_row_err(got, want):
    """largest relative L2 error of a row judged on that row's own norm (rows that are exactly zero must be exactly zero)."""
    n = np.sqrt((want ** 2).sum(-1))
    e = np.sqrt(((got - want) ** 2).sum(-1))
    assert np.all(e[n == 0] == 0)
    return float((e[n > 0] / n[n > 0]).max())


FWD_CASES = [  # N, Mo, Ch, K, F, affine(DUAL), act, bias
    (2, 203, 128, 2, 72, False, 'leaky', True),       # 64 x 64 tiles, ragged everything
    (2, 203, 64, 2, 72, False, 'leaky', True),        # 128 contraction indices into 72 columns: short -> six-product kernel
    (16, 1000, 256, 2, 512, False, None, False),       # 128 x 128 tiles (16*8*4 = 512), 16 chunks
    (16, 300, 256, 1, 128, False, 'relu', True),       # K total 256: 64 x 64 tiles
    (3, 257, 128, 2, 96, True, None, False),           # DUAL 128 x 64
    (16, 862, 512, 2, 256, True, None, False),         # DUAL at the model's widest affine block
    (1, 37, 32, 3, 64, False, None, False),            # short: 3 chunks of 32 -> stays on the six-product kernel (family 2)
]


@pytest.mark.parametrize("case", FWD_CASES, ids=lambda c: "N%d_Mo%d_C%dx%d_F%d%s" % (c[0], c[1], c[2], c[3], c[4], "_dual" if c[5] else ""))
def test_gemm_h2_forward_planes(case):
    from cape_amd import _lib, ops
    N, Mo, Ch, K, F, dual, act, with_bias = case
    rng = np.random.default_rng(Mo + F)
    W = (rng.standard_normal((Ch * K, F)) * 0.3 * 2.0 ** (-rng.integers(0, 12, (1, F)))).astype(np.float32)      # columns of different size
    Wa = (rng.standard_normal((Ch, F)) * 0.2).astype(np.float32) if dual else None
    hW = torch.tensor(W, device=DEV)
    hWa = torch.tensor(Wa, device=DEV) if dual else None
    P = ops.pieces_for(hW, Ch, K, pair=(hWa, 1) if dual else None)
    Pa = ops.pieces_for(hWa, Ch, 1, pair=(hW, K)) if dual else None
    scales = _row_scales(N, Mo, rng)
    xs = [(rng.standard_normal((N, Mo, Ch)) * scales[:, :, None]).astype(np.float32) for _ in range(K)]
    hx = [_act(x) for x in xs]
    acc = sum(xs[k].astype(np.float64) @ W[k::K].astype(np.float64) for k in range(K))          # rows c*K + k
    entries = []
    for k in range(K):
        e = dict(x=hx[k], csr=None, w=(hW, k * F, K * F, 1), p=P.fwd(k), rm=ops.rowmax(hx[k]))
        if dual and k == 0:
            e["w2"], e["p2"] = (hWa, 0, F, 1), Pa.fwd(0)
        entries.append(e)
    bias = None
    if dual:
        want = np.maximum(acc, 0) + xs[0].astype(np.float64) @ Wa.astype(np.float64)
    else:
        want = acc
        if with_bias:
            b = (rng.standard_normal(F) * 1e-3).astype(np.float32)
            bias = torch.tensor(b, device=DEV)
            want = want + b
        want = np.where(want > 0, want, 0.2 * want) if act == 'leaky' else np.maximum(want, 0) if act == 'relu' else want
    y = ops.alloc_act(N, Mo, F, torch.device(DEV))
    rm_y = ops.alloc_rm(y)
    mask = torch.empty((N, Mo, (F + 31) // 32), device=DEV, dtype=torch.int32) if dual else None
    ops.PLAN_LOG = set()
    try:
        ops.gconv_fwd(entries, y, bias=bias, bias_mode=_lib.BIAS_CHANNEL, act=act or "none", mask=mask, wsi=ops._ptr(P.fsi),
                      wsi2=ops._ptr(Pa.fsi) if dual else None, rm_out=rm_y)
        plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    # (csrc/gemm_h2.h h2_eligible: short contractions are latency-bound launches and stay on the six-product kernel)
    h2 = Ch * K >= 256 or (Ch * K >= 128 and F >= 128)
    fam = {p[1] for p in plans if p[0] == "fwd"}
    assert fam == ({3} if h2 else {2}), plans
    got = y.cpu().numpy().astype(np.float64)
    if with_bias and not dual:
        # the bias is not scaled with the row: judge those rows on the tensor's scale like the other kernels' tests
        err = np.sqrt(((got - want) ** 2).sum(-1)).max() / np.sqrt((want ** 2).sum(-1)).max()
    else:
        err = _per_row_err(got, want)
    assert err < 2e-6, err
    # the epilogue's row bounds: >= the true maximum of every 32-column block, <= the maximum over the block's four-row group
    rm = rm_y.cpu().numpy()
    nb = (F + 31) // 32
    a = np.abs(y.cpu().numpy())
    for j in range(nb):
        blk = a[:, :, 32 * j:32 * j + 32].max(-1)
        assert np.all(rm[:, :, j] >= blk)
        Mp = (Mo + 3) // 4 * 4
        grp = np.zeros((N, Mp))
        grp[:, :Mo] = blk
        grp = np.repeat(grp.reshape(N, Mp // 4, 4).max(-1), 4, axis=1)[:, :Mo]
        assert np.all(rm[:, :, j] <= grp)
    assert np.all(rm[:, :, nb:] == 0)


BWD_CASES = [  # N, Mo, Ch, K, Fout, form: 'deint' = one launch for all orders, 'multi' = one source per order (+ affine term)
    (4, 431, 128, 2, 256, 'deint', False),
    (16, 862, 256, 2, 512, 'deint', False),
    (2, 300, 128, 2, 128, 'multi', True),
    (16, 1723, 128, 2, 128, 'multi', True),
]


@pytest.mark.parametrize("case", BWD_CASES, ids=lambda c: "N%d_Mo%d_C%dx%d_F%d_%s" % c[:6])
def test_gemm_h2_backward_planes(case):
    """Data gradient: contraction over the Fout columns; output column c (de-interleaved launch: c*K + k)."""
    from cape_amd import ops
    N, Mo, Ch, K, Fo, form, aff = case
    rng = np.random.default_rng(Mo + Fo + K)
    W = (rng.standard_normal((Ch * K, Fo)) * 0.3 * 2.0 ** (-rng.integers(0, 10, (Ch, 1)).repeat(K, 0))).astype(np.float32)
    Wa = (rng.standard_normal((Ch, Fo)) * 0.2).astype(np.float32) if aff else None
    hW = torch.tensor(W, device=DEV)
    hWa = torch.tensor(Wa, device=DEV) if aff else None
    P = ops.pieces_for(hW, Ch, K, pair=(hWa, 1) if aff else None)
    Pa = ops.pieces_for(hWa, Ch, 1, pair=(hW, K)) if aff else None
    scales = _row_scales(N, Mo, rng)
    ops.PLAN_LOG = set()
    try:
        if form == 'deint':
            dz = (rng.standard_normal((N, Mo, Fo)) * scales[:, :, None]).astype(np.float32)
            hdz = _act(dz)
            ChP = (Ch + 3) // 4 * 4
            G = ops.alloc_act(N, Mo, K * ChP, torch.device(DEV))
            ops.gconv_fwd([dict(x=hdz, csr=None, w=(hW, 0, 1, Fo), p=(P.b_hi.data_ptr(), P.b_lo.data_ptr(), Fo), rm=ops.rowmax(hdz))], G,
                          deinterleave=K, F=K * Ch, wsi=ops._ptr(P.bsi))
            got = G.cpu().numpy().astype(np.float64)
            full = dz.astype(np.float64) @ W.astype(np.float64).T                   # column j = c*K + k
            want = np.zeros_like(got)
            for j in range(K * Ch):
                want[:, :, (j % K) * ChP + j // K] = full[:, :, j]
        else:
            Ts = [(rng.standard_normal((N, Mo, Fo)) * scales[:, :, None]).astype(np.float32) for _ in range(K + (1 if aff else 0))]
            hT = [_act(t) for t in Ts]
            ent = [dict(x=hT[k], csr=None, w=(hW, k * Fo, 1, K * Fo), p=P.bwd(k), rm=ops.rowmax(hT[k])) for k in range(K)]
            want = sum(Ts[k].astype(np.float64) @ W[k::K].astype(np.float64).T for k in range(K))
            if aff:
                ent.append(dict(x=hT[K], csr=None, w=(hWa, 0, 1, Fo), p=Pa.bwd(0), rm=ops.rowmax(hT[K])))
                want = want + Ts[K].astype(np.float64) @ Wa.astype(np.float64).T
                assert torch.equal(P.bsc, Pa.bsc)                                # the paired tensors share their channel scales
            dx = ops.alloc_act(N, Mo, Ch, torch.device(DEV))
            ops.gconv_fwd(ent, dx, wsi=ops._ptr(P.bsc))
            got = dx.cpu().numpy().astype(np.float64)
        plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    assert {p[1] for p in plans if p[0] == "fwd"} == {3}, plans
    assert _per_row_err(got, want) < 2e-6


@pytest.mark.parametrize("shape", [(64, 2, 72, None), (512, 2, 256, 1), (96, 3, 40, None)], ids=lambda s: "C%dx%d_F%d" % s[:3])
def test_weight_pieces_match_the_numpy_restatement(shape):
    from cape_amd import ops
    Ch, K, F, pairK = shape
    rng = np.random.default_rng(Ch + F)
    W = (rng.standard_normal((Ch * K + 5 * K, F)) * 2.0 ** (-rng.integers(0, 16, (1, F)))).astype(np.float32)      # 5 further (condition) channels
    W[:, 3] = 0.0                                                                   # an all-zero column
    Wp = (rng.standard_normal((Ch * pairK, F)) * 3.0).astype(np.float32) if pairK else None
    hW = torch.tensor(W, device=DEV)
    hWp = torch.tensor(Wp, device=DEV) if pairK else None
    P = ops.pieces_for(hW, Ch, K, pair=(hWp, pairK) if pairK else None)
    Wf = W[:Ch * K]
    # forward: column scales from the column maxima over the feature rows; planes [K][F][Ch]
    s, inv = scale_of(np.abs(Wf).max(0))
    assert np.array_equal(P.fsi.cpu().numpy().reshape(K, F), np.tile(inv, (K, 1)))
    if Ch % 32 == 0:
        hi, lo = split2(Wf * s[None, :])
        fh = P.f_hi.cpu().numpy().view(np.float16).reshape(K, F, Ch).astype(np.float32)
        fl = P.f_lo.cpu().numpy().view(np.float16).reshape(K, F, Ch).astype(np.float32)
        for k in range(K):
            assert np.array_equal(fh[k], hi[k::K].T) and np.array_equal(fl[k], lo[k::K].T)
    # backward: one scale per channel from its K rows (and the partner's), planes in W's own order
    gm = np.abs(Wf).reshape(Ch, K * F).max(1)
    if pairK:
        gm = np.maximum(gm, np.abs(Wp).reshape(Ch, pairK * F).max(1))
    sb, ib = scale_of(gm)
    assert np.array_equal(P.bsc.cpu().numpy(), ib) and np.array_equal(P.bsi.cpu().numpy().reshape(Ch, K), np.repeat(ib[:, None], K, 1))
    hi, lo = split2(Wf * np.repeat(sb, K)[:, None])
    assert np.array_equal(P.b_hi.cpu().numpy().view(np.float16).reshape(Ch * K, F).astype(np.float32), hi)
    assert np.array_equal(P.b_lo.cpu().numpy().view(np.float16).reshape(Ch * K, F).astype(np.float32), lo)


def test_row_bounds_of_the_row_owning_producers(mesh_ops):
    """spmm / spmm_multi / spmm_combine / bwd_prep write [bound, 0, 0, 0] per row in the same launch; the standalone pass likewise."""
    import scipy.sparse as sp
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    rng = np.random.default_rng(5)
    N, C_ = 3, 64
    L = sp.csr_matrix(mesh_ops["L"][6], dtype=np.float64)                           # 862 vertices
    M = L.shape[0]
    S = ops.DeviceCSR(HostCSR(L), torch.device(DEV))
    x = (rng.standard_normal((N, M, C_)) * _row_scales(N, M, rng)[:, :, None]).astype(np.float32)
    hx = _act(x)

    def check(t, exact=True):
        rm = ops.rm_of(t)
        assert rm is not None and rm.shape == (N, t.shape[1], 4)
        r = rm.cpu().numpy()
        true = np.abs(t.cpu().numpy()).max(-1)
        assert np.array_equal(r[:, :, 0], true) if exact else np.all(r[:, :, 0] >= true)
        assert np.all(r[:, :, 1:] == 0)

    check(ops.spmm(hx, S))
    for t in ops.spmm_multi([hx, hx], [S, None]):
        check(t)
    check(ops.spmm_multi([hx, hx], [S, S], sum=True, scales=[1.0, -0.5]))
    y = ops.alloc_act(N, M, C_, torch.device(DEV))
    check(ops.spmm_combine([hx, hx], [S, None], y, act="leaky"))
    ops.drop_rm(hx)
    assert ops.rm_of(hx) is None
    ops.rowmax(hx)
    check(hx)
    # backward-prep: the bound of g bounds dz when g carries one; otherwise the kernel reduces dz itself (256 and 512 channels
    # take the 8-wide form with one column pass, 1024 two passes of the 4-wide one: one entry per pass)
    for F in (64, 256, 1024):
        g = _act((rng.standard_normal((N, 130, F)) * _row_scales(N, 130, rng)[:, :, None]).astype(np.float32))
        yv = _act(rng.standard_normal((N, 130, F)).astype(np.float32))
        dz = ops.bwd_prep(g, y=yv, act="leaky")[0]
        rm = ops.rm_of(dz).cpu().numpy()
        true = np.abs(dz.cpu().numpy()).max(-1)
        assert np.array_equal(rm.max(-1), np.abs(g.cpu().numpy()).max(-1)) and np.all(rm.max(-1) >= true)      # the bound of g bounds dz
        assert ops.rm_of(g) is ops.rm_of(dz)
        ops.drop_rm(g)
        ops.rowmax(g)
        dz2 = ops.bwd_prep(g, y=yv, act="leaky")[0]
        assert ops.rm_of(dz2) is ops.rm_of(g) and np.all(ops.rm_of(dz2).cpu().numpy().max(-1) >= true)


@pytest.mark.parametrize("Cn", [32, 64, 256, 96, 512])
def test_group_norm_bounds_its_rows(Cn):
    """The group-norm apply passes (forward output, backward dx) write the row bounds in the same launch when a row is a
    power-of-two lane group (32 / 64 / 256 channels), through one standalone pass otherwise (96, 512): exact maxima either way."""
    from cape_amd import ops
    rng = np.random.default_rng(Cn)
    N, V = 3, 131
    x = _act((rng.standard_normal((N, V, Cn)) * _row_scales(N, V, rng)[:, :, None]).astype(np.float32)).requires_grad_(True)
    gamma = torch.tensor(rng.standard_normal(Cn).astype(np.float32), device=DEV, requires_grad=True)
    beta = torch.tensor(rng.standard_normal(Cn).astype(np.float32), device=DEV, requires_grad=True)
    for relu in (0, 1):
        y = ops.GroupNormFn.apply(x, gamma, beta, ops.group_count(N, Cn), 1e-5, relu)
        rm = ops.rm_of(y)
        assert rm is not None and rm.shape == (N, V, 4)
        r = rm.cpu().numpy()
        assert np.array_equal(r[:, :, 0], np.abs(y.detach().cpu().numpy()).max(-1)) and np.all(r[:, :, 1:] == 0)
        g = _act((rng.standard_normal((N, V, Cn)) * _row_scales(N, V, rng)[:, :, None]).astype(np.float32))
        seen = {}
        h = x.register_hook(lambda t: seen.setdefault("dx", t))
        y.backward(g)
        h.remove()
        dx = seen["dx"]
        rd = ops.rm_of(dx)
        assert rd is not None, "the gradient tensor handed to the next backward carries its bounds"
        assert np.array_equal(rd.cpu().numpy()[:, :, 0], np.abs(dx.cpu().numpy()).max(-1))
        x.grad = None


DW_CASES = [(2, 203, [64], 72), (16, 862, [256, 256], 512), (5, 330, [128, 64], 132), (16, 1723, [128, 128], 128)]


@pytest.mark.parametrize("case", DW_CASES, ids=lambda c: "N%d_Mo%d_C%s_F%d" % (c[0], c[1], "+".join(map(str, c[2])), c[3]))
def test_dw_h2(case):
    from cape_amd import ops
    N, Mo, Cs, F = case
    rng = np.random.default_rng(Mo + F)
    # rows of very different size inside every workgroup's range: the slab-uniform scales must cope
    sc = 2.0 ** (-rng.integers(0, 12, (N, Mo, 1))).astype(np.float64)
    dz = (rng.standard_normal((N, Mo, F)) * sc).astype(np.float32)
    hdz = _act(dz)
    ops.rowmax(hdz)
    ent, want = [], []
    for C_ in Cs:
        x = (rng.standard_normal((N, Mo, C_)) * sc[::-1]).astype(np.float32)
        hx = _act(x)
        ops.rowmax(hx)
        dW = torch.zeros((C_, F), device=DEV)
        ent.append(dict(x=hx, csr=None, w=(dW, 0, F, 1)))
        want.append(np.einsum('nrc,nrf->cf', x.astype(np.float64), dz.astype(np.float64)))
    ops.PLAN_LOG = set()
    try:
        ops.gconv_dw(ent, hdz)
        plans = set(ops.PLAN_LOG)
    finally:
        ops.PLAN_LOG = None
    assert {p[1] for p in plans if p[0] == "dw"} == {4}, plans
    for e, w in zip(ent, want):
        got = e["w"][0].cpu().numpy().astype(np.float64)
        assert np.abs(got - w).max() / np.abs(w).max() < 2e-6


def test_fused_activation_gradient_chain_equals_op_by_op(mesh_ops):
    """sole_consumer_chain: the layer above differentiates the bias + leaky-ReLU epilogue of the layer below inside the summed
    operator application that produces its incoming gradient (cape_spmm_multi_actgrad).  Same gradients as the op-by-op form for
    every input, weight and bias; the two lower layers run no backward-prep launch (the top layer's own epilogue gradient is
    taken by its backward-prep as before)."""
    from cape_amd import ops
    from cape_amd.graph import ConvOperators
    dev = torch.device(DEV)
    L, D = mesh_ops["L"], mesh_ops["D"]
    rng = np.random.default_rng(11)
    N = 3
    # 1723 vertices: conv (64 -> 64), conv + row-selection pool to 862 (64 -> 128), conv on 862 (128 -> 128)
    specs = [(L[4], None, 64, 64), (L[5], D[5], 64, 128), (L[6], None, 128, 128)]
    dops = [ops.DeviceConvOps(ConvOperators(Lm, 2, pool=Dm), dev) for Lm, Dm, _, _ in specs]
    x0 = torch.tensor(rng.standard_normal((N, 1723, 64)), dtype=torch.float32, device=dev)
    Ws = [torch.tensor(rng.standard_normal((2 * ci, co)) * 0.1, dtype=torch.float32, device=dev) for _, _, ci, co in specs]
    bs = [torch.tensor(rng.standard_normal((1, 1, co)) * 0.1, dtype=torch.float32, device=dev) for _, _, _, co in specs]
    gy = torch.tensor(rng.standard_normal((N, 862, 128)), dtype=torch.float32, device=dev)

    def run(chain):
        x = x0.clone().requires_grad_(True)
        W = [w.clone().requires_grad_(True) for w in Ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        gB = [torch.zeros_like(t) for t in bs]                  # the bias-gradient "bucket" views the fused form writes into
        ops.LAUNCH_LOG = []
        try:
            with ops.sole_consumer_chain(chain):
                h = x
                for k in range(3):
                    h = ops.chebyshev5(h, W[k], dops[k], bias=b[k], activation="b1leakyrelu", bias_grad_buf=gB[k])
            grads = torch.autograd.grad(h, [x] + W + b, grad_outputs=gy)
            names = [e[0] for e in ops.LAUNCH_LOG]
        finally:
            ops.LAUNCH_LOG = None
        torch.cuda.synchronize()
        return [g.cpu().numpy().astype(np.float64) for g in grads], names

    ref, names_ref = run(False)
    got, names = run(True)
    assert names_ref.count("bwd_prep") == 3 and names.count("bwd_prep") == 1, (names_ref.count("bwd_prep"), names.count("bwd_prep"))
    for a, b_ in zip(got, ref):
        assert np.abs(a - b_).max() <= 2e-6 * np.abs(b_).max()


@pytest.mark.parametrize("tamper", ["second_gradient_summed_in", "tag_dropped"])
def test_fused_activation_gradient_hand_over_is_guarded(tamper, mesh_ops):
    """The pre-activated gradient travels between two layers as a Python attribute on an autograd tensor.  If the tagged tensor
    is written afterwards (the engine sums a second consumer's gradient into it) or arrives without its tag (a hook returned a
    copy), the result would be silently wrong -- act' applied to a mixed sum, or applied twice with the bias partials queued
    twice.  Both must raise."""
    from cape_amd import ops
    from cape_amd.graph import ConvOperators
    dev = torch.device(DEV)
    L = mesh_ops["L"]
    rng = np.random.default_rng(12)
    dops = [ops.DeviceConvOps(ConvOperators(L[4], 2), dev) for _ in range(2)]
    x = torch.tensor(rng.standard_normal((2, 1723, 64)), dtype=torch.float32, device=dev, requires_grad=True)
    W = [torch.tensor(rng.standard_normal((128, 64)) * 0.1, dtype=torch.float32, device=dev, requires_grad=True) for _ in range(2)]
    b = [torch.tensor(rng.standard_normal((1, 1, 64)) * 0.1, dtype=torch.float32, device=dev, requires_grad=True) for _ in range(2)]
    gB = [torch.zeros_like(t) for t in b]
    with ops.sole_consumer_chain(True):
        h1 = ops.chebyshev5(x, W[0], dops[0], bias=b[0], activation="b1leakyrelu", bias_grad_buf=gB[0])
        if tamper == "second_gradient_summed_in":
            h1.register_hook(lambda g: g.add_(1.0))            # in place: the tag stays on the object, the version moves
        else:
            h1.register_hook(lambda g: g.clone())              # a copy: the contents are pre-activated, the tag is gone
        h2 = ops.chebyshev5(h1, W[1], dops[1], bias=b[1], activation="b1leakyrelu", bias_grad_buf=gB[1])
    with pytest.raises(RuntimeError, match="sole_consumer_chain"):
        torch.autograd.grad(h2.sum(), [x] + W + b)
    torch.cuda.synchronize()
