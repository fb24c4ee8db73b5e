"""Ragged / edge-case GPU parity for the C-ABI kernels on small random meshes: channel counts that are
not multiples of 4, single-sample batches, outputs narrower than one MFMA tile, operators with EMPTY rows
and with long rows, vertex counts that are not multiples of any tile size, strided (channel-sliced) views.
Checked against dense float64 numpy."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-5


def rel(a, ref):
    a, r = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    return np.abs(a - r).max() / max(np.abs(r).max(), 1e-30)


def rand_csr(rng, rows, cols, density, empty_rows=0, long_row=None):
    m = sp.random(rows, cols, density=density, random_state=np.random.RandomState(rng.integers(1 << 30)), format="lil")
    for r in rng.choice(rows, size=empty_rows, replace=False) if empty_rows else []:
        m.rows[r], m.data[r] = [], []
    if long_row is not None:
        m[0, :long_row] = rng.standard_normal(long_row)
    return sp.csr_matrix(m, dtype=np.float64)


CASES = [  # N, Mi, Mo, [C per source], F, dual, empty_rows
    (1, 37, 37, [5], 1, False, 0),
    (3, 131, 97, [7, 7], 3, False, 4),
    (2, 200, 260, [12, 9, 4], 33, False, 0),
    (2, 129, 129, [36, 36], 70, True, 3),
    (1, 300, 150, [64], 130, False, 0),
    (4, 65, 65, [3, 3, 3], 40, True, 0),
    (2, 150, 150, [12, 8, 4], 32, False, 2),      # aligned: pipelined kernels, three sources packed into one dW tile
    (3, 170, 90, [36, 36], 72, False, 0),
    (2, 140, 140, [68, 132], 136, False, 0),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "N%d_Mi%d_Mo%d_F%d%s" % (c[0], c[1], c[2], c[4], "_dual" if c[5] else ""))
def test_gather_gemm_ragged(case):
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    N, Mi, Mo, Cs, F, dual, empty = case
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(Mi * 1000 + F)
    ref = np.zeros((N, Mo, F))
    ref2 = np.zeros((N, Mo, F))
    entries, keep = [], []
    for i, C in enumerate(Cs):
        # source 0 is the identity when shapes allow, the others are random operators (some rows empty, one long)
        S = None if (i == 0 and Mi == Mo) else rand_csr(rng, Mo, Mi, 0.05, empty_rows=empty, long_row=min(Mi, 40))
        # activations live inside a wider buffer: a channel-offset, row-padded view
        buf = torch.zeros(N, Mi, C + 9, device=dev)
        x = rng.standard_normal((N, Mi, C))
        buf[:, :, 5:5 + C] = torch.tensor(x, dtype=torch.float32)
        W = rng.standard_normal((C, F)) * 0.3
        hW = torch.tensor(W, dtype=torch.float32, device=dev)
        A = x if S is None else np.stack([S @ x[n] for n in range(N)])
        ref += A @ W
        e = dict(x=buf[:, :, 5:5 + C], csr=None if S is None else ops.DeviceCSR(HostCSR(S), dev), w=(hW, 0, F, 1))
        if dual and i == 0:
            W2 = rng.standard_normal((C, F)) * 0.3
            hW2 = torch.tensor(W2, dtype=torch.float32, device=dev)
            ref2 += A @ W2
            e["w2"] = (hW2, 0, F, 1)
            keep.append(hW2)
        entries.append(e)
        keep += [buf, hW]
    y = ops.alloc_act(N, Mo, F, dev)
    if dual:
        mask = torch.zeros((N, Mo, (F + 31) // 32), device=dev, dtype=torch.int32)
        ops.gconv_fwd(entries, y, mask=mask)
        want = np.maximum(ref, 0) + ref2
        bits = ((mask.cpu().numpy().astype(np.int64)[..., None] >> np.arange(32)) & 1).reshape(N, Mo, -1)[:, :, :F]
        safe = np.abs(ref) > 1e-4 * np.abs(ref).max()          # sign is only meaningful away from 0
        assert np.array_equal(bits[safe] == 1, (ref > 0)[safe])
    else:
        b = rng.standard_normal(F) * 0.1
        hb = torch.tensor(b, dtype=torch.float32, device=dev)
        ops.gconv_fwd(entries, y, bias=hb, bias_mode=1, act="leaky")
        z = ref + b
        want = np.where(z > 0, z, 0.2 * z)
    torch.cuda.synchronize()
    assert rel(y.cpu().numpy(), want) < TOL
    # weight gradients of the same launch geometry
    dz = rng.standard_normal((N, Mo, F))
    hdz = torch.tensor(dz, dtype=torch.float32, device=dev)
    grads = [torch.empty(C, F, device=dev) for C in Cs]
    ops.gconv_dw([dict(x=e["x"], csr=e["csr"], w=(gw, 0, F, 1)) for e, gw in zip(entries, grads)], hdz)
    torch.cuda.synchronize()
    for i, (e, gw) in enumerate(zip(entries, grads)):
        x = e["x"].cpu().numpy().astype(np.float64)
        S = None if e["csr"] is None else e["csr"]
        A = x if S is None else np.stack([sp.csr_matrix((S.vals_t.cpu().numpy().astype(np.float64), S.colidx_t.cpu().numpy(),
                                                         S.rowptr_t.cpu().numpy()), shape=S.shape) @ x[n] for n in range(N)])
        want_w = np.einsum('nrc,nrf->cf', A, dz)
        assert rel(gw.cpu().numpy(), want_w) < TOL, i


def test_spmm_empty_rows_and_axpby():
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    for C in (1, 6, 8, 33):
        S = rand_csr(rng, 77, 50, 0.1, empty_rows=10, long_row=30)
        x, z = rng.standard_normal((2, 50, C)), rng.standard_normal((2, 77, C))
        csr = ops.DeviceCSR(HostCSR(S), dev)
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
        y = ops.spmm(t(x), csr, alpha=2.0, z=t(z), beta=-1.0)
        want = np.stack([2.0 * (S @ x[n]) - z[n] for n in range(2)])
        assert rel(y.cpu().numpy(), want) < TOL


def test_bwd_prep_ragged():
    from cape_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(9)
    for (N, M, F, R) in ((1, 19, 5, 0), (3, 333, 36, 2), (2, 140, 130, 3)):
        g, y = rng.standard_normal((N, M, F)), rng.standard_normal((N, M, F))
        rs = rng.standard_normal((max(R, 1) + 1, M))
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
        dz, db, dc, dcg = ops.bwd_prep(t(g), y=t(y), act="leaky", want_bias=True, rowscale=t(rs).contiguous(), R=R,
                                       rg=R if R else None)
        want = g * np.where(y > 0, 1.0, 0.2)
        assert rel(dz.cpu().numpy(), want) < TOL and rel(db.cpu().numpy(), want.sum((0, 1))) < TOL
        if R:
            assert rel(dc.cpu().numpy(), np.einsum('jr,nrf->njf', rs[:R], want)) < TOL
            assert rel(dcg.cpu().numpy(), np.einsum('r,nrf->nf', rs[R], g)) < TOL


def test_cond_coef_all_layers():
    """cape_cond_coef_fwd / _bwd (all consumers of one condition vector in one launch) against numpy."""
    from cape_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(21)
    N, Cc = 5, 24
    cond = rng.standard_normal((N, Cc))
    spec = [(40, 2, 64, True), (8, 3, 300, False), (16, 1, 36, True), (12, 2, 3, False)]     # Ch, K, F, affine
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    layers, ref = [], []
    for Ch, K, F, aff in spec:
        W = rng.standard_normal(((Ch + Cc) * K, F))
        Wa = rng.standard_normal((Ch + Cc, F)) if aff else None
        hW, hWa = t(W), (t(Wa) if aff else None)
        layers.append(dict(W=hW, Wa=hWa, Ch=Ch, K=K, gW=torch.full_like(hW, 7.0), gWa=torch.full_like(hWa, 7.0) if aff else None))
        Wc = W[Ch * K:].reshape(Cc, K, F)
        c = np.einsum('nc,ckf->nkf', cond, Wc)
        if aff:
            c = np.concatenate([c, (cond @ Wa[Ch:])[:, None]], 1)
        ref.append((Wc, Wa[Ch:] if aff else None, c))
    hc = t(cond).requires_grad_(True)
    coefs = ops.CondCoefFn.apply(hc, layers)
    for got, (_, _, c) in zip(coefs, ref):
        assert rel(got.detach().cpu().numpy(), c) < TOL
    dco = [rng.standard_normal(c.shape) for _, _, c in ref]
    torch.autograd.backward(list(coefs), [t(d) for d in dco])
    want_dc = np.zeros((N, Cc))
    for ly, (Wc, Wac, c), d, (Ch, K, F, aff) in zip(layers, ref, dco, spec):
        want_dc += np.einsum('nkf,ckf->nc', d[:, :K], Wc)
        gw = ly["gW"].cpu().numpy()
        assert np.all(gw[:Ch * K] == 7.0)                      # feature rows untouched
        assert rel(gw[Ch * K:].reshape(Cc, K, F), np.einsum('nc,nkf->ckf', cond, d[:, :K])) < TOL
        if aff:
            want_dc += d[:, K] @ Wac.T
            ga = ly["gWa"].cpu().numpy()
            assert np.all(ga[:Ch] == 7.0)
            assert rel(ga[Ch:], cond.T @ d[:, K]) < TOL
    assert rel(hc.grad.cpu().numpy(), want_dc) < TOL


@pytest.mark.parametrize("N,kin,out", [(5, 9001, 18), (16, 8256, 64), (11, 8200, 20), (11, 8256, 64), (16, 55168, 64), (7, 4100, 128)])
def test_fc_long_pair(N, kin, out):
    """Encoder fc_mean / fc_var kernels (csrc/fc.hip) against float64 numpy, ragged sizes."""
    from cape_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + out)
    x = rng.standard_normal((N, kin))
    W = [rng.standard_normal((kin, out)) * 0.05 for _ in range(2)]
    b = [rng.standard_normal(out) for _ in range(2)]
    gy = [rng.standard_normal((N, out)) for _ in range(2)]
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=True)
    hx, hW, hb = t(x), [t(w) for w in W], [t(v) for v in b]
    y0, y1 = ops.dense_pair(hx, hW[0], hb[0], hW[1], hb[1])
    assert rel(y0.detach().cpu().numpy(), x @ W[0] + b[0]) < TOL and rel(y1.detach().cpu().numpy(), x @ W[1] + b[1]) < TOL
    torch.autograd.backward([y0, y1], [torch.tensor(g, dtype=torch.float32, device=dev) for g in gy])
    assert rel(hx.grad.cpu().numpy(), gy[0] @ W[0].T + gy[1] @ W[1].T) < TOL
    for m in range(2):
        assert rel(hW[m].grad.cpu().numpy(), x.T @ gy[m]) < TOL
        assert rel(hb[m].grad.cpu().numpy(), gy[m].sum(0)) < TOL
    # single-matrix entry point
    hx2 = t(x)
    y = ops.dense(hx2, hW[0].detach(), hb[0].detach())
    assert rel(y.detach().cpu().numpy(), x @ W[0] + b[0]) < TOL


@pytest.mark.parametrize("N,kin,out,act", [(3, 50, 9001, "leaky_relu"), (16, 128, 8256, None), (13, 132, 8448, "leaky_relu"),
                                          (7, 128, 8256, "leaky_relu"), (16, 128, 55168, "leaky_relu"), (16, 64, 8256, None), (9, 256, 8256 + 64, "leaky_relu")])
def test_fc_wide(N, kin, out, act):
    """Decoder fc1 kernels (bias + leaky-ReLU fused) against float64 numpy."""
    from cape_amd import ops
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(N + kin)
    x, W, b = rng.standard_normal((N, kin)), rng.standard_normal((kin, out)) * 0.2, rng.standard_normal(out)
    gy = rng.standard_normal((N, out))
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev, requires_grad=True)
    hx, hW, hb = t(x), t(W), t(b)
    y = ops.dense(hx, hW, hb, activation=act)
    z = x @ W + b
    want = np.where(z > 0, z, 0.2 * z) if act else z
    assert rel(y.detach().cpu().numpy(), want) < TOL
    y.backward(torch.tensor(gy, dtype=torch.float32, device=dev))
    dz = gy * (np.where(z > 0, 1.0, 0.2) if act else 1.0)
    assert rel(hW.grad.cpu().numpy(), x.T @ dz) < TOL
    assert rel(hb.grad.cpu().numpy(), dz.sum(0)) < TOL
    assert rel(hx.grad.cpu().numpy(), dz @ W.T) < 5 * TOL


def test_spmm_multi_separate_and_sum():
    """cape_spmm_multi: several operators in one launch (separate outputs / summed), identity terms, empty rows."""
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    for C in (3, 8, 36):
        N, Mi, Mo = 3, 61, 47
        S = [rand_csr(rng, Mo, Mi, 0.1, empty_rows=5, long_row=20) for _ in range(3)]
        x = rng.standard_normal((N, Mi, C))
        cs = [ops.DeviceCSR(HostCSR(m), dev) for m in S]
        outs = ops.spmm_multi([t(x)] * 3, cs)
        for m, o in zip(S, outs):
            assert rel(o.cpu().numpy(), np.stack([m @ x[n] for n in range(N)])) < TOL
        # sum mode with distinct inputs and an identity term (square operators)
        Sq = [rand_csr(rng, Mi, Mi, 0.1, empty_rows=3) for _ in range(2)]
        xs = [rng.standard_normal((N, Mi, C)) for _ in range(3)]
        y = ops.spmm_multi([t(v) for v in xs], [None] + [ops.DeviceCSR(HostCSR(m), dev) for m in Sq], sum=True)
        want = xs[0] + np.stack([Sq[0] @ xs[1][n] + Sq[1] @ xs[2][n] for n in range(N)])
        assert rel(y.cpu().numpy(), want) < TOL


@pytest.mark.parametrize("dual", [False, True])
def test_spmm_combine_epilogues(dual):
    """cape_spmm_combine: operators applied after the contraction, rank-1 terms, single and DUAL epilogues."""
    from cape_amd import ops
    from cape_amd.graph import HostCSR
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11 + dual)
    N, Mi, Mo, F = 3, 53, 101, 64 if dual else 36
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    S = [rand_csr(rng, Mo, Mi, 0.08, empty_rows=4, long_row=16) for _ in range(3)]
    Z = [rng.standard_normal((N, Mi, F)) for _ in range(3)]
    R = 3
    rowscale, coef = rng.standard_normal((R, Mo)), rng.standard_normal((N, R, F))
    cs = [ops.DeviceCSR(HostCSR(m), dev) for m in S]
    app = lambda m, z: np.stack([m @ z[n] for n in range(N)])
    r1 = lambda j: rowscale[j][None, :, None] * coef[:, j][:, None, :]
    y = ops.alloc_act(N, Mo, F, dev)
    if dual:
        mask = torch.zeros((N, Mo, F // 32), device=dev, dtype=torch.int32)
        ops.spmm_combine([t(z) for z in Z], cs, y, to_acc2=0b100, rank=(t(rowscale).contiguous(), t(coef).contiguous(), 0b100),
                         dual=True, mask=mask)
        a1 = app(S[0], Z[0]) + app(S[1], Z[1]) + r1(0) + r1(1)
        a2 = app(S[2], Z[2]) + r1(2)
        want = np.maximum(a1, 0) + a2
        bits = ((mask.cpu().numpy().astype(np.int64)[..., None] >> np.arange(32)) & 1).reshape(N, Mo, -1)[:, :, :F]
        safe = np.abs(a1) > 1e-4 * np.abs(a1).max()
        assert np.array_equal(bits[safe] == 1, (a1 > 0)[safe])
    else:
        b = rng.standard_normal((Mo, F))
        ops.spmm_combine([t(z) for z in Z], cs, y, rank=(t(rowscale).contiguous(), t(coef).contiguous(), 0), bias=t(b),
                         bias_mode=2, act="leaky")
        z = sum(app(S[k], Z[k]) + r1(k) for k in range(3)) + b
        want = np.where(z > 0, z, 0.2 * z)
    torch.cuda.synchronize()
    assert rel(y.cpu().numpy(), want) < TOL
