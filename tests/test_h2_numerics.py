"""Host-side (numpy) restatement of the fp16 two-piece operand split of cape_amd/csrc/gemm_h2.h, pinning the numerical claims
DESIGN.md section 4 makes for the three-product contractions:
  1. the power-of-two scale taken from a row bound m puts m into [2^13, 2^14) and its reciprocal undoes it exactly; zero,
     denormal and huge bounds are clamped without overflow (h2_scale_of);
  2. x*s = hi + lo with both pieces round-to-nearest fp16 represents x to 2^-22 relative for elements down to 2^-16 of the
     row bound and to 2^-38 of the BOUND below that (graceful loss, never worse);
  3. a contraction from the three products lo*hi + hi*lo + hi*hi (accumulated in fp32) is as accurate as an fp32 FMA chain on
     rows spanning many binades -- including a zero row, a 1e-30 row, a row at 6e4 times the others, and a bound that
     over-estimates the true maximum by a few binades (the MFMA epilogues bound groups of four rows);
  4. inf operands give inf / NaN (as the six-product kernels do), they do not corrupt other rows.
The device kernels are compared with float64 by tests/test_gpu_h2.py."""
import numpy as np


def scale_of(m):
    """h2_scale_of: (s, inv) from the biased exponent of the bound, clamped to [14, 253]."""
    m = np.asarray(m, dtype=np.float32)
    e = ((m.view(np.uint32) >> np.uint32(23)) & np.uint32(255)).astype(np.int64)
    e = np.clip(e, 14, 253)
    s = ((267 - e).astype(np.uint32) << np.uint32(23)).view(np.float32)
    inv = ((e - 13).astype(np.uint32) << np.uint32(23)).view(np.float32)
    return s, inv


def split2(xs):
    """hi = fp16(x), lo = fp16(x - hi), both round-to-nearest-even (numpy's float16 conversion), as fp32 values."""
    xs = np.asarray(xs, dtype=np.float32)
    with np.errstate(over='ignore', invalid='ignore'):
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def h2_matmul(A, B, row_bound, col_bound):
    """The kernel's arithmetic: three products per multiply-add, fp32 accumulation (products of two fp16 numbers are exact
    in fp32; numpy accumulates each dot product in fp32 pairwise -- an fp32 accumulator like the MFMA's)."""
    sa, ia = scale_of(row_bound)
    sb, ib = scale_of(col_bound)
    ah, al = split2(A * sa[:, None])
    bh, bl = split2(B * sb[None, :])
    with np.errstate(over='ignore', invalid='ignore'):
        acc = (al @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (ah @ bh).astype(np.float32)
        return acc * (ia[:, None] * ib[None, :])


def test_scale_of_puts_bounds_into_range_and_is_exactly_invertible():
    rng = np.random.default_rng(0)
    m = np.concatenate([np.exp(rng.uniform(-80, 80, 20000)).astype(np.float32),
                        np.array([0.0, 1e-45, 1e-38, 2.0 ** -113, 1.0, 6e4, 3.0e38, np.finfo(np.float32).max], dtype=np.float32)])
    s, inv = scale_of(m)
    assert np.all(s * inv == 1.0) and np.all(np.isfinite(s)) and np.all(np.isfinite(inv))
    normal = (m >= 2.0 ** -112) & (m < 2.0 ** 127)
    scaled = m[normal].astype(np.float64) * s[normal]
    assert np.all(scaled >= 2.0 ** 13) and np.all(scaled < 2.0 ** 14)
    # clamped ends: tiny bounds are scaled by at most 2^126 * ... and stay far below the fp16 maximum; huge ones stay finite
    assert np.all(m.astype(np.float64) * s < 2.0 ** 15)


def test_two_piece_representation_error():
    rng = np.random.default_rng(1)
    # elements from the row bound (2^14 after scaling) down to 2^-30 of it
    x = (rng.uniform(1.0, 2.0, 200000) * 2.0 ** rng.integers(-16, 14, 200000)).astype(np.float32) * rng.choice([-1.0, 1.0], 200000).astype(np.float32)
    hi, lo = split2(x)
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64))
    big = np.abs(x) >= 2.0 ** -3                           # 2^-16 of a bound scaled to 2^13: above the subnormal floor 2^-22 |x| >= 2^-25
    assert np.all(err[big] <= np.abs(x[big]) * 2.0 ** -22)
    assert np.all(err <= 2.0 ** -25 + np.abs(x) * 2.0 ** -22)          # below: half the fp16 subnormal spacing 2^-24 = 2^-39 of the bound


def _case(rng, M, K, F, row_scales):
    A = (rng.standard_normal((M, K)) * row_scales[:, None]).astype(np.float32)
    B = (rng.standard_normal((K, F)) * 0.05).astype(np.float32)
    return A, B


def test_contraction_matches_an_fp32_chain_over_many_binades():
    rng = np.random.default_rng(2)
    M, K, F = 96, 1024, 48
    scales = 2.0 ** (-(np.arange(M) * 7 % 23)).astype(np.float64)
    scales[3] = 0.0                                          # a zero row
    scales[5] = 1e-30                                        # far below everything else
    scales[7] = 6e4                                          # far above
    A, B = _case(rng, M, K, F, scales)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    got = h2_matmul(A, B, np.abs(A).max(1), np.abs(B).max(0))
    chain = np.zeros((M, F), dtype=np.float32)
    for k in range(K):                                       # a sequential fp32 FMA-free chain (mul + add rounding each)
        chain += A[:, k:k + 1] * B[k:k + 1, :]
    live = scales > 0
    rms = lambda e: np.sqrt((e[live] ** 2).sum(1) / (ref[live] ** 2).sum(1))      # per row: every row judged on its own scale
    e_h2, e_chain = rms(got - ref), rms(chain - ref)
    assert np.all(got[~live] == 0.0)
    assert e_h2.max() < 1.2e-6 and np.median(e_h2) < 5e-7
    assert np.median(e_h2) <= 1.5 * np.median(e_chain)       # the class of an fp32 chain (measured: ~0.7x)


def test_over_estimated_bounds_lose_nothing_until_many_binades():
    """The MFMA epilogues bound GROUPS of four rows, the backward-prep bound of dz is that of g: a bound up to 2^6 above the
    true row maximum must not change the accuracy class; at 2^12 the loss is still graceful."""
    rng = np.random.default_rng(3)
    A, B = _case(rng, 64, 512, 32, np.ones(64))
    ref = A.astype(np.float64) @ B.astype(np.float64)
    true = np.abs(A).max(1)
    base = None
    for slack, tol in ((1.0, None), (2.0 ** 3, 1.3), (2.0 ** 6, 1.6), (2.0 ** 12, 60.0)):
        got = h2_matmul(A, B, true * slack, np.abs(B).max(0))
        e = np.sqrt(((got - ref) ** 2).sum() / (ref ** 2).sum())
        if base is None:
            base = e
            assert e < 5e-7
        else:
            assert e < tol * base, (slack, e, base)


def test_inf_and_nan_stay_in_their_rows():
    rng = np.random.default_rng(4)
    A, B = _case(rng, 8, 64, 16, np.ones(8))
    A[2, 5] = np.inf
    A[6, 0] = np.nan
    bound = np.abs(A).max(1)                                 # inf / nan for rows 2 / 6
    got = h2_matmul(A, B, bound, np.abs(B).max(0))
    ref = A.astype(np.float64) @ B.astype(np.float64)
    ok = [0, 1, 3, 4, 5, 7]
    assert np.all(~np.isfinite(got[2])) and np.all(np.isnan(got[6]))
    e = np.abs(got[ok] - ref[ok]).max() / np.abs(ref[ok]).max()
    assert e < 2e-6


def test_row_bounds_are_voided_by_in_place_writes():
    """The bounds attached to an activation tensor describe its contents at that moment (cape_amd.ops.set_rm / rm_of): any
    in-place torch write afterwards -- e.g. the autograd engine accumulating a second gradient into the tensor -- must void
    them (a stale, too small bound overflows fp16), views do not inherit them, an explicit drop removes them."""
    import torch
    from cape_amd import ops
    t = torch.zeros(2, 5, 32)
    rm = torch.ones(2, 5, 4)
    assert ops.rm_of(t) is None
    assert ops.set_rm(t, rm) is t and ops.rm_of(t) is rm
    assert ops.rm_of(t[:, :, :16]) is None                    # a view is a new object
    u = t.detach()                                            # shares storage and version counter, but not the attribute
    assert ops.rm_of(u) is None
    t.add_(1.0)                                               # what the engine's in-place accumulation does
    assert ops.rm_of(t) is None
    ops.set_rm(t, rm)
    assert ops.rm_of(t) is rm
    u.mul_(2.0)                                               # a write through an alias bumps the shared counter too
    assert ops.rm_of(t) is None
    ops.set_rm(t, rm)
    ops.drop_rm(t)
    assert ops.rm_of(t) is None
    assert ops.rm_of(ops.set_rm(torch.zeros(2, 6, 32), rm)) is None      # bounds of another row count never apply
