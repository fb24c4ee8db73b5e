"""Host-side (numpy) restatement of the operand split used by cape_amd/csrc/gemm_split.h, pinning the two
numerical claims DESIGN.md section 4 makes for the GEMMs on the bf16 pipe:
  1. x = hi + mid + lo holds EXACTLY for every finite fp32 x, each piece being a bf16 number for |x| >= 2^-110
     (below that the third piece loses at most 2^-133 absolute when it is truncated to 16 bits);
  2. dropping the three smallest of the nine cross products loses 2^-24 (rms; at most 2^-21) relative to |a||b| per
     product, and a contraction computed from six products per multiply-add has the error of an fp32 FMA chain.
The device results themselves are compared with the float64 oracle by the GPU parity tests (tests/test_gpu_*.py)."""
import numpy as np

MASK = np.uint32(0xFFFF0000)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = (x.view(np.uint32) & MASK).view(np.float32)
    r1 = x - hi                                            # exact in fp32
    mid = (r1.view(np.uint32) & MASK).view(np.float32)
    lo = r1 - mid                                          # exact in fp32, <= 8 significant bits
    return hi, mid, lo


def is_bf16(v):
    return np.all((np.asarray(v, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0)


def sample(rng, n):
    # full 24-bit mantissas over the whole normal exponent range, both signs, plus the awkward values
    bits = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[np.isfinite(x)]
    special = np.array([0.0, -0.0, 1.0, -1.0, np.finfo(np.float32).max, -np.finfo(np.float32).max,
                        np.finfo(np.float32).tiny, 1e-40, -1e-45, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 0.1, 3.0e38],
                       dtype=np.float32)
    return np.concatenate([x, special])


def bf16_trunc(v):
    return (np.asarray(v, dtype=np.float32).view(np.uint32) & MASK).view(np.float32)


def test_three_way_split_is_exact_and_bf16():
    rng = np.random.default_rng(0)
    x = sample(rng, 400000)
    with np.errstate(all='raise'):                         # no overflow / invalid anywhere in the split
        hi, mid, lo = split3(x)
    s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(s, x.astype(np.float64))         # the fp32 identity holds for every finite x
    assert is_bf16(hi) and is_bf16(mid)
    # lo is a bf16 number whenever its last bit is at or above the bf16 subnormal spacing 2^-133, i.e. |x| >= 2^-110;
    # below that (|x| < 7.8e-34) the kernel's 16-bit truncation of lo loses at most 2^-133 absolute
    big = np.abs(x) >= 2.0 ** -110
    assert is_bf16(lo[big]) and big.sum() > 0.8 * x.size
    lost = np.abs(lo.astype(np.float64) - bf16_trunc(lo).astype(np.float64))
    assert lost[big].max() == 0.0 and lost.max() <= 2.0 ** -133
    # the pieces do not overlap: each is at most 2^-8 of the previous one
    nz = (hi != 0) & big
    assert np.all(np.abs(mid[nz]) <= np.abs(hi[nz]) * 2.0 ** -7)
    assert np.all(np.abs(lo[nz]) <= np.abs(hi[nz]) * 2.0 ** -15)


def test_six_products_are_an_fp32_multiply():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(200000) * 10.0 ** rng.integers(-6, 6, 200000)).astype(np.float32)
    b = (rng.standard_normal(200000) * 10.0 ** rng.integers(-6, 6, 200000)).astype(np.float32)
    pa = [p.astype(np.float64) for p in split3(a)]
    pb = [p.astype(np.float64) for p in split3(b)]
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = sum(pa[i] * pb[j] for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)))   # the kernel's terms
    nine = sum(pa[i] * pb[j] for i in range(3) for j in range(3))
    assert np.array_equal(nine, exact)                     # all nine products reproduce the fp32 x fp32 product exactly
    rel = np.abs(six - exact) / np.abs(exact)
    # dropped: mid*lo + lo*mid + lo*lo.  The truncation split leaves |mid| < 2^-7 |x| and |lo| < 2^-15 |x|, so a single
    # product can lose up to 2^-21 (8 fp32 roundings, operands just above a power of two); the rms loss is one fp32
    # rounding (2^-24), and in a contraction the signed losses average out (next test)
    assert rel.max() < 2.0 ** -21
    assert np.sqrt(np.mean(rel ** 2)) < 2.0 ** -23.5


def test_contraction_error_matches_fp32():
    """K = 1024 dot products: six-term products accumulated in fp32 (as the MFMA accumulator does, here in
    index order) against float64 -- the error is that of an fp32 FMA chain, as measured on the device
    (profiles/r01_ubench_bf16x6_tiles_accuracy.txt: 4.96e-07 vs 5.74e-07 of rms(ref))."""
    rng = np.random.default_rng(2)
    K, M = 1024, 512
    A = ((rng.random((M, K)) - 0.5) * 2.0 ** rng.integers(0, 4, (M, K))).astype(np.float32)
    B = (0.05 * (rng.random((M, K)) - 0.5) * 2.0 ** rng.integers(0, 4, (M, K))).astype(np.float32)
    ref = np.einsum('mk,mk->m', A.astype(np.float64), B.astype(np.float64))
    pa, pb = split3(A), split3(B)
    acc6 = np.zeros(M, dtype=np.float32)
    acc32 = np.zeros(M, dtype=np.float32)
    for k in range(K):
        for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)):
            acc6 += pa[i][:, k] * pb[j][:, k]              # bf16 x bf16 is exact in fp32; the add rounds
        acc32 = (acc32.astype(np.float64) + A[:, k].astype(np.float64) * B[:, k]).astype(np.float32)   # fma chain
    scale = np.sqrt(np.mean(ref ** 2))
    e6 = np.sqrt(np.mean((acc6 - ref) ** 2)) / scale
    e32 = np.sqrt(np.mean((acc32 - ref) ** 2)) / scale
    assert e32 < 2e-6 and e6 < 4 * e32, (e6, e32)


# ---- round-to-nearest variant (prototype in tools/ubench/gemm_bf16x3.hip, split2v<true>; v_cvt_pk_bf16_f32 rounds to
# nearest even): measured within +-5 % of the truncation split (profiles/r02_ubench_split_variants.txt), not in the library
def bf16_rne(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3_rn(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = x - hi
    mid = bf16_rne(r1)
    r2 = r1 - mid
    return hi, mid, bf16_rne(r2), r2


def test_round_to_nearest_split_is_exact_with_tighter_pieces():
    rng = np.random.default_rng(3)
    x = sample(rng, 400000)
    x = x[np.abs(x) < 3.0e38]                              # rounding hi up must not overflow to inf
    hi, mid, lo, r2 = split3_rn(x)
    big = np.abs(x) >= 2.0 ** -100
    assert np.array_equal(lo[big], r2[big])                # the third piece needs no rounding
    s = hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)
    assert np.array_equal(s[big], x[big].astype(np.float64))
    nz = big & (x != 0)
    assert np.all(np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -8)
    assert np.all(np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -16)
    # dropped products mid*lo + lo*mid + lo*lo: at most 2^-23 |ab| (truncation split: 2^-21)
    a, b = x[nz][:100000], x[nz][100000:200000]
    keep = (np.abs(a) < 1e18) & (np.abs(b) < 1e18) & (np.abs(a) > 1e-18) & (np.abs(b) > 1e-18)
    a, b = a[keep], b[keep]
    pa = [p.astype(np.float64) for p in split3_rn(a)[:3]]
    pb = [p.astype(np.float64) for p in split3_rn(b)[:3]]
    exact = a.astype(np.float64) * b.astype(np.float64)
    six = sum(pa[i] * pb[j] for i, j in ((0, 2), (2, 0), (1, 1), (0, 1), (1, 0), (0, 0)))
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -22.9 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -25


def test_lds_layouts_of_the_split_kernels_are_conflict_free_on_reads():
    """tools/lds_bank_check.py models the gfx950 ds_read_b128 / ds_write_b128 lane groups: the MFMA operand reads of
    gemm_split_kernel / dw_split_kernel (row pitch 80 bytes) and of the prepared swizzled 64-byte layout must be
    conflict-free; an unpadded, unswizzled layout would be 4-way."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lds_bank_check
    rows = dict(lds_bank_check.main())
    assert rows["operand read, pitch 80, k16 step 0"] == 1 and rows["operand read, pitch 80, k16 step 1"] == 1
    assert rows["operand read, pitch 64, no swizzle"] == 4
    assert rows["operand read, pitch 64, seg ^ (row>>2)&3, step 0"] == 1
    assert rows["stage store k-contiguous, pitch 64 swizzled"] == 1
    assert rows["stage store k-contiguous, pitch 80"] <= 2 and rows["stage store transposed, 1 row per lane, pitch 80"] == 1
