"""CPU-only checks: the C-ABI library loads and exports every symbol include/cape_hip.h declares,
host-side operator algebra, loaders, config assembly, and the loud-failure rule (no CPU fallback)."""
import ctypes
import os
import sys
import re

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_match_header():
    hdr = open(os.path.join(ROOT, "include", "cape_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|int32_t|int64_t)\s+(cape_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 18
    lib = ctypes.CDLL(os.path.join(ROOT, "cape_amd", "libcape_hip.so"))
    for name in declared:
        assert hasattr(lib, name), "missing export %s" % name
    from cape_amd import _lib
    assert set(_lib.SIGNATURES) == declared
    assert _lib.lib.cape_abi_version() == int(re.search(r"#define CAPE_ABI_VERSION (\d+)", hdr).group(1))


def test_csr_validate_error_codes():
    from cape_amd._lib import lib
    rp = np.array([0, 1, 3], dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    assert lib.cape_csr_validate(2, 2, 3, p(rp), p(np.array([0, 0, 1], dtype=np.int32))) == 0
    assert lib.cape_csr_validate(2, 2, 3, p(rp), p(np.array([0, 1, 0], dtype=np.int32))) == -2      # unsorted
    assert lib.cape_csr_validate(2, 2, 3, p(rp), p(np.array([0, 0, 5], dtype=np.int32))) == -3      # range
    assert lib.cape_csr_validate(2, 2, 3, None, None) == -1


def test_sparse_entry_points_reject_bad_arguments_before_launching():
    """Argument checks of the sparse entry points run on the host before any launch: NULL operands, leading dimensions
    narrower than the channel count, and per-sample extents that would overflow the kernels' 32-bit work-item index."""
    import ctypes as C
    from cape_amd._lib import lib
    P = C.c_void_p
    x, y, rp, ci, va = P(0x100000), P(0x200000), P(0x300000), P(0x400000), P(0x500000)
    args = lambda **kw: [kw.get("x", x), 64 * 64, kw.get("ldx", 64), rp, kw.get("ci", ci), va, 8, kw.get("ew", 0), 1.0, None, 0, 0, 0.0,
                         kw.get("y", y),
                         64 * 64, kw.get("ldy", 64), kw.get("N", 2), kw.get("Mo", 64), kw.get("C", 64), None, None]
    assert lib.cape_spmm(*args(x=None)) == -1
    assert lib.cape_spmm(*args(y=None)) == -1
    assert lib.cape_spmm(*args(ldx=32)) == -1                      # leading dimension below the channel count
    assert lib.cape_spmm(*args(N=0)) == -1
    big = 1 << 20                                                  # Mo * C = 2^31: beyond the 32-bit item index
    assert lib.cape_spmm(*args(Mo=big, C=2048, ldx=2048, ldy=2048)) == -1
    assert lib.cape_spmm_bf16(*args(Mo=big, C=2048, ldx=2048, ldy=2048)) == -1
    assert lib.cape_spmm(*args(ew=5)) == -1                        # ELL operands: width 4, 8 or 12 ...
    assert lib.cape_spmm(*args(ew=8, ci=P(0x400004))) == -1        # ... and 16-byte aligned arrays


def test_compute_entry_points_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cape_amd import _lib
    with pytest.raises(_lib.CapeHipError):
        _lib.require_gpu()
    from cape_amd.configs import cape_params
    from cape_amd.load_data import load_graph_mtx
    from cape_amd.models import CAPE
    L, D, U, p, Ld, Dd, _ = load_graph_mtx(None, True)
    with pytest.raises((RuntimeError, AssertionError, _lib.CapeHipError)):
        CAPE(L=L, D=D, U=U, L_d=Ld, D_d=Dd, **cape_params(p=p, batch_size=2)).build_graph(6890, 3, 'demo')


def test_laplacian_and_rescale_semantics(mesh_ops):
    from cape_amd import mesh_sampling as ms
    from oracle import cape_oracle as co
    for L in mesh_ops["L"][::2] + mesh_ops["L_d"]:
        assert L.dtype == np.float32 and sp.isspmatrix_csr(L)
        Lt = ms.rescale_L(sp.csr_matrix(L), 2)
        Lo = co.rescale_L(sp.csr_matrix(L), 2)
        assert (Lt != Lo).nnz == 0
        assert abs(Lt - Lt.T).max() == 0 and Lt.diagonal().max() == 0          # symmetric, zero diagonal
        assert Lt.has_sorted_indices


def test_operator_composition(mesh_ops):
    from cape_amd.graph import ConvOperators, is_identity, is_row_selection
    L, D, U = mesh_ops["L"], mesh_ops["D"], mesh_ops["U"]
    assert is_identity(D[0]) and is_row_selection(D[1]) and not is_identity(D[1])
    rng = np.random.default_rng(0)
    # encoder layer with pooling: S_k = D T_k(L~)
    ops = ConvOperators(L[1], 2, pool=D[1])
    x = rng.standard_normal((6890, 5))
    Lt = sp.csr_matrix(L[1], dtype=np.float64) - sp.identity(6890)
    want = [D[1].astype(np.float64) @ x, D[1].astype(np.float64) @ (Lt @ x)]
    for k in range(2):
        got = ops.fwd[k].to_scipy().astype(np.float64) @ x
        assert np.abs(got - want[k]).max() < 1e-6
        assert np.abs(ops.bwd[k].to_scipy().toarray() - ops.fwd[k].to_scipy().toarray().T).max() == 0
    # decoder layer with unpooling: S_k = T_k(L~) U ; identity-level U prunes to an exact identity
    ops = ConvOperators(L[0], 2, unpool=U[0])
    assert ops.fwd[0].identity and ops.Mi == 6890
    ops = ConvOperators(L[1], 3, unpool=U[1])
    assert (ops.Mi, ops.Mo) == (3445, 6890) and len(ops.fwd) == 3 and not ops.fwd[0].identity
    xs = rng.standard_normal((3445, 4))
    Um = U[1].astype(np.float64)
    T2 = 2 * (Lt @ (Lt @ (Um @ xs))) - Um @ xs
    assert np.abs(ops.fwd[2].to_scipy().astype(np.float64) @ xs - T2).max() < 1e-5
    # rank-1 condition terms: S_k 1
    rs = ops.cond_row_terms()
    assert np.abs(rs[1] - np.asarray((Lt @ Um).sum(axis=1)).ravel()).max() < 1e-5
    # K above FUSE_MAX_K falls back to the explicit recurrence
    assert not ConvOperators(L[0], 6).fused


def test_load_graph_mtx_and_configs(mesh_ops):
    from cape_amd.configs import cape_params
    from cape_amd.load_data import filter_cloth_pose, load_graph_mtx
    assert mesh_ops["p"] == [6890, 6890, 3445, 3445, 1723, 1723, 862, 862, 862]
    L3 = load_graph_mtx(None, load_for_demo=False)
    assert len(L3) == 3 and [m.shape[0] for m in L3[0]] == [6890, 3445, 1723, 862, 431]
    rot = mesh_ops["pack"]["demo_rot"]
    assert filter_cloth_pose(rot).shape == (6, 126) and filter_cloth_pose(rot[:, :72]).shape == (6, 42)
    P = cape_params('CAPE-affineconv_nz64_pose32_clotype32_male', p=mesh_ops["p"])
    assert P["F"] == [64, 64, 128, 128, 256, 256, 512, 512] and P["K"] == [2] * 8 and P["affine"] and P["nz"] == 64
    assert cape_params('CAPE_nz18_pose24_clotype8_male')["affine"] is False


def test_vertex_edge_table(mesh_ops):
    from cape_amd.graph import vertex_edge_table
    edges = mesh_ops["pack"]["edges_smpl"]
    ptr, idx = vertex_edge_table(edges, 6890)
    assert ptr[-1] == 2 * len(edges) and len(idx) == 2 * len(edges)
    v = 1234
    inc = idx[ptr[v]:ptr[v + 1]]
    for code in inc:
        assert edges[code >> 1][code & 1] == v


def test_build_entry_point_runs():
    """__graft_entry__.build() (what the driver calls): make is a no-op when the library is current; the ABI
    version the library reports must be the header's."""
    import __graft_entry__ as g
    g.build()


def test_forward_plan_query_is_pure_host_logic():
    """cape_gconv_fwd_plan / cape_gconv_dw_workspace_bytes run without a GPU: kernel family and tile selection."""
    import ctypes as C
    from cape_amd import _lib
    lib = _lib.lib

    def srcs(specs):
        arr = (_lib.CapeSrc * len(specs))()
        for s, (Cn, ldx, gather, w_rs, w_cs, base) in zip(arr, specs):
            s.x, s.x_sample_stride, s.ldx, s.C = base, 6890 * ldx, ldx, Cn
            s.rowptr = s.colidx = s.vals = (0x2000 if gather else None)
            s.w, s.w_rs, s.w_cs = 0x10000, w_rs, w_cs
            s.w2, s.w2_rs, s.w2_cs = None, 0, 0
        return arr

    def plan(specs, N, Mo, F):
        out = (C.c_int32 * 4)()
        assert lib.cape_gconv_fwd_plan(srcs(specs), len(specs), N, Mo, F, out) == 0
        return list(out)

    split = int(os.environ.get("CAPE_GEMM_BF16X6", "1"))     # bf16x6 family (2) replaces the fp32-MFMA family (1) where eligible
    fam = 2 if split else 1
    # plain aligned sources, forward weight layout (rows c*K+k of [C*K, F], F contiguous) -> pipelined kernel, 64x64
    assert plan([(64, 64, False, 2 * 128, 1, 0x4000), (64, 64, False, 2 * 128, 1, 0x8000)], 16, 862, 128) == [fam, 64, 64, 0]
    # data-gradient layout (contraction contiguous)
    assert plan([(128, 128, False, 1, 2 * 128, 0x4000)], 16, 862, 64) == [fam, 64, 64, 1]
    # wide coarse-level layer: 128x128 tiles for the split family once every CU gets two of them
    assert plan([(512, 512, False, 512, 1, 0x4000)] * 2, 16, 862, 512) == ([2, 128, 128, 0] if split else [1, 64, 64, 0])
    assert plan([(512, 512, False, 256, 1, 0x4000)], 16, 862, 256) == [fam, 64, 64, 0]
    # a source that is not a whole number of 32-wide chunks, or a second weight set, stays on the fp32-MFMA family
    assert plan([(48, 48, False, 128, 1, 0x4000)], 16, 862, 128)[0] == 1
    # narrow output -> 128 x 32 tiles
    assert plan([(64, 64, False, 32, 1, 0x4000)], 16, 6890, 32) == [1, 128, 32, 0]
    # gathered source, unaligned base, or unpadded odd channel count -> gather kernel
    assert plan([(64, 64, True, 128, 1, 0x4000)], 16, 862, 128)[0] == 0
    assert plan([(64, 64, False, 128, 1, 0x4004)], 16, 862, 128)[0] == 0
    assert plan([(3, 3, False, 64, 1, 0x4000)], 16, 6890, 64)[0] == 0
    # row-padded 3-channel input: at most 8 input channels in total -> the narrow-input form (csrc/narrow.h), also for two sources;
    # nine channels are one too many
    narrow = int(os.environ.get("CAPE_NARROW", "1"))
    assert plan([(3, 4, False, 64, 1, 0x4000)], 16, 6890, 64)[0] == (4 if narrow else 1)
    assert plan([(3, 4, False, 2 * 64, 1, 0x4000), (3, 4, False, 2 * 64, 1, 0x8000)], 16, 6890, 64)[0] == (4 if narrow else 1)
    assert plan([(3, 4, False, 3 * 64, 1, 0x4000)] * 3, 16, 6890, 64)[0] == 1
    assert lib.cape_gconv_fwd_plan(None, 1, 16, 862, 128, (C.c_int32 * 4)()) == -1
    # weight-gradient workspace: positive, grows with the output size, argument errors reported
    w1 = lib.cape_gconv_dw_workspace_bytes(srcs([(64, 64, False, 64, 1, 0x4000)]), 1, 16, 862, 64)
    w2 = lib.cape_gconv_dw_workspace_bytes(srcs([(512, 512, False, 512, 1, 0x4000)] * 2), 2, 16, 862, 512)
    assert 0 < w1 < w2 and lib.cape_gconv_dw_workspace_bytes(None, 1, 16, 862, 64) == -1


def test_weight_gradient_split_plan_is_one_balanced_round():
    """cape_gconv_dw_plan (pure host logic): the contraction of every weight-gradient launch of the nz64 model at batch 16
    is cut so that (tiles x splits) fits the 512 resident 128x128 workgroups -- ONE round -- and the split count is a whole
    number of groups of 8 (split s runs on XCD s % 8).  The rule it replaced gave 21 splits = 672 workgroups on the widest
    layer."""
    import ctypes as C
    from cape_amd import _lib
    lib = _lib.lib

    def srcs(Cs, Mo):
        arr = (_lib.CapeSrc * len(Cs))()
        for i, (s, Cn) in enumerate(zip(arr, Cs)):
            s.x, s.x_sample_stride, s.ldx, s.C = 0x100000 * (i + 1), Mo * Cn, Cn, Cn
            s.rowptr = s.colidx = s.vals = None
            s.w, s.w_rs, s.w_cs = 0x10000000, 512, 1
            s.w2, s.w2_rs, s.w2_cs = None, 0, 0
        return arr

    def plan(Cs, Mo, F, N=16):
        out = (C.c_int32 * 4)()
        assert lib.cape_gconv_dw_plan(srcs(Cs, Mo), len(Cs), C.c_void_p(0x40000000), Mo * F, F, None, 0, N, Mo, F, out) == 0
        fam, ct, ft, nsplit = list(out)
        ntiles = sum(-(-c // ct) for c in Cs) * -(-F // ft)
        return fam, ct, ft, nsplit, ntiles

    # (sources, vertices, output channels): encoder conv8 / conv7, decoder affine1 / affine2, mid and fine levels
    for Cs, Mo, F in (([512, 512], 862, 512), ([256, 256], 862, 512), ([512, 512, 512], 862, 256), ([256, 256, 256], 862, 256),
                      ([128, 128], 1723, 256), ([128, 128, 128], 1723, 128), ([64, 64], 3445, 128), ([64, 64, 64], 3445, 64)):
        fam, ct, ft, nsplit, ntiles = plan(Cs, Mo, F)
        assert fam in (2, 3), (Cs, Mo, F, fam)
        if fam == 3:
            assert ntiles * nsplit <= 512, (Cs, Mo, F, ct, ft, nsplit, ntiles)          # one round of two workgroups per CU
            assert ntiles * nsplit >= 256, (Cs, Mo, F, ct, ft, nsplit, ntiles)          # and at least one per CU
            assert nsplit % 8 == 0, (Cs, Mo, F, nsplit)
    # the widest layer: 32 tiles x 16 splits (one sample each), not 21 uneven ones
    assert plan([512, 512], 862, 512)[3:] == (16, 32)


def test_bench_exact_fp32_comparison_is_fault_tolerant(monkeypatch):
    """bench.py's optional child run (exact-fp32 MFMA comparison) must never break the JSON line: a failing child gives
    an error record, a good child gives its numbers."""
    import subprocess
    import types
    import bench
    args = types.SimpleNamespace(steps=30, warmup=5, batch=16, config='CAPE-affineconv_nz64_pose32_clotype32_male', gan=False)

    def fail(*a, **k):
        raise subprocess.CalledProcessError(1, a[0])
    monkeypatch.setattr(subprocess, "run", fail)
    r = bench.exact_fp32_run(args)
    assert set(r) == {"error"} and "CalledProcessError" in r["error"]

    seen = {}

    def good(cmd, env=None, **k):
        seen["cmd"], seen["env"] = cmd, env
        line = '{"value": 4000.5, "unit": "meshes/s", "ms_per_step": 3.9995, "steps": 30}'
        return types.SimpleNamespace(stdout=("warning: x\n" + line + "\n").encode())
    monkeypatch.setattr(subprocess, "run", good)
    r = bench.exact_fp32_run(args)
    assert r["value"] == 4000.5 and r["ms_per_step"] == 3.9995 and "CAPE_GEMM_BF16X6=0" in r["note"]
    assert seen["env"]["CAPE_GEMM_BF16X6"] == "0" and "--no-ab" in seen["cmd"] and "--no-roofline" in seen["cmd"]

    monkeypatch.setattr(subprocess, "run", lambda *a, **k: types.SimpleNamespace(stdout=b"no json here\n"))
    assert "error" in bench.exact_fp32_run(args)


def test_ell_form_of_the_operators_matches_csr(mesh_ops):
    """DeviceCSR's ELL arrays (what the streaming sparse kernels read): same entries in CSR order packed to the front, the
    padding slots (column of slot 0, 0.0); operators with more than 12 entries per row have none."""
    import scipy.sparse as sp
    import torch
    from cape_amd import ops
    from cape_amd.graph import HostCSR, ConvOperators
    pack = mesh_ops["pack"] if "pack" in mesh_ops else None
    L = mesh_ops["L"][0] if "L" in mesh_ops else None
    mats = []
    if L is not None:
        mats.append(sp.csr_matrix(L, dtype=np.float64))
    rng = np.random.default_rng(3)
    R = sp.random(50, 40, density=0.1, random_state=5, format="csr", dtype=np.float64)
    R.data = rng.standard_normal(R.nnz)
    mats.append(R)                                       # ragged rows, some empty
    for Mtx in mats:
        Mtx.sort_indices()
        d = ops.DeviceCSR(HostCSR(Mtx), torch.device("cpu"))
        deg = np.diff(Mtx.indptr)
        if deg.max() > 12:
            assert d.ell_w == 0
            continue
        assert d.ell_w == (deg.max() + 3) // 4 * 4 and d.ell_w in (4, 8, 12)
        ec, ev = d.ell_col_t.numpy(), d.ell_val_t.numpy()
        assert ec.shape == (Mtx.shape[0], d.ell_w) and ec.dtype == np.int32 and ev.dtype == np.float32
        for r in range(Mtx.shape[0]):
            a, b = Mtx.indptr[r], Mtx.indptr[r + 1]
            assert np.array_equal(ec[r, :b - a], Mtx.indices[a:b])
            assert np.array_equal(ev[r, :b - a], Mtx.data[a:b].astype(np.float32))
            assert np.all(ev[r, b - a:] == 0) and np.all(ec[r, b - a:] == (Mtx.indices[a] if b > a else 0))
    wide = sp.random(30, 30, density=0.6, random_state=1, format="csr", dtype=np.float64)
    assert ops.DeviceCSR(HostCSR(wide), torch.device("cpu")).ell_w == 0
    # an all-zero group of four stored IN FRONT of a non-zero entry would end the row early in the kernel: CSR is kept
    Z = sp.csr_matrix((np.array([1., 2., 3., 4., 0., 0., 0., 0., 5.]), np.arange(9), np.array([0, 9])), shape=(1, 9))
    assert ops.DeviceCSR(HostCSR(Z), torch.device("cpu")).ell_w == 0
    Z2 = sp.csr_matrix((np.array([1., 2., 3., 4., 0., 0., 7., 0., 5.]), np.arange(9), np.array([0, 9])), shape=(1, 9))
    assert ops.DeviceCSR(HostCSR(Z2), torch.device("cpu")).ell_w == 12          # zeros inside a live group are harmless
    Z3 = sp.csr_matrix((np.array([1., 2., 3., 4., 0., 0., 0., 0.]), np.arange(8), np.array([0, 8])), shape=(1, 8))
    assert ops.DeviceCSR(HostCSR(Z3), torch.device("cpu")).ell_w == 8           # a trailing zero group ends the row correctly


def test_deferred_sweep_flushes_inside_a_callers_except_block(monkeypatch):
    """a backward sweep that SUCCEEDS while the caller is handling another exception (a retry loop) must flush its
    queued reductions; a sweep that itself raises must drop them."""
    from cape_amd import ops
    from cape_amd.models import CAPE
    calls = []
    monkeypatch.setattr(ops, "flush_deferred", lambda: calls.append(list(ops.DEFERRED)))
    m = CAPE.__new__(CAPE)
    try:
        raise RuntimeError("ambient failure of the caller")
    except RuntimeError:
        with m._deferred_sweep():
            ops.DEFERRED.append("queued reduction")
    assert calls == [["queued reduction"]] and ops.DEFERRED is None
    ops.DEFERRED_DW.append("stale")
    with pytest.raises(ValueError):
        with m._deferred_sweep():
            ops.DEFERRED.append("abandoned")
            raise ValueError("the sweep failed")
    assert len(calls) == 1 and ops.DEFERRED is None and ops.DEFERRED_DW == [] and ops.DEFERRED_GN == []


def test_committed_pmc_summary_belongs_to_the_committed_kernel_sources():
    """bench.py attaches `roofline.traffic` only while profiles/pmc_summary.json carries the fingerprint of cape_amd/csrc: a kernel
    edit after the last evidence collection would silently turn the driver's line into `traffic: null`."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    meta = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))["_meta"]
    assert meta["csrc_sha"] == bench._csrc_fingerprint(), "re-collect the PMC passes (tools/collect_profiles.sh) after kernel changes"
    # and the summary covers the kernel the headline line names as dominant
    line = [l for l in open(os.path.join(ROOT, "profiles", "r06_bench.json")).read().splitlines() if l.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    assert roof["traffic"] and roof["traffic"] > roof["alg_bytes_per_launch"] * 0.5
    # ... and carries the normalised MFMA utilisation of that kernel (SQ_VALU_MFMA_BUSY_CYCLES over SIMD-cycles)
    key = next(k for k in json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json"))) if k.replace(" ", "") == roof["kernel"].replace(" ", ""))
    util = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))[key]["mfma_util"]
    assert 0.0 < util < 1.0
