"""Graph operators of the CAPE hot path on MI355X: thin torch.autograd wrappers around the
C-ABI kernels of libcape_hip.so.

Names follow the reference's operator plug-points (lib/models.py:58-62: ``chebyshev5``,
``poolwT``, ``b1leakyrelu`` / ``b1relu`` / ``b1tanh`` / ``b2relu``); PyTorch supplies device
memory, streams and the autograd tape only -- all arithmetic on [N, M, C] mesh activations
happens in the HIP kernels.  Activations are fp32 ``[N, M, C]`` *views* of buffers whose row
stride is padded to a multiple of 4 floats (16-byte aligned rows for float4 gathers).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, CapeSrc, check
from .graph import ConvOperators, HostCSR

# One evaluation form per layer: sparse operators applied by the streaming spmm kernels, dense contraction as a plain GEMM
# ("two-pass").  The single-launch form of rounds 1-5 (operators gathered inside the GEMM's A-tile staging for WHOLE layers,
# CAPE_MODE=fused) was slower at every mesh level and reached only 8 x the fp32 restatement's error on one data gradient where
# the contract is 4 x (one fp32 MFMA chain per output element); it is gone.  The gather-form kernels themselves remain what the
# library launches for sources that cannot be staged plainly (the 3-channel output layer), through the same C entry point.
import os as _os
# polynomial orders above the precomposed-operator limit: 1 = recurrence on chip where the layer qualifies
# (csrc/cheb_fused.hip), 0 = always the materialised K-stack (ChebConvRecurrenceFn; the A/B reference)
FUSED_RECURRENCE = int(_os.environ.get("CAPE_FUSED_RECURRENCE", "1"))
# GraphCMR decoder block: 1 = its two closing 1x1 filters, the addition and the condition concat as one two-source
# contraction (ResidualLinearFn), 0 = the reference's op-by-op formulation (the A/B reference)
CMR_FUSED_TAIL = int(_os.environ.get("CAPE_CMR_FUSED_TAIL", "1"))
# adversarial step: D(generated) and D(real) as one pass over the concatenated batch ("1"), as two passes the way the
# reference builds them ("0", the A/B reference; the bug-compatible mode always takes two), or "auto" (default): merged
# while the concatenated batch is at most 32 meshes -- measured: batch 16 + 16: 4.29 -> 4.19 ms (the discriminator's small
# layers are launch-bound, half the launches win); batch 32 + 32: 13.66 -> 13.73 ms (no longer launch-bound, and the
# generator's sweep through D then carries the real half along)
MERGED_D_PASS = _os.environ.get("CAPE_MERGED_D_PASS", "auto")


def merged_d_pass(batch):
    if MERGED_D_PASS == "auto":
        return 2 * int(batch) <= 32
    return bool(int(MERGED_D_PASS))
# sparse operators with at most 12 entries per row are handed to the streaming kernels in ELL form (no row pointer in
# the dependent-load chain, csrc/elementwise.hip cape_gather_row_ell); 0 = always CSR (the A/B reference, same sums bit for bit)
SPMM_ELL = int(_os.environ.get("CAPE_SPMM_ELL", "1"))

_ACT_OF = {"b1leakyrelu": ("leaky", _lib.BIAS_CHANNEL), "b1relu": ("relu", _lib.BIAS_CHANNEL),
           "b1tanh": ("tanh", _lib.BIAS_CHANNEL), "b2relu": ("relu", _lib.BIAS_VERTEX)}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pad4(c):
    return (int(c) + 3) // 4 * 4


ACT_DTYPES = (torch.float32, torch.bfloat16)     # activation storage types (bf16: BASELINE configs[4], the *_bf16 C entry points)


def alloc_act(N, M, Cn, device, zero=False, dtype=torch.float32):
    """[N, M, C] view of a fresh buffer whose rows are padded to 16 bytes (4 fp32 / 8 bf16 elements)."""
    q = 4 if dtype == torch.float32 else 8
    ld = (int(Cn) + q - 1) // q * q
    buf = (torch.zeros if zero else torch.empty)((N, M, ld), device=device, dtype=dtype)
    return buf[:, :, :Cn] if ld != Cn else buf


def _fn(name, t):
    """The C entry point ``name`` for the storage type of activation tensor ``t``."""
    return getattr(lib, name + "_bf16") if t.dtype == torch.bfloat16 else getattr(lib, name)


def as_act(t):
    """Make ``t`` [N, M, C] usable by the kernels (unit channel stride, fp32 or bf16 storage, on device)."""
    assert t.dim() == 3 and t.dtype in ACT_DTYPES and t.is_cuda
    if t.stride(2) != 1 and t.shape[2] != 1:
        t = t.contiguous()
    if t.shape[1] > 1 and t.stride(1) < t.shape[2]:
        t = t.contiguous()
    return t


def _row_aligned(x):
    """``x`` itself when its rows start on 16-byte boundaries, else a copy in a row-padded buffer (the kernels use aligned
    16-byte accesses; e.g. a contiguous [N, M, 3] or [N, M, 262] tensor handed in from outside the layer stack)."""
    q = 4 if x.dtype == torch.float32 else 8
    if (x.stride(1) % q) or (x.shape[0] > 1 and (x.stride(0) % q)) or (x.data_ptr() & 15):
        xp = alloc_act(x.shape[0], x.shape[1], x.shape[2], x.device, zero=True, dtype=x.dtype)
        xp.copy_(x)
        return xp
    return x


def _v(t):
    """(ptr, sample_stride, ld) of an activation view."""
    ld = t.stride(1) if t.shape[1] > 1 else max(t.shape[2], t.stride(1))
    ss = t.stride(0) if t.shape[0] > 1 else t.shape[1] * ld
    return C.c_void_p(t.data_ptr()), int(ss), int(ld)


def _vec_ok(*views):
    """The condition under which the streaming sparse kernels take their vector form (csrc/elementwise.hip aligned4): four
    consecutive elements of every row addressable as one access.  Only that form reads ELL operands."""
    for t in views:
        if t is None:
            continue
        p, ss, ld = _v(t)
        if (p.value % (4 * t.element_size())) or (ss & 3) or (ld & 3) or (t.shape[2] & 3):
            return False
    return True


def _ptr(t, offset_elems=0):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr() + 4 * int(offset_elems))


def ell_arrays(host):
    """ELL form of a host CSR operator for the streaming sparse kernels: width 4 / 8 / 12, entries in CSR order packed to the
    front, the slots past a row's end = (column of slot 0, 0.0) -- an empty row gets column 0.  Returns (cols, vals) or None.
    The kernel (csrc/elementwise.hip cape_gather_row_ell) stops at the first group of four whose values are all zero, so an
    operator with a stored all-zero group IN FRONT of a non-zero one (explicit zeros, cancellation in a precomposed chain)
    would lose entries: such operators keep the CSR form."""
    if not (1 <= host.max_row <= 12):
        return None
    w = (int(host.max_row) + 3) // 4 * 4
    rows = host.shape[0]
    rp = host.rowptr.astype(np.int64)
    deg = (rp[1:] - rp[:-1]).astype(np.int64)
    first = np.where(deg > 0, host.colidx[np.minimum(rp[:-1], max(host.nnz - 1, 0))], 0).astype(np.int32)
    ec = np.repeat(first[:, None], w, axis=1)
    ev = np.zeros((rows, w), dtype=np.float32)
    slot = np.arange(host.nnz, dtype=np.int64) - np.repeat(rp[:-1], deg)
    rr = np.repeat(np.arange(rows, dtype=np.int64), deg)
    ec[rr, slot] = host.colidx
    ev[rr, slot] = host.vals
    live = (ev.reshape(rows, w // 4, 4) != 0).any(axis=2)                # [rows, groups]
    if w > 4 and bool((~live[:, :-1] & (np.cumsum(live[:, ::-1], axis=1)[:, ::-1][:, 1:] > 0)).any()):
        return None
    return np.ascontiguousarray(ec), np.ascontiguousarray(ev)


class DeviceCSR(object):
    def __init__(self, host, device):
        assert isinstance(host, HostCSR)
        self.shape = host.shape
        self.identity = host.identity
        self.nnz = host.nnz
        self.max_row, self.min_row = host.max_row, host.min_row
        if host.identity:
            self.rowptr = self.colidx = self.vals = None
        else:
            rc = lib.cape_csr_validate(host.shape[0], host.shape[1], host.nnz,
                                       host.rowptr.ctypes.data_as(C.c_void_p),
                                       host.colidx.ctypes.data_as(C.c_void_p))
            check(rc, "cape_csr_validate")
        # identity operators still keep real arrays for the standalone spmm path
        self.rowptr_t = torch.from_numpy(host.rowptr).to(device)
        self.colidx_t = torch.from_numpy(host.colidx).to(device)
        self.vals_t = torch.from_numpy(host.vals).to(device)
        if not host.identity:
            self.rowptr, self.colidx, self.vals = self.rowptr_t, self.colidx_t, self.vals_t
        # ELL form for the streaming sparse kernels (None when the operator does not qualify)
        self.ell_w, self.ell_col_t, self.ell_val_t = 0, None, None
        ell = ell_arrays(host)
        if ell is not None:
            self.ell_w = int(ell[0].shape[1])
            self.ell_col_t = torch.from_numpy(ell[0]).to(device)
            self.ell_val_t = torch.from_numpy(ell[1]).to(device)

    def operands(self):
        """(rowptr, colidx, vals, ell_width) pointers for the streaming sparse kernels: the ELL arrays when the operator
        has them and the knob is on, the CSR arrays otherwise."""
        if SPMM_ELL and self.ell_w:
            return self.rowptr_t.data_ptr(), self.ell_col_t.data_ptr(), self.ell_val_t.data_ptr(), self.ell_w
        return self.rowptr_t.data_ptr(), self.colidx_t.data_ptr(), self.vals_t.data_ptr(), 0


class DeviceConvOps(object):
    """Device copy of graph.ConvOperators."""

    def __init__(self, host, device):
        assert isinstance(host, ConvOperators)
        self.K, self.fused = host.K, host.fused
        self.Mi, self.Mo = host.Mi, host.Mo
        if host.fused:
            self.fwd = [DeviceCSR(h, device) for h in host.fwd]
            self.bwd = [DeviceCSR(h, device) for h in host.bwd]
            # row sums S_k 1 (rank-1 handling of vertex-constant condition channels); the last row
            # repeats S_0 1 for the affine (K=1) branch of res_block_affine
            rs = host.cond_row_terms()
            self.rowscale = torch.from_numpy(np.stack(rs + [rs[0]]).astype(np.float32)).to(device).contiguous()
        else:
            self.Lt = DeviceCSR(host.Lt, device)
            self.LtT = DeviceCSR(host.LtT, device)
        self.host = host
        self.device = device
        self._patch_plans = {}

    def patch_plan(self, Cin, Fout):
        """Device copy of graph.ChebPatchPlan for the on-chip recurrence (csrc/cheb_fused.hip), or None when the layer
        does not qualify: precomposed operators (K <= 3), pool / unpool around the layer, an asymmetric operator,
        channel counts outside the kernel's set, or no patch plan that fits the LDS."""
        key = (int(Cin), int(Fout))
        if key not in self._patch_plans:
            plan = None
            host = self.host
            ok = (not self.fused and host.unfused_unpool is None and host.unfused_pool is None
                  and lib.cape_cheb_fused_supported(int(Cin), int(Fout), int(self.K)) == 1)
            if ok:
                from .graph import ChebPatchPlan
                Lt = host.Lt.to_scipy()
                if abs(Lt - Lt.T).max() <= 1e-6 * max(abs(Lt).max(), 1e-30):       # the adjoint kernel applies L~ itself
                    try:
                        plan = DevicePatchPlan(ChebPatchPlan(Lt, self.K, Cin, reserve_bytes=8 * 4 * int(Cin) * int(Fout)), self.device)
                    except ValueError:
                        plan = None
            self._patch_plans[key] = plan
        return self._patch_plans[key]


class DevicePatchPlan(object):
    def __init__(self, host, device):
        self.host = host
        self.P, self.rmax, self.K, self.M = host.P, host.rmax, host.K, host.M
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
        self.pinfo, self.vid, self.ell_col, self.ell_val = t(host.pinfo), t(host.vid), t(host.ell_col), t(host.ell_val)


def _mk_srcs(entries):
    """entries: list of dict(x=view, csr=DeviceCSR|None, C=int, w=(tensor, off, rs, cs), w2=... or None)."""
    arr = (CapeSrc * len(entries))()
    for s, e in zip(arr, entries):
        x = e["x"]
        p, ss, ld = _v(x)
        s.x, s.x_sample_stride, s.ldx, s.C = p, ss, ld, int(e.get("C", x.shape[2]))
        csr = e.get("csr")
        if csr is not None and not csr.identity:
            s.rowptr, s.colidx, s.vals = csr.rowptr.data_ptr(), csr.colidx.data_ptr(), csr.vals.data_ptr()
        else:
            s.rowptr = s.colidx = s.vals = None
        wt, off, rs, cs = e["w"]
        s.w, s.w_rs, s.w_cs = wt.data_ptr() + 4 * int(off), int(rs), int(cs)
        if e.get("w2") is not None:
            wt2, off2, rs2, cs2 = e["w2"]
            s.w2, s.w2_rs, s.w2_cs = wt2.data_ptr() + 4 * int(off2), int(rs2), int(cs2)
        else:
            s.w2, s.w2_rs, s.w2_cs = None, 0, 0
    return arr


# --------------------------------------------------------------------------------------------
# per-launch instrumentation (bench.py's roofline leg and the kernel-coverage parity test)
#   LAUNCH_LOG: a list -> every C-ABI compute call is bracketed by HIP events on the launch stream and logged as
#               (kernel name as rocprofv3 prints it, algorithmic flops, algorithmic bytes, event0, event1)
#   PLAN_LOG:   a set  -> the kernel selection the library reports for every contraction launch is recorded as
#               ("fwd", family, BM, BN, layout, dual) / ("dw", family, CT, FT)
# Both None (the default): no overhead, no extra calls.
# --------------------------------------------------------------------------------------------
LAUNCH_LOG = None
PLAN_LOG = None
# ACT_TRACE: a list -> every (leaky-)ReLU site of a forward pass appends the branch each unit took (bool tensor, True =
# positive side), in execution order.  The gradient-parity tests replay that pattern in the fp64 twin, so that a unit whose
# pre-activation lies within fp32 rounding of zero takes the SAME branch on both sides (tests/test_gpu_model.py).
ACT_TRACE = None
L1_SIGN_TRACE = None      # a list -> the L1 reconstruction loss appends sign(pred - gt) (the one other branch point of the graph)


def _trace_sign(y, act):
    if ACT_TRACE is not None and act in ("leaky", "relu"):
        ACT_TRACE.append((y.detach() > 0).cpu())


def _trace_mask_bits(mask, F):
    """Unpack the 1-bit ReLU mask [N, Mo, ceil(F/32)] the DUAL kernels write."""
    if ACT_TRACE is not None:
        sh = torch.arange(32, device=mask.device, dtype=torch.int32)
        bits = ((mask.unsqueeze(-1) >> sh) & 1).reshape(mask.shape[0], mask.shape[1], -1)[:, :, :F]
        ACT_TRACE.append((bits != 0).cpu())


def _gconv_work(entries, N, Mo, F):
    """Algorithmic work of one gather-GEMM launch: dense contraction 2*N*Mo*C*F per weight block plus
    2*N*nnz*C for the sparse operator application; bytes = operands touched once (fp32)."""
    es = entries[0]["x"].element_size()
    flops, byts = 0, es * N * Mo * F
    seen = set()
    for e in entries:
        Cs = int(e.get("C", e["x"].shape[2]))
        nblk = 2 if e.get("w2") is not None else 1
        flops += 2 * N * Mo * Cs * F * nblk
        csr = e.get("csr")
        if csr is not None and not csr.identity:
            flops += 2 * N * csr.nnz * Cs
            byts += 8 * csr.nnz + 4 * (csr.shape[0] + 1)
        byts += 4 * Cs * F * nblk
        key = e["x"].data_ptr()
        if key not in seen:
            seen.add(key)
            byts += es * N * e["x"].shape[1] * Cs
    return flops, byts


def _dw_work(entries, N, Mo, F, two_dz):
    """Weight-gradient launch: 2*N*Mo*C*F per source; every distinct source and gradient operand read once, every
    gradient block written once."""
    es = entries[0]["x"].element_size()
    flops, byts = 0, es * N * Mo * F * (2 if two_dz else 1)
    seen = set()
    for e in entries:
        Cs = int(e.get("C", e["x"].shape[2]))
        flops += 2 * N * Mo * Cs * F
        byts += 4 * Cs * F
        key = e["x"].data_ptr()
        if key not in seen:
            seen.add(key)
            byts += es * N * e["x"].shape[1] * Cs
    return flops, byts


def _csr_bytes(csr):
    return 0 if (csr is None or csr.identity) else 8 * csr.nnz + 4 * (csr.shape[0] + 1)


# per-launch timing (bench.py): a spacer keeps the stream busy while the start event, the launch and the stop event are
# queued, so that the bracket holds the kernel alone -- without it the host's launch path (5-25 us of argument marshalling)
# sits between the two events of every kernel that finds the GPU idle
LOG_SPACER_US = int(_os.environ.get("CAPE_LOG_SPACER_US", "80"))


def _log_launch(name, flops, byts, fn):
    if LAUNCH_LOG is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if LOG_SPACER_US:
        lib.cape_spin_us(LOG_SPACER_US, _stream())
    e0.record()
    out = fn()
    e1.record()
    LAUNCH_LOG.append((name, int(flops), int(byts), e0, e1))
    return out


_FWD_FAMILY = {0: "gconv_fwd_kernel", 1: "gemm_plain_kernel", 2: "gemm_split_kernel", 3: "gemm_h2_kernel", 4: "fwd_narrow_in_kernel"}
_DW_FAMILY = {0: "gconv_dw_kernel", 1: "dw_plain_kernel", 2: "dw_packed_kernel", 3: "dw_split_kernel", 4: "dw_h2_kernel",
              5: "dw_narrow_in_kernel", 6: "dw_narrow_out_kernel"}


def fwd_kernel_name(fam, bm, bn, layout, dual, bf16=False):
    """Kernel instantiation name as rocprofv3 prints it."""
    waves = "2, 2" if (bm, bn) in ((64, 128), (128, 128), (64, 64)) else "4, 1"
    tf = lambda b: "true" if b else "false"
    at = ", unsigned short" if bf16 else ", float"
    if fam == 4:
        return "fwd_narrow_in_kernel"
    if fam == 3:
        if (bm, bn) == (128, 256) and not dual:
            return "gemm_h2x_kernel<0, 0>"                  # the wide tile (csrc/gemm_h2x.h)
        return "gemm_h2_kernel<%d, %d, %s>" % (bm, bn, tf(dual))
    if fam == 2:
        return "gemm_split_kernel<%d, %d, %s, %s%s>" % (bm, bn, tf(layout), tf(dual), at)
    if fam == 1:
        return "gemm_plain_kernel<%d, %d, %s, %s, %s>" % (bm, bn, waves, tf(dual), tf(layout))
    return "gconv_fwd_kernel<%d, %d, %s, %s, 32%s>" % (bm, bn, waves, tf(dual), at)


def dw_kernel_name(fam, ct, ft, bf16=False):
    if fam in (5, 6):
        return _DW_FAMILY[fam]
    if fam == 4:
        return "dw_h2_kernel<%d, %d, %s>" % (ct, ft, "true" if (DW_V4 and (ct, ft) == (128, 128)) else "false")
    if fam == 3 or fam == 0:
        return "%s<%d, %d, %s>" % (_DW_FAMILY[fam], ct, ft, "unsigned short" if bf16 else "float")
    waves = "4, 1" if (fam == 2 and ft == 32) else "2, 2"
    return "%s<%d, %d, %s, %s>" % (_DW_FAMILY[fam], ct, ft, waves, "unsigned short" if bf16 else "float")


# --------------------------------------------------------------------------------------------
# sole-consumer chains: inside ``with sole_consumer_chain():`` every ChebConvFn may assume that its INPUT tensor has no other
# consumer (the model's plain encoder stack, reference lib/models.py:541-545: x = cnp(x) in a loop).  Its backward then hands
# the layer below dz = dx * act'(x) instead of dx (the derivative of that layer's own fused bias + (leaky-)ReLU epilogue, taken
# from the sign of x) together with the bias-gradient partial sums, all from the epilogue of the summed operator application
# that produces dx (csrc/elementwise.hip spmm_multi_kernel, SpmmActGrad) -- and that layer skips its backward-prep launch.
# FUSE_ACT_GRAD = 0 keeps the op-by-op form (the A/B reference).
# --------------------------------------------------------------------------------------------
FUSE_ACT_GRAD = int(_os.environ.get("CAPE_FUSE_ACT_GRAD", "1"))
# FUSE_PREP_SPMM = 0: the affine blocks at one resolution run cape_bwd_prep and cape_spmm instead of cape_bwd_prep_spmm (the A/B form)
FUSE_PREP_SPMM = int(_os.environ.get("CAPE_FUSE_PREP_SPMM", "1"))
DW_V4 = int(_os.environ.get("CAPE_DW_V4", "1"))        # mirror of the library's switch (csrc/gemm_h2.h h2_dw_launch): kernel NAMES only
_CHAIN = [False]


class _ActOffer(object):
    """What a layer attaches to its output (``_cape_act_out``) when its consumer may differentiate its bias + activation
    epilogue for it.  The hand-over travels on Python attributes of autograd tensors, so both ends check it: the consumer's
    backward marks ``fused`` (it multiplied by act' and queued the bias partials) and tags the gradient with the version
    counter it had then (``_cape_is_dz``); the offering layer's backward refuses a tagged gradient that was written since (the
    autograd engine sums a second gradient INTO a tensor it owns) and a fused hand-over whose tag did not arrive (a hook or a
    copy dropped it: act' would be applied twice and the bias partials queued twice)."""
    __slots__ = ("act", "gB", "fused")

    def __init__(self, act, gB):
        self.act, self.gB, self.fused = act, gB, False


class sole_consumer_chain(object):
    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev, _CHAIN[0] = _CHAIN[0], self.on
        return self

    def __exit__(self, *exc):
        _CHAIN[0] = self.prev
        return False


# --------------------------------------------------------------------------------------------
# fp16 two-piece contractions (csrc/gemm_h2.h): row bounds of activation tensors and piece planes of the weights
# --------------------------------------------------------------------------------------------
# H2 = 0 keeps every contraction on the bf16 six-product kernels (the A/B reference of round 3)
H2 = int(_os.environ.get("CAPE_H2", "1"))
RM_TRACE = int(_os.environ.get("CAPE_RM_TRACE", "0"))      # debugging: print who needed a standalone row-bound pass


def rm_width(F):
    """Entries per row of a row-bound tensor for ``F`` channels: one per 32-column block, padded to a multiple of 4."""
    return ((int(F) + 31) // 32 + 3) // 4 * 4


def alloc_rm(t, F=None):
    """Fresh row-bound tensor [N, M, W] for activation tensor ``t`` (filled by the kernel that produces ``t``)."""
    return torch.empty((t.shape[0], t.shape[1], rm_width(t.shape[2] if F is None else F)), device=t.device, dtype=torch.float32)


def set_rm(t, rm):
    """Attach the row bounds ``rm`` to activation tensor ``t``.  The attribute travels with the tensor object through
    autograd.Function.apply, saved tensors and the backward pass (PyTorch preserves the Python object of a live tensor);
    views are new objects and do not inherit it, and every in-place writer of ``t`` must drop it (``drop_rm``)."""
    if rm is not None:
        t._cape_rm = rm
        t._cape_rm_version = t._version           # (see rm_of)
    return t


def drop_rm(t):
    if getattr(t, "_cape_rm", None) is not None:
        t._cape_rm = None


def rm_of(t):
    rm = getattr(t, "_cape_rm", None)
    if rm is None or rm.shape[0] != t.shape[0] or rm.shape[1] != t.shape[1] or rm.device != t.device:
        return None
    # the bounds describe the contents at attach time: an in-place torch write since then -- e.g. the autograd engine summing a
    # second gradient INTO this tensor (it accumulates in place when it holds the only reference) -- bumps the version counter
    # and voids them (a bound that is too small overflows fp16).  Kernels of this library that write through raw pointers do not
    # bump it: those call sites use drop_rm.
    if getattr(t, "_cape_rm_version", None) != t._version:
        return None
    return rm


def h2_shape_ok(ktot, F):
    """Mirror of csrc/gemm_h2.h h2_eligible's size rule: short contractions stay on the six-product kernel, so their operands
    need no row bounds."""
    return F >= 64 and (ktot >= 256 or (ktot >= 128 and F >= 128))


def _want_rm(t, Cn=None, any_width=False):
    """Row bounds are worth writing for an output that can be an operand of the fp16 two-piece contraction: fp32, whole
    32-channel chunks -- and, for the streaming sparse kernels, a power-of-two channel count (a row is then a lane group):
    with other widths (the 288 / 544 channels of [features | condition]) they would add a standalone pass for every output,
    wanted or not, so the consumer asks for one lazily instead (``rowmax``).  ``any_width``: producers that bound any row in
    their own launch (group norm's wave-per-row apply pass)."""
    Cn = t.shape[2] if Cn is None else Cn
    if not (bool(H2) and t.dtype == torch.float32 and Cn % 32 == 0 and Cn >= 32):
        return False
    return any_width or (Cn & (Cn - 1)) == 0


def _new_rm(t):
    return torch.empty((t.shape[0], t.shape[1], 4), device=t.device, dtype=torch.float32)


def rowmax(t):
    """Row bounds of ``t``: the ones its producer attached, else one standalone pass (csrc/pieces.hip rowmax_kernel)."""
    rm = rm_of(t)
    if rm is not None:
        return rm
    _lib.require_gpu()
    t = as_act(t)
    assert t.dtype == torch.float32
    if RM_TRACE:
        import traceback
        print("standalone rowmax for", tuple(t.shape), "<-", " <- ".join("%s:%d" % (f.name, f.lineno) for f in traceback.extract_stack()[-6:-1]), flush=True)
    rm = torch.empty((t.shape[0], t.shape[1], 4), device=t.device, dtype=torch.float32)
    p, ss, ld = _v(t)
    _log_launch("rowmax_kernel", 0, 4 * t.shape[0] * t.shape[1] * (t.shape[2] + 4),
                lambda: check(lib.cape_rowmax(p, ss, ld, t.shape[0], t.shape[1], t.shape[2], _ptr(rm), 4, _stream()), "cape_rowmax"))
    set_rm(t, rm)
    return rm


class WeightPieces(object):
    """Piece planes of one Chebyshev-layer weight W[Ch*K (+ condition rows), F] (views of a PiecePlan's arena):
    forward planes [K][F][Ch], backward planes [Ch*K][F], reciprocal scales (include/cape_hip.h cape_wpiece_item_t)."""
    __slots__ = ("W", "Ch", "K", "F", "pair", "fpair", "f_hi", "f_lo", "b_hi", "b_lo", "fsi", "bsi", "bsc")

    def fwd(self, k):
        """(hi pointer, lo pointer, pitch) of forward source k (contraction over the Ch feature channels)."""
        o = 2 * k * self.F * self.Ch
        return self.f_hi.data_ptr() + o, self.f_lo.data_ptr() + o, self.Ch

    def bwd(self, k):
        """Data-gradient source k (contraction over F, output column = channel c): rows c*K + k of the backward planes."""
        o = 2 * k * self.F
        return self.b_hi.data_ptr() + o, self.b_lo.data_ptr() + o, self.K * self.F


class PiecePlan(object):
    """All piece planes of a set of weights in two launches (cape_weight_pieces).  ``specs``: dicts W (tensor), Ch, K and
    optionally pair = (W2, K2): a second weight whose data-gradient term adds into the same accumulator; fpair = (W2, rows): a
    second weight whose FORWARD product adds into the same accumulator."""

    def __init__(self, specs, device):
        self.items, sizes = [], []
        for sp in specs:
            W, Ch, K = sp["W"], int(sp["Ch"]), int(sp["K"])
            F = int(W.shape[1])
            assert W.is_contiguous() and W.dtype == torch.float32 and W.shape[0] >= Ch * K and Ch % 8 == 0 and F % 8 == 0
            sizes.append((Ch, K, F))
        al = lambda nbytes: (nbytes + 255) // 256 * 256
        rpatch = lambda sp, Ch, K: (Ch * K + (int(sp["fpair"][1]) if sp.get("fpair") is not None else 0) + 63) // 64
        total = sum(4 * al(2 * Ch * K * F) + al(4 * K * F) + al(4 * Ch * K) + al(4 * Ch) + al(4 * rpatch(sp, Ch, K) * F)
                    for sp, (Ch, K, F) in zip(specs, sizes))
        self.arena = torch.empty(max(total, 256), device=device, dtype=torch.uint8)
        off = 0

        def take(nbytes, dtype):
            nonlocal off
            t = self.arena[off:off + nbytes].view(dtype)
            off += al(nbytes)
            return t

        arr = (_lib.CapeWpieceItem * len(specs))()
        for a, sp, (Ch, K, F) in zip(arr, specs, sizes):
            wp = WeightPieces()
            wp.W, wp.Ch, wp.K, wp.F, wp.pair, wp.fpair = sp["W"], Ch, K, F, sp.get("pair"), sp.get("fpair")
            wp.f_hi, wp.f_lo = take(2 * Ch * K * F, torch.int16), take(2 * Ch * K * F, torch.int16)
            wp.b_hi, wp.b_lo = take(2 * Ch * K * F, torch.int16), take(2 * Ch * K * F, torch.int16)
            wp.fsi, wp.bsi, wp.bsc = take(4 * K * F, torch.float32), take(4 * Ch * K, torch.float32), take(4 * Ch, torch.float32)
            a.w, a.Ch, a.K, a.F = sp["W"].data_ptr(), Ch, K, F
            a.pair_w, a.pair_K = (wp.pair[0].data_ptr(), int(wp.pair[1])) if wp.pair is not None else (None, 0)
            a.fpair_w, a.fpair_rows = (wp.fpair[0].data_ptr(), int(wp.fpair[1])) if wp.fpair is not None else (None, 0)
            a.f_hi, a.f_lo, a.b_hi, a.b_lo = wp.f_hi.data_ptr(), wp.f_lo.data_ptr(), wp.b_hi.data_ptr(), wp.b_lo.data_ptr()
            a.fscale_inv, a.bscale_inv, a.bscale_c_inv = wp.fsi.data_ptr(), wp.bsi.data_ptr(), wp.bsc.data_ptr()
            a.colmax_partial = take(4 * rpatch(sp, Ch, K) * F, torch.float32).data_ptr()
            self.items.append(wp)
        n = len(specs)
        mo, po = (C.c_int32 * (n + 1))(), (C.c_int32 * (n + 1))()
        check(lib.cape_weight_pieces_blocks(C.addressof(arr), n, mo, po), "cape_weight_pieces_blocks")
        self.n, self.max_blocks, self.planes_blocks = n, int(mo[n]), int(po[n])
        raw = np.frombuffer(bytes(arr), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(device)
        self.max_off = torch.tensor(list(mo), dtype=torch.int32, device=device)
        self.planes_off = torch.tensor(list(po), dtype=torch.int32, device=device)
        self.flops_bytes = sum(12 * Ch * K * F for Ch, K, F in sizes)

    def run(self):
        _lib.require_gpu()
        _log_launch("wplanes_kernel", 0, self.flops_bytes,
                    lambda: check(lib.cape_weight_pieces(_ptr(self.table), self.n, _ptr(self.max_off), self.max_blocks,
                                                         _ptr(self.planes_off), self.planes_blocks, _stream()), "cape_weight_pieces"))
        return self.items


# weight data_ptr -> WeightPieces of the model-level plan (models.CAPE keeps it fresh: re-run at the start of every
# forward pass); weights outside it get their planes on demand (two small launches per call: unit tests, first trace)
PIECES = {}


def drop_pieces(ptrs):
    """Remove registry entries (a model's finaliser: the entries hold its weights and planes)."""
    for q in ptrs:
        PIECES.pop(q, None)


def pieces_for(W, Ch, K, pair=None, fpair=None):
    """Piece planes of W for ``Ch`` feature channels at order K, or None when the layer does not qualify."""
    if not H2 or not W.is_cuda or W.dtype != torch.float32 or Ch % 8 or W.shape[1] % 8 or not W.is_contiguous():
        return None
    ptr = lambda pr: pr[0].data_ptr() if pr is not None else None
    wp = PIECES.get(W.data_ptr())
    if wp is not None and (wp.Ch, wp.K, wp.F) == (Ch, K, int(W.shape[1])) and ptr(wp.pair) == ptr(pair) and ptr(wp.fpair) == ptr(fpair):
        return wp
    return PiecePlan([dict(W=W.detach(), Ch=Ch, K=K, pair=pair, fpair=fpair)], W.device).run()[0]


def _h2_arg(entries, N, wsi=None, wsi2=None, rm_out=None):
    """cape_h2_t for a launch: entries carry "p" = (hi, lo, pitch) [and "p2"] and "rm" when the launch may take the fp16
    two-piece kernel; ``rm_out``: row-bound tensor the epilogue fills.  Returns (struct or None, keep-alive)."""
    full = wsi is not None and all(e.get("p") is not None and e.get("rm") is not None for e in entries)
    if not full and rm_out is None:
        return None, None
    h = _lib.CapeH2()
    keep = None
    if full:
        arr = (_lib.CapeH2Src * len(entries))()
        for a, e in zip(arr, entries):
            a.w_hi, a.w_lo, a.w_pitch = e["p"]
            if e.get("p2") is not None:
                a.w2_hi, a.w2_lo, a.w2_pitch = e["p2"]
            rm = e["rm"]
            assert rm.is_contiguous() and rm.shape[0] == N and rm.shape[1] == e["x"].shape[1]
            a.rowmax, a.rowmax_w = rm.data_ptr(), int(rm.shape[2])
        h.src, keep = C.addressof(arr), arr
        h.wscale_inv = wsi
        h.w2scale_inv = wsi2
    if rm_out is not None:
        h.rowmax_out, h.rowmax_out_w = rm_out.data_ptr(), int(rm_out.shape[2])
    return h, keep


# --------------------------------------------------------------------------------------------
# raw kernel wrappers (no autograd)
# --------------------------------------------------------------------------------------------
def gconv_fwd(entries, y, bias=None, bias_mode=_lib.BIAS_NONE, act="none", mask=None, rank=None, deinterleave=0, F=None,
              wsi=None, wsi2=None, rm_out=None):
    """rank = (rowscale [R,Mo], coef [N,R,F], to_acc2 bitmask) or None.  ``deinterleave`` = K: the launch computes
    ``F`` = K*C output columns (column c*K + k) and stores them as K channel blocks of ``y`` [N, Mo, K*round_up(C,4)].
    fp16 two-piece operands (csrc/gemm_h2.h): entries with "p" / "p2" (piece planes) and "rm" (row bounds) plus ``wsi`` /
    ``wsi2`` (pointers to the reciprocal column scales); ``rm_out``: row-bound tensor of y for the epilogue to fill."""
    _lib.require_gpu()
    arr = _mk_srcs(entries)
    N, Mo = y.shape[0], y.shape[1]
    F = int(y.shape[2]) if F is None else int(F)
    p, ss, ld = _v(y)
    rk = None
    if rank is not None:
        rowscale, coef, to2 = rank
        assert coef.is_contiguous() and rowscale.is_contiguous() and coef.shape[0] == N and coef.shape[2] == F
        assert rowscale.shape[1] == Mo and rowscale.shape[0] >= coef.shape[1]
        rk = _lib.CapeRank(int(coef.shape[1]), rowscale.data_ptr(), coef.data_ptr(), int(to2))

    assert all(e["x"].dtype == y.dtype for e in entries), "sources and output share one storage type"

    h2, h2_keep = (None, None) if y.dtype != torch.float32 else _h2_arg(entries, N, wsi, wsi2, rm_out)

    def launch():
        if h2 is not None:
            rc = lib.cape_gconv_fwd_h2(arr, len(entries), p, ss, ld, N, Mo, F, _ptr(bias),
                                       bias_mode if bias is not None else _lib.BIAS_NONE, _lib.ACT[act],
                                       _ptr(mask), C.byref(rk) if rk is not None else None, int(deinterleave), C.byref(h2), _stream())
            check(rc, "cape_gconv_fwd_h2")
            return
        rc = _fn("cape_gconv_fwd", y)(arr, len(entries), p, ss, ld, N, Mo, F, _ptr(bias),
                                bias_mode if bias is not None else _lib.BIAS_NONE, _lib.ACT[act],
                                _ptr(mask), C.byref(rk) if rk is not None else None, int(deinterleave), _stream())
        check(rc, "cape_gconv_fwd")

    if LAUNCH_LOG is None and PLAN_LOG is None:
        launch()
    else:
        # the library reports the kernel it selects; names as rocprofv3 prints them
        dual = any(e.get("w2") is not None for e in entries)
        plan = (C.c_int32 * 4)()
        if h2 is not None:
            check(lib.cape_gconv_fwd_plan_h2(arr, len(entries), N, Mo, F, C.byref(h2), plan), "cape_gconv_fwd_plan_h2")
        else:
            check(_fn("cape_gconv_fwd_plan", y)(arr, len(entries), N, Mo, F, plan), "cape_gconv_fwd_plan")
        fam, bm, bn, layout = list(plan)
        bf = y.dtype == torch.bfloat16
        if PLAN_LOG is not None:
            PLAN_LOG.add(("fwd", fam, bm, bn, layout, int(dual)) + (("bf16",) if bf else ()))
        flops, byts = _gconv_work(entries, N, Mo, F)
        _log_launch(fwd_kernel_name(fam, bm, bn, layout, dual, bf), flops, byts, launch)
    return y


def gconv_dw(entries, dz, accumulate=False, dz2=None, defer=False):
    """entries' ``w`` fields name the gradient blocks to write; entries with ``use_dz2`` contract against
    ``dz2`` (same shape and strides as ``dz``) instead of ``dz``.  ``defer`` (honoured while DEFERRED is a list): only
    the contraction runs now; the fixed-order slab reduction is queued and flush_deferred() performs the reductions of
    all queued layers in batched launches -- legal when nothing reads the gradient blocks before the flush (they are
    views of the flat gradient bucket)."""
    _lib.require_gpu()
    arr = _mk_srcs(entries)
    N, Mo, F = dz.shape
    need = lib.cape_gconv_dw_workspace_bytes(arr, len(entries), N, Mo, F)
    if need < 0:
        check(int(need), "cape_gconv_dw_workspace_bytes")
    ws = torch.empty((need + 3) // 4, device=dz.device, dtype=torch.float32)
    p, ss, ld = _v(dz)
    mask, p2 = 0, None
    if dz2 is not None:
        p2, ss2, ld2 = _v(dz2)
        assert (ss2, ld2) == (ss, ld) and dz2.shape == dz.shape
        for i, e in enumerate(entries):
            if e.get("use_dz2"):
                mask |= 1 << i
    assert all(e["x"].dtype == dz.dtype for e in entries), "sources and gradient share one storage type"
    bf = dz.dtype == torch.bfloat16
    # fp16 two-piece form (csrc/gemm_h2.h dw_h2_kernel): taken when every operand already carries its row bounds
    h2 = None
    if H2 and not bf:
        rms = [rm_of(e["x"]) for e in entries] + [rm_of(dz)] + ([rm_of(dz2)] if dz2 is not None else [])
        if all(r is not None for r in rms):
            h2 = _lib.CapeH2Dw()
            for i, r in enumerate(rms[:len(entries)]):
                h2.src_rowmax[i], h2.src_rowmax_w[i] = r.data_ptr(), int(r.shape[2])
            h2.dz_rowmax, h2.dz_rowmax_w = rms[len(entries)].data_ptr(), int(rms[len(entries)].shape[2])
            if dz2 is not None:
                h2.dz2_rowmax, h2.dz2_rowmax_w = rms[-1].data_ptr(), int(rms[-1].shape[2])
            h2._keep = rms

    def dw_plan(plan):
        if h2 is not None:
            check(lib.cape_gconv_dw_plan_h2(arr, len(entries), p, ss, ld, p2, mask, N, Mo, F, C.byref(h2), plan), "cape_gconv_dw_plan_h2")
        else:
            check(_fn("cape_gconv_dw_plan", dz)(arr, len(entries), p, ss, ld, p2, mask, N, Mo, F, plan), "cape_gconv_dw_plan")

    def dw_stage(which):
        if h2 is not None:
            check(lib.cape_gconv_dw_stage_h2(arr, len(entries), p, ss, ld, p2, mask, N, Mo, F, 1 if accumulate else 0,
                                             C.c_void_p(ws.data_ptr()), need, which, C.byref(h2), _stream()), "cape_gconv_dw_stage_h2")
        else:
            check(_fn("cape_gconv_dw_stage", dz)(arr, len(entries), p, ss, ld, p2, mask, N, Mo, F, 1 if accumulate else 0,
                                                 C.c_void_p(ws.data_ptr()), need, which, _stream()), "cape_gconv_dw_stage")

    def launch():
        dw_stage(0)

    if defer and DEFERRED is not None and LAUNCH_LOG is None:
        if PLAN_LOG is not None:
            plan = (C.c_int32 * 4)()
            dw_plan(plan)
            PLAN_LOG.add(("dw", plan[0], plan[1], plan[2]) + (("bf16",) if bf else ()))
        dw_stage(1)
        it = _lib.CapeDwItem()
        it.srcs, it.nsrc = C.addressof(arr), len(entries)
        it.dz, it.dz_sample_stride, it.lddz = p.value, ss, ld
        it.dz2, it.dz2_mask = (p2.value if p2 is not None else None), mask
        it.N, it.Mo, it.F, it.accumulate, it.bf16 = N, Mo, F, 1 if accumulate else 0, 1 if bf else 0
        it.workspace, it.workspace_bytes = ws.data_ptr(), need
        it.h2 = C.addressof(h2) if h2 is not None else None        # (the reduction counts the slabs the two-piece kernel wrote)
        DEFERRED_DW.append((it, arr, ws, dz, dz2, [e["x"] for e in entries], h2))   # keep every buffer alive until the flush
        return
    if LAUNCH_LOG is None and PLAN_LOG is None:
        launch()
    else:
        plan = (C.c_int32 * 4)()
        dw_plan(plan)
        fam, ct, ft, nslab = list(plan)
        if PLAN_LOG is not None:
            PLAN_LOG.add(("dw", fam, ct, ft) + (("bf16",) if bf else ()))
        if LAUNCH_LOG is None:
            launch()
            return
        flops, byts = _dw_work(entries, N, Mo, F, bool(mask))
        sumCF = sum(int(e.get("C", e["x"].shape[2])) for e in entries) * F
        # the contraction kernel and its fixed-order slab reduction, each with its own bracket
        _log_launch(dw_kernel_name(fam, ct, ft, bf), flops, byts, lambda: dw_stage(1))
        _log_launch("dw_reduce", 0, 4 * sumCF * (nslab + 1), lambda: dw_stage(2))


def spmm(x, csr, y=None, alpha=1.0, z=None, beta=0.0):
    _lib.require_gpu()
    N, Mi, Cn = x.shape
    Mo = csr.shape[0]
    if y is None:
        y = alloc_act(N, Mo, Cn, x.device, dtype=x.dtype)
    assert y.dtype == x.dtype and (z is None or z.dtype == x.dtype)
    es = x.element_size()
    xp, xs, xl = _v(x)
    yp, ys, yl = _v(y)
    if z is not None:
        zp, zs, zl = _v(z)
    else:
        zp, zs, zl = None, 0, 0
    rp_, ci_, va_, ew_ = csr.operands() if _vec_ok(x, y, z) else \
        (csr.rowptr_t.data_ptr(), csr.colidx_t.data_ptr(), csr.vals_t.data_ptr(), 0)          # scalar fallback: CSR only

    drop_rm(y)                           # (y may be a caller's tensor written in place)
    rm = _new_rm(y) if _want_rm(y) else None

    def launch():
        rc = _fn("cape_spmm", x)(xp, xs, xl, C.c_void_p(rp_), C.c_void_p(ci_), C.c_void_p(va_), int(csr.max_row), ew_,
                           float(alpha), zp, zs, zl, float(beta), yp, ys, yl, N, Mo, Cn, _ptr(rm), _stream())
        check(rc, "cape_spmm")

    _log_launch("spmm_kernel", 2 * N * csr.nnz * Cn, es * N * Cn * (Mi + Mo * (2 if z is not None else 1)) + 8 * csr.nnz + 4 * (Mo + 1),
                launch)
    return set_rm(y, rm)


def spmm_multi(xs, csrs, sum=False, scales=None, act_x=None, act=None):
    """Several operator applications in one launch: ``sum=False`` -> [s_k S_k x_k for k]; ``sum=True`` -> sum_k s_k S_k x_k
    (``scales`` default 1).  ``csrs[k]`` None or an identity DeviceCSR = identity term.  All operators have the same
    number of rows.  ``act_x`` (sum mode, fp32): the fused activation-gradient form (cape_spmm_multi_actgrad): returns
    (y * act'(act_x), bias partials, chunks) or None when the arguments do not allow it."""
    _lib.require_gpu()
    n = len(xs)
    assert 1 <= n <= 4 and len(csrs) == n
    N, _, Cn = xs[0].shape
    ident = [c is None or c.identity for c in csrs]
    Mo = next((c.shape[0] for c, i in zip(csrs, ident) if not i), xs[0].shape[1])
    arr = (_lib.CapeSpmmTerm * n)()
    outs = []
    vec_ok = _vec_ok(*xs)          # (fresh outputs are row-padded: only the inputs decide; the ELL form needs the vector kernels)
    for k in range(n):
        assert xs[k].shape[0] == N and xs[k].shape[2] == Cn and (not ident[k] or xs[k].shape[1] == Mo)
        t = arr[k]
        xp, t.x_sample_stride, t.ldx = _v(xs[k])
        t.x = xp.value
        t.scale = 1.0 if scales is None else float(scales[k])
        if ident[k]:
            t.rowptr = t.colidx = t.vals = None
        else:
            assert csrs[k].shape[0] == Mo and csrs[k].shape[1] == xs[k].shape[1]
            t.rowptr, t.colidx, t.vals, t.ell_width = csrs[k].operands() if vec_ok else \
                (csrs[k].rowptr_t.data_ptr(), csrs[k].colidx_t.data_ptr(), csrs[k].vals_t.data_ptr(), 0)
        assert xs[k].dtype == xs[0].dtype
        if not sum:
            yk = alloc_act(N, Mo, Cn, xs[0].device, dtype=xs[0].dtype)
            yp, t.y_sample_stride, t.ldy = _v(yk)
            t.y = yp.value
            if _want_rm(yk):
                rmk = _new_rm(yk)
                t.rowmax_out = rmk.data_ptr()
                set_rm(yk, rmk)
            outs.append(yk)
    rm = None
    if sum:
        y = alloc_act(N, Mo, Cn, xs[0].device, dtype=xs[0].dtype)
        yp, ys, yl = _v(y)
        if _want_rm(y):
            rm = _new_rm(y)
            set_rm(y, rm)
    else:
        y, yp, ys, yl = None, None, 0, 0
    es = xs[0].element_size()
    flops, byts = 0, es * N * Mo * Cn * (1 if sum else n)       # (``sum`` is this function's flag, not the builtin)
    for k in range(n):
        flops += 2 * N * (Mo if ident[k] else csrs[k].nnz) * Cn
        byts += es * N * Cn * xs[k].shape[1] + (0 if ident[k] else _csr_bytes(csrs[k]))
    if act_x is not None:
        assert sum and y.dtype == act_x.dtype and act in ("leaky", "relu") and act_x.shape == y.shape
        ap, as_, al = _v(act_x)
        chunks = int(_fn("cape_spmm_multi_actgrad_chunks", y)(yp, ys, yl, ap, as_, al, Mo, Cn))
        part = torch.empty((N, chunks, 2, Cn), device=y.device, dtype=torch.float32)
        _log_launch("spmm_multi_kernel", flops, byts + es * N * Mo * Cn,
                    lambda: check(_fn("cape_spmm_multi_actgrad", y)(arr, n, yp, ys, yl, N, Mo, Cn, _ptr(rm), ap, as_, al, _lib.ACT[act],
                                                                    _ptr(part), _stream()), "cape_spmm_multi_actgrad"))
        return y, part, chunks
    _log_launch("spmm_multi_kernel", flops, byts,
                lambda: check(_fn("cape_spmm_multi", xs[0])(arr, n, 1 if sum else 0, yp, ys, yl, N, Mo, Cn, _ptr(rm), _stream()), "cape_spmm_multi"))
    return y if sum else outs


def actgrad_fusable(xs, act_x, Cn):
    """The fused activation-gradient form of spmm_multi needs the 8-wide vector kernel with the lanes of a row forming one
    power-of-two group of at most 64: 64..512 channels in powers of two, every operand aligned for 8-element accesses (fresh
    outputs are).  fp32 or bf16 storage (all operands the same)."""
    if not (FUSE_ACT_GRAD and Cn in (64, 128, 256, 512)) or _os.environ.get("CAPE_SPMM_WIDE", "1") == "0":
        return False
    for t in list(xs) + [act_x]:
        p, ss, ld = _v(t)
        if t.dtype != act_x.dtype or t.dtype not in ACT_DTYPES or (p.value % (32 if t.dtype == torch.float32 else 16)) or (ss % 8) or (ld % 8):
            return False
    return True


def spmm_combine(xs, csrs, y, to_acc2=0, rank=None, bias=None, bias_mode=_lib.BIAS_NONE, act="none", dual=False, mask=None):
    """y = epilogue(sum_k S_k x_k [+ rank-1 terms]): the operators applied after the dense contraction (cape_spmm_combine)."""
    _lib.require_gpu()
    n = len(xs)
    N, Mo, F = y.shape
    arr = (_lib.CapeSpmmTerm * n)()
    vec_ok = _vec_ok(y, *xs)
    for k in range(n):
        t = arr[k]
        xp, t.x_sample_stride, t.ldx = _v(xs[k])
        t.x = xp.value
        assert xs[k].shape[2] == F and xs[k].dtype == y.dtype
        t.scale = 1.0
        if csrs[k] is None or csrs[k].identity:
            t.rowptr = t.colidx = t.vals = None
        else:
            assert csrs[k].shape[0] == Mo and csrs[k].shape[1] == xs[k].shape[1]
            t.rowptr, t.colidx, t.vals, t.ell_width = csrs[k].operands() if vec_ok else \
                (csrs[k].rowptr_t.data_ptr(), csrs[k].colidx_t.data_ptr(), csrs[k].vals_t.data_ptr(), 0)
        t.y, t.y_sample_stride, t.ldy = None, 0, 0
    rk = None
    if rank is not None:
        rowscale, coef, to2 = rank
        assert coef.is_contiguous() and rowscale.is_contiguous() and coef.shape[0] == N and coef.shape[2] == F
        rk = _lib.CapeRank(int(coef.shape[1]), rowscale.data_ptr(), coef.data_ptr(), int(to2))
    yp, ys, yl = _v(y)
    drop_rm(y)
    rm = _new_rm(y) if _want_rm(y) else None
    flops = sum(2 * N * (Mo if (c is None or c.identity) else c.nnz) * F for c in csrs)
    es = y.element_size()
    byts = sum(es * N * F * xs[k].shape[1] + _csr_bytes(csrs[k]) for k in range(n)) + es * N * Mo * F
    _log_launch("spmm_combine_kernel", flops, byts,
                lambda: check(_fn("cape_spmm_combine", y)(arr, n, int(to_acc2), C.byref(rk) if rk is not None else None, _ptr(bias),
                                                    bias_mode if bias is not None else _lib.BIAS_NONE, _lib.ACT[act],
                                                    1 if dual else 0, _ptr(mask), yp, ys, yl, N, Mo, F, _ptr(rm), _stream()),
                              "cape_spmm_combine"))
    return set_rm(y, rm)


def bias_act_fwd(x, bias, bias_mode, act, y=None):
    _lib.require_gpu()
    N, M, Cn = x.shape
    if y is None:
        y = alloc_act(N, M, Cn, x.device)
    xp, xs, xl = _v(x)
    yp, ys, yl = _v(y)
    rc = lib.cape_bias_act_fwd(xp, xs, xl, _ptr(bias), bias_mode if bias is not None else 0, _lib.ACT[act],
                               yp, ys, yl, N, M, Cn, _stream())
    check(rc, "cape_bias_act_fwd")
    return y


def act_bwd(dy, y, act, dz=None):
    _lib.require_gpu()
    N, M, Cn = dy.shape
    if dz is None:
        dz = alloc_act(N, M, Cn, dy.device)
    gp, gs, gl = _v(dy)
    yp, ys, yl = _v(y)
    zp, zs, zl = _v(dz)
    rc = lib.cape_act_bwd(gp, gs, gl, yp, ys, yl, _lib.ACT[act], zp, zs, zl, N, M, Cn, _stream())
    check(rc, "cape_act_bwd")
    return dz


def colsum(x, out, per_vertex=False, accumulate=False):
    _lib.require_gpu()
    N, M, Cn = x.shape
    xp, xs, xl = _v(x)
    if x.dtype == torch.bfloat16:
        if not per_vertex:
            raise NotImplementedError("channel column sums of a bf16 tensor come from cape_bwd_prep_bf16")
        check(lib.cape_colsum_vertex_bf16(xp, xs, xl, N, M, Cn, 1 if accumulate else 0, C.c_void_p(out.data_ptr()), _stream()),
              "cape_colsum_vertex_bf16")
        return out
    if per_vertex:
        ws, need = None, 0
    else:
        need = lib.cape_colsum_workspace_bytes(N, M, Cn)
        ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
    rc = lib.cape_colsum(xp, xs, xl, N, M, Cn, 1 if per_vertex else 0, 1 if accumulate else 0,
                         C.c_void_p(out.data_ptr()), _ptr(ws), need, _stream())
    check(rc, "cape_colsum")
    return out


def mask_mul(dy, mask, dz=None):
    _lib.require_gpu()
    N, M, F = dy.shape
    if dz is None:
        dz = alloc_act(N, M, F, dy.device)
    gp, gs, gl = _v(dy)
    zp, zs, zl = _v(dz)
    rc = lib.cape_mask_mul(gp, gs, gl, C.c_void_p(mask.data_ptr()), zp, zs, zl, N, M, F, _stream())
    check(rc, "cape_mask_mul")
    return dz


def fill_cond(cond, y, scale=None):
    _lib.require_gpu()
    N, M, Cn = y.shape
    assert cond.shape == (N, Cn) and cond.stride(1) == 1
    yp, ys, yl = _v(y)
    rc = lib.cape_fill_cond(C.c_void_p(cond.data_ptr()), cond.stride(0) if N > 1 else Cn, _ptr(scale), yp, ys, yl,
                            N, M, Cn, _stream())
    check(rc, "cape_fill_cond")
    return y


def reduce_cond(dy, scale=None, out=None, accumulate=False):
    _lib.require_gpu()
    N, M, Cn = dy.shape
    if out is None:
        out = torch.empty((N, Cn), device=dy.device, dtype=torch.float32)
    gp, gs, gl = _v(dy)
    rc = lib.cape_reduce_cond(gp, gs, gl, _ptr(scale), C.c_void_p(out.data_ptr()), out.stride(0) if N > 1 else Cn,
                              N, M, Cn, 1 if accumulate else 0, _stream())
    check(rc, "cape_reduce_cond")
    return out


# Weights (by data pointer) whose layers are differentiated for their DATA gradient only in the current sweep.
# ``needs_input_grad`` is static (True for every variable), so the generator sweep of the adversarial step -- which passes
# through D(fake) only to reach the generator -- would compute every discriminator weight gradient and discard it; the
# training step names the discriminator's variables here around that sweep (cape_amd.models).
NO_WEIGHT_GRAD = None


# Deferred finalisation of bwd_prep's reductions: while DEFERRED is a list (set by the training step around the backward
# pass) calls with defer=True leave their bias / coefficient-gradient reductions as partial slabs and queue them;
# flush_deferred() finishes all queued ones in ONE launch (csrc cape_bwd_prep_finalize).  Consumers flush before reading.
DEFERRED = None
DEFERRED_DW = []          # queued weight-gradient slab reductions (gconv_dw(defer=True)), same lifetime as DEFERRED
DEFERRED_GN = []          # queued batch sums of group-norm parameter-gradient partials (GroupNormFn.backward), likewise

def _flush_dw():
    """The queued fixed-order slab reductions of the weight gradient, as batched launches."""
    if DEFERRED_DW:
        queued, DEFERRED_DW[:] = list(DEFERRED_DW), []
        nmax = 12                                    # CAPE_MAX_DW_REDUCE_ITEMS
        for i0 in range(0, len(queued), nmax):
            chunk = queued[i0:i0 + nmax]
            arr = (_lib.CapeDwItem * len(chunk))(*[q[0] for q in chunk])
            check(lib.cape_gconv_dw_reduce_batch(C.addressof(arr), len(chunk), _stream()), "cape_gconv_dw_reduce_batch")


def flush_deferred():
    global DEFERRED
    if DEFERRED_GN:
        queued, DEFERRED_GN[:] = list(DEFERRED_GN), []
        nmax = 32                                    # CAPE_MAX_GN_REDUCE_ITEMS
        for i0 in range(0, len(queued), nmax):
            chunk = queued[i0:i0 + nmax]
            arr = (_lib.CapeGnParamItem * len(chunk))()
            for a, (dgb, dst_g, dst_b) in zip(arr, chunk):
                a.dgamma_partial, a.dbeta_partial = dgb[0].data_ptr(), dgb[1].data_ptr()
                a.dgamma, a.dbeta = dst_g.data_ptr(), dst_b.data_ptr()
                a.N, a.C = int(dgb.shape[1]), int(dgb.shape[2])
            check(lib.cape_groupnorm_param_reduce_batch(C.addressof(arr), len(chunk), _stream()), "cape_groupnorm_param_reduce_batch")
    _flush_dw()
    if not DEFERRED:
        return
    items, DEFERRED[:] = list(DEFERRED), []
    _finalize_bwd_prep(items)


def _finalize_bwd_prep(items):
    """The final reductions of queued backward-prep partials (cape_bwd_prep_finalize, up to 16 layers per launch); ``chunks``:
    partials written by another producer (spmm_multi's fused activation-gradient form)."""
    for i0 in range(0, len(items), 16):
        chunk = items[i0:i0 + 16]
        arr = (_lib.CapeBwdPrepItem * len(chunk))()
        for a, it in zip(arr, chunk):
            a.workspace = it["ws"].data_ptr()
            a.N, a.Mo, a.F, a.R = it["N"], it["Mo"], it["F"], it["R"]
            a.dbias = None if it["dbias"] is None else it["dbias"].data_ptr()
            a.dcoef = None if it["dcoef"] is None else it["dcoef"].data_ptr()
            a.dcoef_g = None if it["dcoef_g"] is None else it["dcoef_g"].data_ptr()
            a.dcoef_sample_stride = it["cstride"]
            a.chunks = int(it.get("chunks", 0))
        check(lib.cape_bwd_prep_finalize(arr, len(chunk), _stream()), "cape_bwd_prep_finalize")


def bwd_prep(g, y=None, act="none", mask=None, want_bias=False, rowscale=None, R=0, rg=None, dbias_out=None, joint=False,
             defer=False):
    """One pass over g: returns (dz, dbias [F] or None, dcoef [N,R,F] or None, dcoef_g [N,F] or None).
    ``dbias_out``: optional contiguous destination of the bias gradient (a view of the gradient bucket).
    ``joint``: dcoef and dcoef_g are slices of ONE [N, R+1, F] buffer (returned as dcoef; the layout
    cape_cond_coef_bwd consumes)."""
    _lib.require_gpu()
    N, Mo, F = g.shape
    dev = g.device
    # no activation and no mask: dz IS g -- the pass is made for its sums alone and writes nothing (dz aliases g; the kernels skip
    # the store when the two pointers are equal)
    alias = act == "none" and mask is None and _vec_ok(g)     # (an unaligned gradient still gets its row-padded copy: the
    #                                                            weight-gradient kernels choose their staging by dz's alignment)
    dz = g if alias else alloc_act(N, Mo, F, dev, dtype=g.dtype)
    assert y is None or y.dtype == g.dtype
    dbias = None
    if want_bias:
        if dbias_out is not None and dbias_out.numel() == F and dbias_out.is_contiguous():
            dbias = dbias_out.view(F)
        else:
            dbias = torch.empty(F, device=dev, dtype=torch.float32)
    cstride = 0
    if joint and R and rg is not None:
        dcoef = torch.empty((N, R + 1, F), device=dev, dtype=torch.float32)
        dcoef_g = dcoef[:, R]
        cstride = (R + 1) * F
    else:
        dcoef = torch.empty((N, R, F), device=dev, dtype=torch.float32) if R else None
        dcoef_g = torch.empty((N, F), device=dev, dtype=torch.float32) if rg is not None else None
    need = lib.cape_bwd_prep_workspace_bytes(N, Mo, F, R)
    ws = torch.empty((need + 3) // 4, device=dev, dtype=torch.float32)
    gp, gs, gl = _v(g)
    zp, zs, zl = _v(dz)
    if y is not None and mask is None and act != "none":
        yp, ys, yl = _v(y)
    else:
        yp, ys, yl = None, 0, 0
    # |dz| <= |g| element by element (|act'| <= 1, or the 0 / 1 mask): the bound of g's rows bounds dz's -- no reduction needed
    # when g carries one (the data-gradient contraction / summed operator application that produced it wrote it)
    rm, rm_g = None, rm_of(g)
    if rm_g is not None:
        set_rm(dz, rm_g)
    elif _want_rm(dz):
        rm = _new_rm(dz)                 # the kernel bounds the rows of g: valid for g and for dz
        set_rm(dz, rm)
        set_rm(g, rm)

    def launch():
        rc = _fn("cape_bwd_prep", g)(gp, gs, gl, yp, ys, yl, _lib.ACT[act] if mask is None else 0, _ptr(mask), zp, zs, zl,
                               _ptr(dbias), _ptr(rowscale), R, _ptr(dcoef), 0 if rg is None else int(rg), _ptr(dcoef_g),
                               cstride, 0 if (defer and DEFERRED is not None) else 1, N, Mo, F, _ptr(ws), need, _ptr(rm), _stream())
        check(rc, "cape_bwd_prep")

    # one pass: read g (+ y or the 1-bit mask), write dz
    _log_launch("bwd_prep", 0, g.element_size() * N * Mo * F * (1 if alias else 3 if yp is not None else 2) + (N * Mo * ((F + 31) // 32) * 4 if mask is not None else 0),
                launch)
    if defer and DEFERRED is not None and (dbias is not None or R or rg is not None):
        DEFERRED.append(dict(ws=ws, N=N, Mo=Mo, F=F, R=R, dbias=dbias, dcoef=dcoef, dcoef_g=dcoef_g, cstride=cstride))
    return dz, dbias, dcoef, dcoef_g


def bwd_prep_spmm(g, mask, csr, rowscale=None, R=0, rg=None, joint=False, defer=False):
    """``bwd_prep(g, mask=mask, ...)`` and ``spmm(dz, csr)`` of an affine block at one resolution in ONE launch
    (cape_bwd_prep_spmm): returns (dz, T1, dcoef, dcoef_g) -- T1 = csr @ dz bit-identical to the two-launch form -- or None when
    the arguments do not allow the fused form (the caller then takes the two launches)."""
    _lib.require_gpu()
    N, Mo, F = g.shape
    if not (FUSE_PREP_SPMM and g.dtype in ACT_DTYPES and mask is not None and F % 32 == 0 and R <= 2 and csr.shape[0] == Mo and csr.shape[1] == Mo):
        return None
    dev = g.device
    dz = alloc_act(N, Mo, F, dev, dtype=g.dtype)
    t1 = alloc_act(N, Mo, F, dev, dtype=g.dtype)
    gp, gs, gl = _v(g)
    zp, zs, zl = _v(dz)
    tp, ts, tl = _v(t1)
    chunks = int(_fn("cape_bwd_prep_spmm_chunks", g)(gp, gs, gl, zp, zs, zl, tp, ts, tl, N, Mo, F))
    if chunks <= 0:
        return None
    cstride = 0
    if joint and R and rg is not None:
        dcoef = torch.empty((N, R + 1, F), device=dev, dtype=torch.float32)
        dcoef_g = dcoef[:, R]
        cstride = (R + 1) * F
    else:
        dcoef = torch.empty((N, R, F), device=dev, dtype=torch.float32) if R else None
        dcoef_g = torch.empty((N, F), device=dev, dtype=torch.float32) if rg is not None else None
    need_part = bool(R) or rg is not None
    part = torch.empty((N, chunks, R + 2, F), device=dev, dtype=torch.float32) if need_part else None
    rp_, ci_, va_, ew_ = csr.operands()
    # row bounds: g's own (from its producer) serve dz as well (|dz| <= |g|); else the kernel bounds g.  T1's come with it.
    rm_g_new, rm_g = None, rm_of(g)
    if rm_g is not None:
        set_rm(dz, rm_g)
    elif _want_rm(dz):
        rm_g_new = _new_rm(dz)
        set_rm(dz, rm_g_new)
        set_rm(g, rm_g_new)
    rm_t1 = _new_rm(t1) if _want_rm(t1) else None

    def launch():
        rc = _fn("cape_bwd_prep_spmm", g)(gp, gs, gl, _ptr(mask), C.c_void_p(rp_), C.c_void_p(ci_), C.c_void_p(va_), ew_, zp, zs, zl, tp, ts, tl,
                                    _ptr(rowscale), R, -1 if rg is None else int(rg), N, Mo, F, _ptr(part),
                                    0 if part is None else part.numel() * 4, _ptr(rm_g_new), _ptr(rm_t1), _stream())
        check(rc, "cape_bwd_prep_spmm")

    # one pass: read g (+ sign words, + the gathered neighbour rows: cache hits), write dz and T1
    _log_launch("bwd_prep_spmm", 2 * N * csr.nnz * F, g.element_size() * N * Mo * F * 3 + N * Mo * (F // 32) * 4 + 8 * csr.nnz, launch)
    set_rm(t1, rm_t1)
    if need_part:
        item = dict(ws=part, N=N, Mo=Mo, F=F, R=R, dbias=None, dcoef=dcoef, dcoef_g=dcoef_g, cstride=cstride, chunks=chunks)
        if defer and DEFERRED is not None:
            DEFERRED.append(item)
        else:
            _finalize_bwd_prep([item])
    return dz, t1, dcoef, dcoef_g


def spmm_multi_prep(g, mask, csrs, masked, want_sums=True, joint=False, defer=False):
    """All operator applications of an up-sampling affine block's data gradient in ONE launch (cape_spmm_multi_prep):
    ``T_k = csrs[k] @ (g * sign bits)`` for the terms flagged in ``masked``, ``csrs[k] @ g`` for the others, and -- ``want_sums`` --
    the column sums of every T_k as the rank-1 condition gradients (dcoef [N, n-1, F] from the first n-1 terms, dcoef_g [N, F]
    from the last; ``joint``: slices of one [N, n, F] buffer).  ``dz = g * sign bits`` is never written.  Returns
    (Ts, dcoef, dcoef_g) or None when the arguments do not allow the fused form."""
    _lib.require_gpu()
    N, Mf, F = g.shape
    n = len(csrs)
    if not (FUSE_PREP_SPMM and g.dtype in ACT_DTYPES and mask is not None and F % 32 == 0 and 2 <= n <= 3
            and all(c is not None and not c.identity and c.shape[1] == Mf for c in csrs)):
        return None
    Mo = csrs[0].shape[0]
    dev = g.device
    arr = (_lib.CapeSpmmTerm * n)()
    outs = []
    gp, gs, gl = _v(g)
    bits = 0
    for k in range(n):
        assert csrs[k].shape[0] == Mo
        t = arr[k]
        t.x, t.x_sample_stride, t.ldx = gp.value, gs, gl
        t.scale = 1.0
        t.rowptr, t.colidx, t.vals, t.ell_width = csrs[k].operands()
        yk = alloc_act(N, Mo, F, dev, dtype=g.dtype)
        yp, t.y_sample_stride, t.ldy = _v(yk)
        t.y = yp.value
        if _want_rm(yk):
            rmk = _new_rm(yk)
            t.rowmax_out = rmk.data_ptr()
            set_rm(yk, rmk)
        outs.append(yk)
        bits |= (1 << k) if masked[k] else 0
    chunks = int(_fn("cape_spmm_multi_prep_chunks", g)(arr, n, N, Mo, F))
    if chunks <= 0:
        return None
    R = n - 1
    dcoef = dcoef_g = part = None
    cstride = 0
    if want_sums:
        if joint:
            dcoef = torch.empty((N, R + 1, F), device=dev, dtype=torch.float32)
            dcoef_g = dcoef[:, R]
            cstride = (R + 1) * F
        else:
            dcoef = torch.empty((N, R, F), device=dev, dtype=torch.float32)
            dcoef_g = torch.empty((N, F), device=dev, dtype=torch.float32)
        part = torch.empty((N, chunks, n + 1, F), device=dev, dtype=torch.float32)
    flops = sum(2 * N * c.nnz * F for c in csrs)
    byts = g.element_size() * N * F * (Mf + n * Mo) + N * Mf * (F // 32) * 4 + sum(_csr_bytes(c) for c in csrs)
    _log_launch("spmm_multi_prep", flops, byts,
                lambda: check(_fn("cape_spmm_multi_prep", g)(arr, n, bits, _ptr(mask), Mf, N, Mo, F, _ptr(part),
                                                       0 if part is None else part.numel() * 4, _stream()), "cape_spmm_multi_prep"))
    if want_sums:
        item = dict(ws=part, N=N, Mo=Mo, F=F, R=R, dbias=None, dcoef=dcoef, dcoef_g=dcoef_g, cstride=cstride, chunks=chunks)
        if defer and DEFERRED is not None:
            DEFERRED.append(item)
        else:
            _finalize_bwd_prep([item])
    return outs, dcoef, dcoef_g


def rowscale_reduce(dz, rowscale, R):
    """out[n, j, f] = sum_r rowscale[j, r] * dz[n, r, f]  for j < R."""
    _lib.require_gpu()
    N, Mo, F = dz.shape
    out = torch.empty((N, R, F), device=dz.device, dtype=torch.float32)
    need = lib.cape_rowscale_reduce_workspace_bytes(N, Mo, F, R)
    ws = torch.empty((need + 3) // 4, device=dz.device, dtype=torch.float32)
    p, ss, ld = _v(dz)
    rc = lib.cape_rowscale_reduce(p, ss, ld, C.c_void_p(rowscale.data_ptr()), R, N, Mo, F, _ptr(out), _ptr(ws), need,
                                  _stream())
    check(rc, "cape_rowscale_reduce")
    return out


# --------------------------------------------------------------------------------------------
# autograd operators
def _grad_buffer(W, view=None):
    """Destination of a weight gradient: the caller-provided view of the flat gradient bucket (the
    kernels then write straight into the bucket and the per-variable copy disappears) or a fresh tensor."""
    if view is not None and view.shape == W.shape and view.is_contiguous():
        return view
    return torch.empty_like(W)


class ChebConvFn(torch.autograd.Function):
    """y = [ epilogue( sum_k (S_k [x | cond_in 1^T]) W_k ) | cond_out tiled over vertices ]

    * plain mode  (W_aff None): epilogue = act(. + bias)            -- chebyshev5 + b1*/b2relu
      (+ poolwT folded into S_k), reference lib/models.py:69-127,154-171,796-810
    * affine mode (W_aff given): relu(sum_k (S_k x) W_k) + (S_0 x) W_aff -- res_block_affine,
      lib/models.py:776-793 (unpool folded into S_k)
    ``cond_in`` [N, Cc] stands for Cc vertex-constant INPUT channels appended after x's channels
    (fit_cond_dim + tf.concat, :591-594, :606-609, :663-666): they are never materialised -- their
    contribution is the rank-1 update (S_k 1)(cond_in W_k[cond rows]) added in the GEMM epilogue.
    ``cond_out`` [N, Cc'] is appended to the OUTPUT as materialised channels (only where a consumer
    needs the concatenated tensor, e.g. group-norm blocks).
    ``mode``: "twopass" (the only form: X_k = S_k x by the streaming spmm kernels, then a plain multi-source GEMM).
    """

    @staticmethod
    def forward(ctx, x, W, bias, W_aff, cond_in, cond_out, ops, act, bias_mode, mode, gW=None, gWa=None, gB=None,
                coef=None):
        x = as_act(x)
        if (x.shape[2] & 3) and (x.stride(1) & 3):
            # e.g. the [N, 6890, 3] network input: re-home it in a row-padded buffer so that the kernels can use
            # aligned float4 accesses (one small copy instead of scalar staging in three GEMM launches)
            xp = alloc_act(x.shape[0], x.shape[1], x.shape[2], x.device, zero=True, dtype=x.dtype)
            xp.copy_(x)
            x = xp
        N, Mi, Ch = x.shape
        K, Fout = ops.K, W.shape[1]
        # ``coef`` [N, K (+1), Fout]: the rank-1 coefficients of the tiled condition channels, precomputed for
        # all layers at once by CondCoefFn (then cond_in is None and the weight rows beyond Ch*K belong to it)
        assert coef is None or cond_in is None
        banked = coef is not None
        Cc = (W.shape[0] // K - Ch) if coef is not None else (0 if cond_in is None else cond_in.shape[1])
        assert ops.fused and W.shape[0] == (Ch + Cc) * K and Mi == ops.Mi
        assert W.is_contiguous() and (W_aff is None or (W_aff.is_contiguous() and W_aff.shape == (Ch + Cc, Fout)))
        Co = 0 if cond_out is None else cond_out.shape[1]
        yfull = alloc_act(N, ops.Mo, Fout + Co, x.device, dtype=x.dtype)
        y = yfull[:, :, :Fout]
        assert mode == "twopass"
        twopass = True
        # up-sampling layer in two-pass mode: contract on the coarse rows, apply the operators to the products
        # (only when backward will not ask for the fine-level X_k: inference, or the coarse weight-gradient form)
        any_grad = any(ctx.needs_input_grad[i] for i in (0, 1, 2, 3, 4))
        coarse = bool(twopass and ops.Mo > Mi and not any(ops.fwd[k].identity for k in range(K))
                      and (W_aff is None or Fout % 32 == 0) and Fout % 4 == 0 and (ctx.needs_input_grad[0] or not any_grad))
        if coarse:
            return ChebConvFn._forward_coarse(ctx, x, W, bias, W_aff, cond_in, cond_out, ops, act, bias_mode, gW, gWa, gB, coef,
                                              banked, Cc, yfull, y)
        xs = [x] * K
        if twopass:
            ks = [k for k in range(K) if not ops.fwd[k].identity]
            if len(ks) == 1:
                xs[ks[0]] = spmm(x, ops.fwd[ks[0]])
            elif ks:                                   # X_k = S_k x of all orders in one launch
                for k, xk in zip(ks, spmm_multi([x] * len(ks), [ops.fwd[k] for k in ks])):
                    xs[k] = xk
        P, Pa = ChebConvFn._pieces(x, W, W_aff, Ch, K, Fout, twopass)
        fw_ok = P is not None and Ch % 32 == 0 and h2_shape_ok(Ch * K, Fout)      # (forward planes: whole 32-channel chunks)
        entries = []
        for k in range(K):
            e = dict(x=xs[k], csr=None, w=(W, k * Fout, K * Fout, 1))
            if W_aff is not None and k == 0:
                e["w2"] = (W_aff, 0, Fout, 1)
            if fw_ok:
                e["p"], e["rm"] = P.fwd(k), rowmax(xs[k])
                if "w2" in e:
                    e["p2"] = Pa.fwd(0)
            entries.append(e)
        h2kw = dict(wsi=_ptr(P.fsi), wsi2=_ptr(Pa.fsi) if Pa is not None else None) if fw_ok else {}
        # the epilogue writes the row bounds of y next to it (consumed by the next contraction; csrc/gconv_shared.h)
        # (only for outputs wide enough that their consumers take the two-piece kernels: on the short 64-column launches the
        # bound reduction costs 10-20 % of the kernel -- profiles/r04_h2_bench_*.txt -- and a consumer that does qualify falls
        # back to one standalone pass)
        rm_y = alloc_rm(y) if (H2 and y.dtype == torch.float32 and Fout >= 128) else None
        rank = None
        if coef is not None:
            assert coef.is_contiguous() and coef.shape == (N, K + (1 if W_aff is not None else 0), Fout)
            rank = ((ops.rowscale if W_aff is not None else ops.rowscale[:K]).contiguous(), coef,
                    (1 << K) if W_aff is not None else 0)
        elif Cc:
            cond_in = cond_in.contiguous()
            coef = torch.mm(cond_in, W[Ch * K:].view(Cc, K * Fout)).view(N, K, Fout)
            to2 = 0
            if W_aff is not None:
                coef = torch.cat([coef, torch.mm(cond_in, W_aff[Ch:]).view(N, 1, Fout)], dim=1)
                to2 = 1 << K
                rowscale = ops.rowscale                      # [K+1, Mo], last row = S_0 1
            else:
                rowscale = ops.rowscale[:K]
            rank = (rowscale.contiguous(), coef.contiguous(), to2)
        mask = None
        if W_aff is not None:
            mask = torch.empty((N, ops.Mo, (Fout + 31) // 32), device=x.device, dtype=torch.int32)
            gconv_fwd(entries, y, mask=mask, rank=rank, rm_out=rm_y, **h2kw)
            _trace_mask_bits(mask, Fout)
        else:
            gconv_fwd(entries, y, bias=bias, bias_mode=bias_mode, act=act, rank=rank, rm_out=rm_y, **h2kw)
            _trace_sign(y, act)
        if Co:
            fill_cond(cond_out.contiguous(), yfull[:, :, Fout:])
        elif rm_y is not None:
            set_rm(yfull, rm_y)
        ctx.ops, ctx.act, ctx.bias_mode, ctx.Fout, ctx.Co, ctx.Cc = ops, act, bias_mode, Fout, Co, Cc
        ctx.has_bias, ctx.twopass, ctx.xshape = bias is not None, twopass, (N, Mi, Ch)
        ctx.gW, ctx.gWa, ctx.gB, ctx.banked = gW, gWa, gB, banked
        ctx.pieces = (P, Pa)
        # sole-consumer chain (see sole_consumer_chain): x is the output of a layer whose bias + (leaky-)ReLU epilogue this
        # layer's backward may differentiate for it; our own output is tagged the same way for the layer above
        ctx.prev_act = getattr(x, "_cape_act_out", None) if (_CHAIN[0] and FUSE_ACT_GRAD and twopass) else None
        ctx.offers_dz = bool(W_aff is None and Co == 0 and Cc == 0 and bias is not None and bias_mode == _lib.BIAS_CHANNEL
                             and act in ("leaky", "relu") and y.dtype in ACT_DTYPES and gB is not None)
        ctx.offer = None
        if ctx.offers_dz:
            ctx.offer = yfull._cape_act_out = _ActOffer(act, gB)
        # up-sampling layers (Mo > Mi): the data gradient needs T_k = S_k^T dz at the Mi input rows anyway, and
        # dW_k = X_k^T dz = x^T T_k -- the weight gradient contracts over the COARSE rows (half the flops) and the
        # fine-level X_k need not be kept for the backward pass at all
        ctx.coarse_dw = bool(twopass and ops.Mo > Mi and ctx.needs_input_grad[0] and not any(ops.fwd[k].identity for k in range(K)))
        ctx.save_for_backward(W, W_aff, mask, yfull if (act != "none" and W_aff is None) else None, cond_in,
                              *(([x] if ctx.coarse_dw else list(xs)) + ([x] if ctx.prev_act is not None else [])))
        return yfull

    @staticmethod
    def _pieces(x, W, W_aff, Ch, K, Fout, twopass):
        """Piece planes of the layer's weights when its contractions may run on the fp16 two-piece kernel (fp32 storage,
        two-pass mode, whole 32-channel chunks on one side at least), else (None, None)."""
        if not (H2 and twopass and x.dtype == torch.float32 and Ch % 8 == 0 and Fout % 8 == 0 and
                ((Ch % 32 == 0 and Fout >= 64) or (Fout % 32 == 0 and Ch >= 64))):
            return None, None
        P = pieces_for(W, Ch, K, pair=(W_aff.detach(), 1) if W_aff is not None else None)
        Pa = pieces_for(W_aff, Ch, 1, pair=(W.detach(), K)) if (W_aff is not None and P is not None) else None
        if W_aff is not None and Pa is None:
            return None, None
        return P, Pa

    @staticmethod
    def _forward_coarse(ctx, x, W, bias, W_aff, cond_in, cond_out, ops, act, bias_mode, gW, gWa, gB, coef, banked, Cc, yfull, y):
        """(S_k x) W_k = S_k (x W_k): one GEMM on the Mi input rows for all K orders (the feature rows of W viewed
        as [Ch, K*Fout]), one for the affine weights, then cape_spmm_combine applies the operators, the rank-1 terms
        and the epilogue.  Backward is the ordinary one in its coarse weight-gradient form (needs only x)."""
        N, Mi, Ch = x.shape
        K, Fout = ops.K, W.shape[1]
        P, Pa = ChebConvFn._pieces(x, W, W_aff, Ch, K, Fout, True)
        fwd_ok = P is not None and Ch % 32 == 0 and h2_shape_ok(Ch, Fout)          # (K * Fout and Fout output columns)
        rmx = rowmax(x) if fwd_ok else None
        Z = alloc_act(N, Mi, K * Fout, x.device, dtype=x.dtype)
        if fwd_ok:
            gconv_fwd([dict(x=x, csr=None, w=(W, 0, K * Fout, 1), p=(P.f_hi.data_ptr(), P.f_lo.data_ptr(), Ch), rm=rmx)], Z,
                      wsi=_ptr(P.fsi))
        else:
            gconv_fwd([dict(x=x, csr=None, w=(W, 0, K * Fout, 1))], Z)
        zs = [Z[:, :, k * Fout:(k + 1) * Fout] for k in range(K)]
        csrs = [ops.fwd[k] for k in range(K)]
        to2 = 0
        if W_aff is not None:
            Za = alloc_act(N, Mi, Fout, x.device, dtype=x.dtype)
            if fwd_ok:
                gconv_fwd([dict(x=x, csr=None, w=(W_aff, 0, Fout, 1), p=Pa.fwd(0), rm=rmx)], Za, wsi=_ptr(Pa.fsi))
            else:
                gconv_fwd([dict(x=x, csr=None, w=(W_aff, 0, Fout, 1))], Za)
            zs.append(Za)
            csrs.append(ops.fwd[0])
            to2 = 1 << K
        rank = None
        if coef is not None:
            assert coef.is_contiguous() and coef.shape == (N, K + (1 if W_aff is not None else 0), Fout)
            rank = ((ops.rowscale if W_aff is not None else ops.rowscale[:K]).contiguous(), coef,
                    (1 << K) if W_aff is not None else 0)
        elif Cc:
            cond_in = cond_in.contiguous()
            cf = torch.mm(cond_in, W[Ch * K:].view(Cc, K * Fout)).view(N, K, Fout)
            if W_aff is not None:
                cf = torch.cat([cf, torch.mm(cond_in, W_aff[Ch:]).view(N, 1, Fout)], dim=1)
            rank = ((ops.rowscale if W_aff is not None else ops.rowscale[:K]).contiguous(), cf.contiguous(),
                    (1 << K) if W_aff is not None else 0)
        mask = None
        if W_aff is not None:
            mask = torch.empty((N, ops.Mo, (Fout + 31) // 32), device=x.device, dtype=torch.int32)
            spmm_combine(zs, csrs, y, to_acc2=to2, rank=rank, dual=True, mask=mask)
            _trace_mask_bits(mask, Fout)
        else:
            spmm_combine(zs, csrs, y, rank=rank, bias=bias, bias_mode=bias_mode, act=act)
            _trace_sign(y, act)
        Co = 0 if cond_out is None else cond_out.shape[1]
        if Co:
            fill_cond(cond_out.contiguous(), yfull[:, :, Fout:])
        else:
            set_rm(yfull, rm_of(y))         # (y is a view of yfull: the bounds spmm_combine wrote belong to the returned object too)
        ctx.ops, ctx.act, ctx.bias_mode, ctx.Fout, ctx.Co, ctx.Cc = ops, act, bias_mode, Fout, Co, Cc
        ctx.has_bias, ctx.twopass, ctx.xshape = bias is not None, True, (N, Mi, Ch)
        ctx.gW, ctx.gWa, ctx.gB, ctx.banked = gW, gWa, gB, banked
        ctx.coarse_dw = True
        ctx.pieces = (P, Pa)
        ctx.save_for_backward(W, W_aff, mask, yfull if (act != "none" and W_aff is None) else None, cond_in, x)
        return yfull

    @staticmethod
    def backward(ctx, gfull):
        W, W_aff, mask, ysaved, cond_in = ctx.saved_tensors[:5]
        xs = ctx.saved_tensors[5:]
        act_x = None
        if getattr(ctx, "prev_act", None) is not None:
            xs, act_x = xs[:-1], xs[-1]                 # (the layer input itself: the output of the layer below)
        ops, act, Fout, Co, Cc = ctx.ops, ctx.act, ctx.Fout, ctx.Co, ctx.Cc
        K, twopass = ops.K, ctx.twopass
        N, Mi, Ch = ctx.xshape
        Mo, dev = ops.Mo, W.device
        gfull = as_act(gfull)
        g = gfull[:, :, :Fout]
        set_rm(g, rm_of(gfull))            # a bound over a superset of the channels is still a bound
        need_x, need_w, need_b, need_wa, need_ci, need_co = (ctx.needs_input_grad[i] for i in range(6))
        if NO_WEIGHT_GRAD and W.data_ptr() in NO_WEIGHT_GRAD:
            need_w = need_b = need_wa = False       # data-gradient-only sweep through this layer (see NO_WEIGHT_GRAD)
        dW = dB = dWa = dci = dco = dx = dcoef_out = None
        T1_pre = None                      # T_1 = S_1^T dz when the backward-prep launch already produced it (bwd_prep_spmm)
        Ts_pre = None                      # all T_k of an up-sampling affine block (spmm_multi_prep): dz is then None
        # one pass over g: dz (activation / ReLU-mask gradient), channel-bias gradient and the rank-1
        # condition-term gradients
        chan_bias = need_b and ctx.has_bias and ctx.bias_mode != _lib.BIAS_VERTEX
        plain = (W_aff is None and act == "none" and not chan_bias and not Cc)
        tag = getattr(gfull, "_cape_is_dz", None)
        if tag is not None and not getattr(ctx, "offers_dz", False):
            raise RuntimeError("a pre-activated gradient reached a layer that did not offer it (sole_consumer_chain misuse)")
        offer = getattr(ctx, "offer", None)
        if tag is not None and (tag["offer"] is not offer or gfull._version != tag["version"]):
            raise RuntimeError("a pre-activated gradient was written after its producer tagged it, or belongs to another layer "
                               "(sole_consumer_chain: the tensor has a second consumer whose gradient the engine summed in)")
        if offer is not None:
            if offer.fused and tag is None:
                raise RuntimeError("the consumer of this layer's output differentiated its activation (sole_consumer_chain) but "
                                   "the gradient arrived without its tag: it would be multiplied by act' twice")
            offer.fused = False
        if plain:
            dz, dbv, dcoef, dca = g, None, None, None
        elif tag is not None:
            # the layer above already multiplied by act'(y) and left the bias sums with the deferred reductions (or in ``dbias``)
            dz, dbv, dcoef, dca = g, (tag["dbias"] if tag["dbias"] is not None else ctx.gB), None, None
        else:
            bp_kw = dict(rowscale=ops.rowscale if Cc else None, R=K if Cc else 0, rg=(K if (Cc and W_aff is not None) else None),
                         joint=ctx.banked,
                         # results read only at the end of the backward pass (bucket view / CondCoefFn)
                         defer=(not chan_bias or ctx.gB is not None) and (not Cc or ctx.banked))
            fused_t1 = None
            if (W_aff is not None and mask is not None and not chan_bias and need_x and K == 2 and Mo == Mi and Ch >= Fout and twopass
                    and ops.bwd[0].identity and not ops.bwd[1].identity and not ctx.coarse_dw):
                # affine block at one resolution: dz, T_1 = L~^T dz and the condition sums in one launch
                fused_t1 = bwd_prep_spmm(g, mask, ops.bwd[1], **bp_kw)
            fused_ts = None
            if (fused_t1 is None and W_aff is not None and mask is not None and not ctx.has_bias and need_x and K == 2 and ctx.coarse_dw
                    and twopass and not any(ops.bwd[k].identity for k in range(K))):
                # up-sampling affine block: T_0, T_1 (from dz = g * sign bits, formed while gathering) and T_aff (from g) in one
                # launch, the condition sums as their column sums; dz itself is needed by nobody (the weight gradient contracts
                # the T_k: coarse_dw)
                fused_ts = spmm_multi_prep(g, mask, [ops.bwd[0], ops.bwd[1], ops.bwd[0]], [True, True, False], want_sums=bool(Cc),
                                           joint=ctx.banked, defer=bp_kw["defer"])
            if fused_t1 is not None:
                dz, T1_pre, dcoef, dca = fused_t1
                dbv = None
            elif fused_ts is not None:
                Ts_pre, dcoef, dca = fused_ts
                dz, dbv = None, None
            else:
                dz, dbv, dcoef, dca = bwd_prep(g, y=None if ysaved is None else ysaved[:, :, :Fout], act=act, mask=mask,
                                               want_bias=chan_bias, dbias_out=ctx.gB, **bp_kw)
        if need_b and ctx.has_bias:
            if ctx.bias_mode == _lib.BIAS_VERTEX:
                gb = ctx.gB                      # the bucket view of the [1, M, F] bias: written in place, no copy later
                dB = gb.view(1, Mo, Fout) if (gb is not None and gb.numel() == Mo * Fout and gb.is_contiguous()) else \
                    torch.empty((1, Mo, Fout), device=dev, dtype=torch.float32)
                colsum(dz, dB, per_vertex=True)
            else:
                dB = dbv.view(1, 1, Fout)
        if dz is not None and H2 and twopass and dz.dtype == torch.float32 and rm_of(dz) is None and Fout % 32 == 0 and Fout >= 64 and (need_w or need_x):
            rowmax(dz)      # (a gradient without bounds, e.g. from a group norm: one pass now serves the weight AND the data gradient)
        csr_of = lambda k: None
        if need_w:
            dW = _grad_buffer(W, ctx.gW)
        if W_aff is not None and need_wa:
            dWa = _grad_buffer(W_aff, ctx.gWa)
        ent = []
        if ctx.coarse_dw:
            pass                                   # computed from the T_k below
        elif need_w:
            ent += [dict(x=xs[k], csr=csr_of(k), w=(dW, k * Fout, K * Fout, 1)) for k in range(K)]
        if W_aff is not None and need_wa and not ctx.coarse_dw:
            ent.append(dict(x=xs[0], csr=csr_of(0), w=(dWa, 0, Fout, 1), use_dz2=True))
        # the slab reductions may wait for the end of the backward pass when every gradient block is a bucket view
        in_bucket = lambda t, v: t is None or (v is not None and t.data_ptr() == v.data_ptr())
        dw_defer = in_bucket(dW, ctx.gW) and in_bucket(dWa, ctx.gWa)
        if ent:
            same = (W_aff is None) or (_v(g)[1:] == _v(dz)[1:])
            if same:
                gconv_dw(ent, dz, dz2=g if W_aff is not None else None, defer=dw_defer)      # one launch, one reduction
            else:
                gconv_dw([e for e in ent if not e.get("use_dz2")], dz, defer=dw_defer)
                gconv_dw([dict(e, use_dz2=False) for e in ent if e.get("use_dz2")], g, defer=dw_defer)
        if Cc and ctx.banked:
            dcoef_out = dcoef        # CondCoefFn turns it into the weight-row and condition gradients of all layers
        elif Cc:
            # rank-1 condition terms: dcoef[n,k,f] = sum_r (S_k 1)[r] dz[n,r,f]  (from bwd_prep)
            dcoef = dcoef.view(N, K * Fout)
            Wc = W[Ch * K:].view(Cc, K * Fout)
            if need_w:
                torch.mm(cond_in.t(), dcoef, out=dW[Ch * K:].view(Cc, K * Fout))
            if need_ci:
                dci = torch.mm(dcoef, Wc.t())
            if W_aff is not None:
                if need_wa:
                    torch.mm(cond_in.t(), dca, out=dWa[Ch:])
                if need_ci:
                    dci = dci + torch.mm(dca, W_aff[Ch:].t())
        if need_x:
            dx = alloc_act(N, Mi, Ch, dev, dtype=gfull.dtype)
            # W^T blocks read in place, contraction index (f) contiguous: B_k[f, c] = W[c*K + k, f]  (the
            # pipelined plain GEMM stages either weight layout at the same speed, so no transposed copy)
            wT = lambda k: (W, k * Fout, 1, K * Fout)
            waT = (W_aff, 0, 1, Fout) if W_aff is not None else None
            # (a tie goes to the form whose summed operator application can carry the activation gradient of the layer below)
            contract_first = (Mo < Mi) or (Mo == Mi and (Ch < Fout or (Ch == Fout and act_x is not None and W_aff is None and K > 1)))
            # fp16 two-piece operands of the data gradient: contraction over the Fout columns (backward planes)
            P, Pa = ctx.pieces
            # contraction length of the shortest launch of the branch taken: one order per launch (contract_first) or all
            # orders (+ the affine term) as sources of one launch
            ktot_bw = Fout if contract_first else Fout * (K + (1 if W_aff is not None else 0))
            bw_ok = P is not None and Fout % 32 == 0 and Ch >= 64 and g.dtype == torch.float32 and h2_shape_ok(ktot_bw, Ch)

            def src(xk, k, aff=False):
                e = dict(x=xk, csr=None, w=waT if aff else wT(k))
                if bw_ok:
                    e["p"], e["rm"] = (Pa.bwd(0) if aff else P.bwd(k)), rowmax(xk)
                return e
            bkw = dict(wsi=_ptr(P.bsc)) if bw_ok else {}
            if contract_first and W_aff is None and K > 1:
                # all K orders in ONE launch: G = dz W[:Ch*K]^T has column c*K + k; the epilogue stores it as K
                # channel blocks G_k (de-interleave), then dx = sum_k S_k^T G_k
                ChP = _pad4(Ch)
                Gall = alloc_act(N, Mo, K * ChP, dev, dtype=gfull.dtype)
                if bw_ok:
                    gconv_fwd([dict(x=dz, csr=None, w=(W, 0, 1, Fout), p=(P.b_hi.data_ptr(), P.b_lo.data_ptr(), Fout), rm=rowmax(dz))],
                              Gall, deinterleave=K, F=K * Ch, wsi=_ptr(P.bsi))
                else:
                    gconv_fwd([dict(x=dz, csr=None, w=(W, 0, 1, Fout))], Gall, deinterleave=K, F=K * Ch)
                Gs = [Gall[:, :, k * ChP:k * ChP + Ch] for k in range(K)]
                fused = None
                if act_x is not None and actgrad_fusable(Gs, act_x, Ch) and tuple(act_x.shape) == (N, Mi, Ch):
                    fused = spmm_multi(Gs, [ops.bwd[k] for k in range(K)], sum=True, act_x=act_x, act=ctx.prev_act.act)
                if fused is not None:
                    dx, part, chunks = fused
                    gB_prev = ctx.prev_act.gB
                    ctx.prev_act.fused = True
                    item = dict(ws=part, N=N, Mo=Mi, F=Ch, R=0, dbias=gB_prev.view(Ch), dcoef=None, dcoef_g=None, cstride=0, chunks=chunks)
                    if DEFERRED is not None:
                        DEFERRED.append(item)
                        dx._cape_is_dz = dict(dbias=None, offer=ctx.prev_act, version=dx._version)
                    else:
                        db = torch.empty(Ch, device=dev, dtype=torch.float32)
                        _finalize_bwd_prep([dict(item, dbias=db)])
                        dx._cape_is_dz = dict(dbias=db, offer=ctx.prev_act, version=dx._version)
                else:
                    dx = spmm_multi(Gs, [ops.bwd[k] for k in range(K)], sum=True)
            elif contract_first:
                # G_k = dz W_k^T at the Mo output rows, then dx = sum_k S_k^T G_k
                first = True
                for k in range(K):
                    ent = [src(dz, k)]
                    if W_aff is not None and k == 0:
                        ent.append(src(g, 0, aff=True))
                    if ops.bwd[k].identity and first:
                        gconv_fwd(ent, dx, **bkw)
                    else:
                        Gk = alloc_act(N, Mo, Ch, dev, dtype=gfull.dtype)
                        gconv_fwd(ent, Gk, **bkw)
                        if ops.bwd[k].identity:
                            dx.add_(Gk)
                        elif first:
                            spmm(Gk, ops.bwd[k], y=dx)
                        else:
                            spmm(Gk, ops.bwd[k], z=dx, beta=1.0, y=dx)
                    first = False
            else:
                # T_k = S_k^T dz at the Mi input rows, then one GEMM over all sources
                srcs = [(dz, ops.bwd[k]) for k in range(K)] + ([(g, ops.bwd[0])] if W_aff is not None else [])
                todo = [i for i, (_, c) in enumerate(srcs) if not c.identity]
                Ts = [t for t, _ in srcs]
                if Ts_pre is not None:
                    Ts = list(Ts_pre)
                elif len(todo) == 1 and todo[0] == 1 and T1_pre is not None:
                    Ts[1] = T1_pre
                elif len(todo) == 1:
                    Ts[todo[0]] = spmm(srcs[todo[0]][0], srcs[todo[0]][1])
                elif todo:                         # every S^T application of this layer in one launch
                    for i, ti in zip(todo, spmm_multi([srcs[i][0] for i in todo], [srcs[i][1] for i in todo])):
                        Ts[i] = ti
                if Fout == 1 and W_aff is None and Cc == 0 and dz.dtype == torch.float32 and W.shape[0] == Ch * K:
                    # one output channel (the discriminator's prediction map): dx[n, r, c] = sum_k T_k[n, r] W[c*K + k]
                    # is a rank-K outer product -- K broadcast multiply-adds instead of a GEMM launch whose
                    # contraction has length 1 (38 us in the generic kernel, three times per adversarial step)
                    Wm = W.detach().view(Ch, K)
                    dxv = dx[:, :, :Ch]
                    torch.mul(Ts[0][:, :, :1], Wm[:, 0], out=dxv)
                    for k in range(1, K):
                        dxv.addcmul_(Ts[k][:, :, :1], Wm[:, k])
                else:
                    ent = [src(Ts[k], k) for k in range(K)]
                    if W_aff is not None:
                        ent.append(src(Ts[K], 0, aff=True))
                    rm_dx = alloc_rm(dx) if (H2 and dx.dtype == torch.float32 and Ch >= 128) else None
                    gconv_fwd(ent, dx, rm_out=rm_dx, **bkw)
                    set_rm(dx, rm_dx)
                if ctx.coarse_dw:
                    # dW_k^T[f, c] = sum_{n, r} T_k[n, r, f] x[n, r, c]: the T_k are the sources, x the gradient operand
                    wen = []
                    if need_w:
                        wen += [dict(x=Ts[k], csr=None, w=(dW, k * Fout, 1, K * Fout)) for k in range(K)]
                    if W_aff is not None and need_wa:
                        wen.append(dict(x=Ts[K], csr=None, w=(dWa, 0, 1, Fout)))
                    if wen:
                        gconv_dw(wen, xs[0], defer=dw_defer)
        if Co and need_co:
            dco = reduce_cond(gfull[:, :, Fout:])
        return dx, dW, dB, dWa, dci, dco, None, None, None, None, None, None, None, dcoef_out


class ChebConvRecurrenceFn(torch.autograd.Function):
    """General-K chebyshev5 (lib/models.py:69-103) by the explicit recurrence
    x_k = 2 L~ x_{k-1} - x_{k-2} with standalone sparse applications; used above FUSE_MAX_K."""

    @staticmethod
    def forward(ctx, x, W, bias, ops, act, bias_mode):
        x = as_act(x)
        N, M, Cin = x.shape
        K, Fout = ops.K, W.shape[1]
        xs = [x, spmm(x, ops.Lt)]
        for k in range(2, K):
            xs.append(spmm(xs[-1], ops.Lt, alpha=2.0, z=xs[-2], beta=-1.0))
        y = alloc_act(N, M, Fout, x.device)
        gconv_fwd([dict(x=xs[k], csr=None, w=(W, k * Fout, K * Fout, 1)) for k in range(K)], y,
                  bias=bias, bias_mode=bias_mode, act=act)
        _trace_sign(y, act)
        ctx.ops, ctx.act, ctx.bias_mode, ctx.has_bias = ops, act, bias_mode, bias is not None
        ctx.save_for_backward(W, y, *xs)
        return y

    @staticmethod
    def backward(ctx, g):
        W, y = ctx.saved_tensors[:2]
        xs = ctx.saved_tensors[2:]
        ops, act = ctx.ops, ctx.act
        K, Fout = ops.K, W.shape[1]
        N, M, Cin = xs[0].shape
        g = as_act(g)
        dz = act_bwd(g, y, act) if act != "none" else g
        dW = dB = dx = None
        skip_w = bool(NO_WEIGHT_GRAD) and W.data_ptr() in NO_WEIGHT_GRAD
        if ctx.needs_input_grad[2] and ctx.has_bias and not skip_w:
            if ctx.bias_mode == _lib.BIAS_VERTEX:
                dB = torch.empty((1, M, Fout), device=g.device, dtype=torch.float32)
                colsum(dz, dB, per_vertex=True)
            else:
                dB = torch.empty((1, 1, Fout), device=g.device, dtype=torch.float32)
                colsum(dz, dB)
        if ctx.needs_input_grad[1] and not skip_w:
            dW = torch.empty_like(W)
            gconv_dw([dict(x=xs[k], csr=None, w=(dW, k * Fout, K * Fout, 1)) for k in range(K)], dz)
        if ctx.needs_input_grad[0]:
            # all G_k = dz W_k^T in ONE launch (de-interleaving epilogue), then Clenshaw for sum_k T_k(L~)^T G_k with
            # every step  b_k = 2 L~^T b_{k+1} + G_k - b_{k+2}  as one scaled multi-term launch
            CP = _pad4(Cin)
            Gall = alloc_act(N, M, K * CP, g.device)
            gconv_fwd([dict(x=dz, csr=None, w=(W, 0, 1, Fout))], Gall, deinterleave=K, F=K * Cin)
            G = [Gall[:, :, k * CP:k * CP + Cin] for k in range(K)]
            b1 = b2 = None   # b_{k+1}, b_{k+2}
            for k in range(K - 1, 0, -1):
                if b1 is None:
                    bk = G[k]
                elif b2 is None:
                    bk = spmm_multi([b1, G[k]], [ops.LtT, None], sum=True, scales=[2.0, 1.0])
                else:
                    bk = spmm_multi([b1, G[k], b2], [ops.LtT, None, None], sum=True, scales=[2.0, 1.0, -1.0])
                b1, b2 = bk, b1
            if b1 is None:
                dx = G[0]
            elif b2 is None:
                dx = spmm_multi([b1, G[0]], [ops.LtT, None], sum=True, scales=[1.0, 1.0])
            else:
                dx = spmm_multi([b1, G[0], b2], [ops.LtT, None, None], sum=True, scales=[1.0, 1.0, -1.0])
        return dx, dW, dB, None, None, None


class ChebConvFusedFn(torch.autograd.Function):
    """General-K chebyshev5 (lib/models.py:69-103) with the recurrence of :88-96 kept on chip (csrc/cheb_fused.hip): one
    launch forward, two (+ the fixed-order weight-gradient reduction) backward; the K-stack never reaches HBM."""

    @staticmethod
    def forward(ctx, x, W, ops, plan):
        _lib.require_gpu()
        x = as_act(x)
        N, M, Cin = x.shape
        K, Fout = ops.K, W.shape[1]
        if (x.stride(1) & 3) or (N > 1 and (x.stride(0) & 3)) or (x.data_ptr() & 15):
            xa = alloc_act(N, M, Cin, x.device)
            xa.copy_(x)
            x = xa
        W = W.contiguous()
        y = alloc_act(N, M, Fout, x.device)
        xp, xs, xl = _v(x)
        yp, ys, yl = _v(y)
        nnz = int(ops.host.Lt.nnz)
        flops = 2 * N * M * Cin * K * Fout + (K - 1) * 2 * nnz * Cin * N + max(K - 2, 0) * 2 * M * Cin * N
        _log_launch("cheb_fused_fwd_kernel", flops, 4 * (N * M * (Cin + Fout) + Cin * K * Fout) + 8 * nnz + 4 * (M + 1),
                    lambda: check(lib.cape_cheb_fused_fwd(xp, xs, xl, _ptr(W), yp, ys, yl, N, M, Cin, Fout, K, plan.P, _ptr(plan.pinfo),
                                                          _ptr(plan.vid), _ptr(plan.ell_col), _ptr(plan.ell_val), plan.rmax, _stream()), "cape_cheb_fused_fwd"))
        ctx.ops, ctx.plan = ops, plan
        ctx.save_for_backward(x, W)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W = ctx.saved_tensors
        ops, plan = ctx.ops, ctx.plan
        N, M, Cin = x.shape
        K, Fout = ops.K, W.shape[1]
        g = as_act(g)
        if (g.stride(1) & 3) or (N > 1 and (g.stride(0) & 3)) or (g.data_ptr() & 15):
            ga = alloc_act(N, M, Fout, g.device)
            ga.copy_(g)
            g = ga
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if NO_WEIGHT_GRAD and W.data_ptr() in NO_WEIGHT_GRAD:
            need_w = False                             # data-gradient-only sweep through this layer (see NO_WEIGHT_GRAD)
        if not (need_x or need_w):
            return None, None, None, None
        dx = alloc_act(N, M, Cin, x.device) if need_x else None
        dW = torch.empty_like(W) if need_w else None
        need = int(lib.cape_cheb_fused_bwd_workspace_bytes(N, Cin, Fout, K, plan.P))
        if need < 0:
            check(need, "cape_cheb_fused_bwd_workspace_bytes")
        ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        xp, xs, xl = _v(x)
        gp, gs, gl = _v(g)
        dp, ds, dl = _v(dx) if dx is not None else (None, 0, 0)
        nnz = int(ops.host.Lt.nnz)
        flops = 2 * (2 * N * M * Cin * K * Fout + (K - 1) * 2 * nnz * Cin * N + max(K - 2, 0) * 2 * M * Cin * N)
        _log_launch("cheb_fused_dw_kernel + cheb_fused_dx_kernel", flops, 4 * (2 * N * M * Cin + N * M * Fout + 2 * Cin * K * Fout) + 8 * nnz + 4 * (M + 1),
                    lambda: check(lib.cape_cheb_fused_bwd(xp, xs, xl, gp, gs, gl, _ptr(W), dp, ds, dl, _ptr(dW), 0, N, M, Cin, Fout, K,
                                                          plan.P, _ptr(plan.pinfo), _ptr(plan.vid), _ptr(plan.ell_col), _ptr(plan.ell_val),
                                                          plan.rmax, _ptr(ws), need, _stream()), "cape_cheb_fused_bwd"))
        return dx, dW, None, None


class SparseOpFn(torch.autograd.Function):
    """y[n] = P x[n]  -- poolwT (lib/models.py:129-152) as a standalone operator."""

    @staticmethod
    def forward(ctx, x, fwd_csr, bwd_csr):
        ctx.bwd_csr = bwd_csr
        return spmm(as_act(x), fwd_csr)

    @staticmethod
    def backward(ctx, g):
        return spmm(as_act(g), ctx.bwd_csr), None, None


class ConcatCondFn(torch.autograd.Function):
    """[x | cond tiled over vertices]  (fit_cond_dim + tf.concat, lib/models.py:533-536,663-666)."""

    @staticmethod
    def forward(ctx, x, cond):
        N, M, Cx = x.shape
        Cc = cond.shape[1]
        out = alloc_act(N, M, Cx + Cc, x.device, zero=False)
        out[:, :, :Cx].copy_(x)
        fill_cond(cond.contiguous(), out[:, :, Cx:])
        ctx.Cx = Cx
        return out

    @staticmethod
    def backward(ctx, g):
        g = as_act(g)
        dx = g[:, :, :ctx.Cx] if ctx.needs_input_grad[0] else None
        dc = reduce_cond(g[:, :, ctx.Cx:]) if ctx.needs_input_grad[1] else None
        return dx, dc


class ResidualLinearFn(torch.autograd.Function):
    """[ x W + r Wr | cond tiled over vertices ] -- the tail of res_block_decoder (lib/models.py:763-774): graph_linear_2
    on the block's features, graph_linear_input on the block's (unpooled) input, their sum and the tf.concat with the
    condition, as ONE two-source contraction that writes straight into the concatenated buffer (the reference's
    formulation is two products, an element-wise add and a concat copy).  Backward: one weight-gradient launch for both
    blocks, one data-gradient contraction per input."""

    @staticmethod
    def forward(ctx, x, r, W, Wr, cond, gW=None, gWr=None):
        x, r = _row_aligned(as_act(x)), _row_aligned(as_act(r))
        N, M, Cx = x.shape
        Cr, F = r.shape[2], W.shape[1]
        assert W.shape == (Cx, F) and Wr.shape == (Cr, F) and W.is_contiguous() and Wr.is_contiguous() and r.shape[:2] == (N, M)
        Cc = 0 if cond is None else cond.shape[1]
        yfull = alloc_act(N, M, F + Cc, x.device, dtype=x.dtype)
        e = [dict(x=x, csr=None, w=(W, 0, F, 1)), dict(x=r, csr=None, w=(Wr, 0, F, 1))]
        # fp16 two-piece operands: the two products add into one accumulator, so the two weights share their column scales
        P = Pr = None
        if H2 and x.dtype == torch.float32 and Cx % 32 == 0 and Cr % 32 == 0 and F % 8 == 0 and h2_shape_ok(Cx + Cr, F):
            P = pieces_for(W, Cx, 1, fpair=(Wr.detach(), Cr))
            Pr = pieces_for(Wr, Cr, 1, fpair=(W.detach(), Cx)) if P is not None else None
        kw = {}
        if Pr is not None:
            e[0]["p"], e[0]["rm"], e[1]["p"], e[1]["rm"] = P.fwd(0), rowmax(x), Pr.fwd(0), rowmax(r)
            kw = dict(wsi=_ptr(P.fsi))
        rm_y = alloc_rm(yfull, F) if (H2 and x.dtype == torch.float32 and F >= 128) else None
        gconv_fwd(e, yfull[:, :, :F], rm_out=rm_y, **kw)
        if Cc:
            fill_cond(cond.contiguous(), yfull[:, :, F:])
        elif rm_y is not None:
            set_rm(yfull, rm_y)
        ctx.F, ctx.Cc, ctx.gW, ctx.gWr, ctx.pieces = F, Cc, gW, gWr, (P, Pr)
        ctx.save_for_backward(x, r, W, Wr)
        return yfull

    @staticmethod
    def backward(ctx, gfull):
        x, r, W, Wr = ctx.saved_tensors
        F = ctx.F
        gfull = _row_aligned(as_act(gfull))
        g = gfull[:, :, :F]
        set_rm(g, rm_of(gfull))
        N, M, Cx = x.shape
        Cr = r.shape[2]
        need_x, need_r, need_w, need_wr, need_c = (ctx.needs_input_grad[i] for i in range(5))
        if NO_WEIGHT_GRAD and W.data_ptr() in NO_WEIGHT_GRAD:
            need_w = need_wr = False
        P, Pr = ctx.pieces
        bw = Pr is not None and F % 32 == 0 and g.dtype == torch.float32
        if (bw or H2) and g.dtype == torch.float32 and F % 32 == 0 and F >= 64:
            rowmax(g)                                   # one pass serves both data-gradient contractions and the weight gradient
        dx = dr = dW = dWr = dc = None
        ent = []
        if need_w:
            dW = _grad_buffer(W, ctx.gW)
            ent.append(dict(x=x, csr=None, w=(dW, 0, F, 1)))
        if need_wr:
            dWr = _grad_buffer(Wr, ctx.gWr)
            ent.append(dict(x=r, csr=None, w=(dWr, 0, F, 1)))
        if ent:
            in_bucket = lambda t, v: t is None or (v is not None and t.data_ptr() == v.data_ptr())
            gconv_dw(ent, g, defer=in_bucket(dW, ctx.gW) and in_bucket(dWr, ctx.gWr))
        if need_x:
            dx = alloc_act(N, M, Cx, x.device, dtype=g.dtype)
            if bw and Cx >= 64 and h2_shape_ok(F, Cx):
                gconv_fwd([dict(x=g, csr=None, w=(W, 0, 1, F), p=P.bwd(0), rm=rowmax(g))], dx, wsi=_ptr(P.bsc))
            else:
                gconv_fwd([dict(x=g, csr=None, w=(W, 0, 1, F))], dx)          # W^T read in place (contraction index contiguous)
        if need_r:
            dr = alloc_act(N, M, Cr, x.device, dtype=g.dtype)
            if bw and Cr >= 64 and h2_shape_ok(F, Cr):
                gconv_fwd([dict(x=g, csr=None, w=(Wr, 0, 1, F), p=Pr.bwd(0), rm=rowmax(g))], dr, wsi=_ptr(Pr.bsc))
            else:
                gconv_fwd([dict(x=g, csr=None, w=(Wr, 0, 1, F))], dr)
        if ctx.Cc and need_c:
            dc = reduce_cond(gfull[:, :, F:])
        return dx, dr, dW, dWr, dc, None, None


def group_count(N, Cn, G=32):
    """Number of normalisation groups the reference's ``gn`` forms for ``Cn`` channels (lib/models.py:693-699): it
    reshapes [N, C, V] to [-1, G, C // G, V] with G = min(32, C) and a FREE leading dimension.  When G divides C that is G
    groups per sample; otherwise the rows of the [N*C, V] matrix are taken w = C // G at a time irrespective of the sample
    boundaries -- C / w groups of w consecutive channels per sample when w divides C (w = 1, i.e. G < C < 2G: a
    per-(sample, channel) normalisation).  Only groups that would straddle samples are refused (w does not divide C).
    TensorFlow additionally rejects the reshape unless N*C is a multiple of G*w; where w divides C the groups are well
    defined for any N, so small-batch / demo inference on such layers is accepted here and equals the reference wherever
    the reference runs at all."""
    Ge = min(int(G), int(Cn))
    if Cn % Ge == 0:
        return Ge
    w = Cn // Ge
    if Cn % w:
        raise ValueError("group norm: %d channels cannot be split into groups of %d inside each sample "
                         "(reference lib/models.py:698 reshape)" % (Cn, w))
    return Cn // w


class GroupNormFn(torch.autograd.Function):
    """gn(norm_type='group') optionally fused with the following tf.nn.relu
    (lib/models.py:681-712, 751-760).  Saved for backward: the input, the per-(sample, group) statistics and the
    per-(sample, channel) scale / shift -- not the output (the ReLU mask is re-derived from the same fma).

    ``passthrough``: also return the input as a second output (a view).  A block that reads the same tensor on a
    residual branch (res_block_decoder, lib/models.py:744-774: ``x + xu``) takes that output instead of the input itself:
    both gradients of the input then arrive HERE and the apply kernel sums them in its own pass (``dx_add``) -- without
    it autograd adds them with a separate element-wise launch per block."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, relu, passthrough=False, g_gamma=None, g_beta=None):
        _lib.require_gpu()
        x = as_act(x)
        assert x.dtype == torch.float32, "group norm reads fp32 activations"
        N, V, Cn = x.shape
        if (x.stride(1) & 3) or (N > 1 and (x.stride(0) & 3)) or (x.data_ptr() & 15):
            xa = alloc_act(N, V, Cn, x.device)
            xa.copy_(x)
            x = xa
        y = alloc_act(N, V, Cn, x.device)
        stats = torch.empty((N, G, 2), device=x.device, dtype=torch.float32)
        coef = torch.empty((N, 4, _pad4(Cn)), device=x.device, dtype=torch.float32)      # per-channel tables: stride round_up(C, 4)
        need = int(lib.cape_groupnorm_workspace_bytes(N, V, Cn))
        ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        xp, xs, xl = _v(x)
        yp, ys, yl = _v(y)
        rm = _new_rm(y) if _want_rm(y, any_width=True) else None      # the apply pass bounds the rows of its output for the contraction next
        _log_launch("groupnorm_fwd", 0, 3 * 4 * N * V * Cn,
                    lambda: check(lib.cape_groupnorm_fwd(xp, xs, xl, _ptr(gamma), _ptr(beta), float(eps), int(G), int(relu), yp, ys, yl,
                                                         _ptr(stats), _ptr(coef), N, V, Cn, _ptr(ws), need, _ptr(rm), _stream()),
                                  "cape_groupnorm_fwd"))
        set_rm(y, rm)
        if relu:
            _trace_sign(y, "relu")             # y = relu(fma(a, x, b)): positive exactly where the kernel's fma was
        ctx.G, ctx.relu, ctx.passthrough = G, relu, bool(passthrough)
        ctx.g_gamma, ctx.g_beta = g_gamma, g_beta          # bucket views of the parameter gradients (or None)
        ctx.save_for_backward(x, gamma, stats, coef)
        if passthrough:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, g, g_pass=None):
        x, gamma, stats, coef = ctx.saved_tensors
        N, V, Cn = x.shape

        def aligned(t):
            t = as_act(t)
            if (t.stride(1) & 3) or (N > 1 and (t.stride(0) & 3)) or (t.data_ptr() & 15):
                ta = alloc_act(N, V, Cn, t.device)
                ta.copy_(t)
                t = ta
            return t

        if g is None:                              # only the pass-through output was used
            return g_pass, None, None, None, None, None, None, None, None
        g = aligned(g)
        dx = alloc_act(N, V, Cn, x.device)
        dgb = torch.empty((2, N, Cn), device=x.device, dtype=torch.float32)        # per-sample dgamma / dbeta partials
        bcoef = torch.empty((N, 3, _pad4(Cn)), device=x.device, dtype=torch.float32)
        need = int(lib.cape_groupnorm_workspace_bytes(N, V, Cn))
        ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        xp, xs, xl = _v(x)
        gp, gs, gl = _v(g)
        dp, ds, dl = _v(dx)
        if g_pass is not None:
            g_pass = aligned(g_pass)
            ap, as_, al = _v(g_pass)
        else:
            ap, as_, al = None, 0, 0
        rm = _new_rm(dx) if _want_rm(dx, any_width=True) else None
        _log_launch("groupnorm_bwd", 0, (5 + (1 if g_pass is not None else 0)) * 4 * N * V * Cn,
                    lambda: check(lib.cape_groupnorm_bwd(xp, xs, xl, gp, gs, gl, _ptr(gamma), _ptr(stats), _ptr(coef), int(ctx.G),
                                                         int(ctx.relu), dp, ds, dl, ap, as_, al, _ptr(dgb[0]), _ptr(dgb[1]),
                                                         _ptr(bcoef), N, V, Cn, _ptr(ws), need, _ptr(rm), _stream()), "cape_groupnorm_bwd"))
        set_rm(dx, rm)
        gg, gb_ = ctx.g_gamma, ctx.g_beta
        if (DEFERRED is not None and LAUNCH_LOG is None and gg is not None and gb_ is not None and gg.shape == (Cn,)
                and gb_.shape == (Cn,) and gg.is_contiguous() and gb_.is_contiguous()):
            # the sums over the batch of all group norms of the sweep run as one launch at its end (flush_deferred), straight
            # into the gradient bucket
            DEFERRED_GN.append((dgb, gg, gb_))
            return dx, gg, gb_, None, None, None, None, None, None
        dgb = dgb.sum(1)                                                           # one launch for both parameter gradients
        return dx, dgb[0], dgb[1], None, None, None, None, None, None


UNIT_GRAD = None        # the 0-dim tensor holding 1.0 that the training step seeds its backward pass with (models._one_scalar)


_SCALARS = {}


def _const_scalar(value, device):
    """A cached 0-dim device tensor holding ``value`` (gradients that are constants of the graph)."""
    key = (float(value), str(device))
    t = _SCALARS.get(key)
    if t is None:
        t = torch.full((), float(value), device=device, dtype=torch.float32)
        _SCALARS[key] = t
    return t


class ReconEdgeLossFn(torch.autograd.Function):
    """total = w_recon * mean|pred-gt| + w_edge * edge_loss [+ w_a * term_a + term_b]  (lib/models.py:357-375, 393-394,
    lib/losses.py:9-25).  Returns (total, [recon, edge]); only ``total`` is differentiable.  ``term_a`` (a 0-dim tensor, e.g. the
    latent term; differentiable, its gradient is the constant w_a) and ``term_b`` (0-dim, a value without gradient: the
    regulariser) are added by the kernel that finishes the loss, not by element-wise launches afterwards."""

    @staticmethod
    def forward(ctx, pred, gt, verts_ref, edges, vptr, vidx, w_recon, w_edge, term_a=None, w_a=0.0, term_b=None):
        _lib.require_gpu()
        gt = gt.contiguous()
        N, M, _ = pred.shape
        # the decoder's output is a [N, M, 3] view of 16-byte rows: read it where it lies (no re-homing launch)
        if pred.stride(2) != 1 or pred.stride(1) < 3 or (N > 1 and pred.stride(0) != M * pred.stride(1)):
            pred = pred.contiguous()
        ldp = int(pred.stride(1)) if M > 1 else 3
        E = edges.shape[0]
        if L1_SIGN_TRACE is not None and w_recon != 0.0:
            L1_SIGN_TRACE.append(torch.sign(pred.detach() - gt).cpu())       # the kernel takes the sign of the same fp32 difference
        need = lib.cape_recon_edge_workspace_bytes(N, M, E)
        ws = torch.empty((need + 3) // 4, device=pred.device, dtype=torch.float32)
        out = torch.empty(2, device=pred.device, dtype=torch.float32)
        total = torch.empty((), device=pred.device, dtype=torch.float32)
        dpred = alloc_act(N, M, 3, pred.device, zero=False)                  # row-padded like every activation gradient
        ldd = int(dpred.stride(1))
        for t in (term_a, term_b):
            assert t is None or (t.dim() == 0 and t.dtype == torch.float32 and t.device == pred.device)
        _log_launch("recon_edge_loss", 0, N * (E * 24 + M * 36),
                    lambda: check(lib.cape_recon_edge_loss_fwd_bwd(_ptr(pred), ldp, _ptr(gt), _ptr(verts_ref), _ptr(edges), _ptr(vptr),
                                                                   _ptr(vidx), N, M, E, float(w_recon), float(w_edge), _ptr(out),
                                                                   _ptr(total), _ptr(term_a), float(w_a), _ptr(term_b), _ptr(dpred),
                                                                   ldd, _ptr(ws), need, _stream()),
                                  "cape_recon_edge_loss_fwd_bwd"))
        ctx.set_materialize_grads(False)         # no zeros for the non-differentiable parts' gradient
        ctx.save_for_backward(dpred)
        ctx.w_a = float(w_a) if term_a is not None else None
        ctx.mark_non_differentiable(out)
        return total, out

    @staticmethod
    def backward(ctx, gtotal, _gout):
        (dpred,) = ctx.saved_tensors
        if gtotal is None:
            return (None,) * 11
        unit = UNIT_GRAD is not None and gtotal.data_ptr() == UNIT_GRAD.data_ptr()      # d(loss)/d(total) is the caller's constant 1
        ga = None
        if ctx.w_a is not None and ctx.needs_input_grad[8]:
            ga = _const_scalar(ctx.w_a, dpred.device) if unit else gtotal * ctx.w_a
        return (dpred if unit else dpred * gtotal), None, None, None, None, None, None, None, ga, None, None


class GanLossFn(torch.autograd.Function):
    """(lambda * gan_g, lambda * gan_d, [gan_g, gan_d]) from the discriminator's logits (lib/models.py:381-390,397):
    sigmoid cross entropy with smoothed labels, both means and both gradients in ONE launch (csrc/loss.hip gan_bce_kernel)
    instead of ~25 element-wise ones.  ``logits``: the merged pass [Nf + Nr, M, 1] (generated samples first, ``real`` None)
    or the generated samples' logits with ``real`` the second tensor.  The third output is not differentiable."""

    @staticmethod
    def forward(ctx, logits, real, nf, smooth, lam):
        _lib.require_gpu()
        logits = as_act(logits)
        assert logits.dtype == torch.float32 and logits.shape[2] == 1
        M = logits.shape[1]
        if real is None:
            fake, rl = logits[:nf], logits[nf:]
        else:
            fake, rl = logits, as_act(real)
            assert rl.dtype == torch.float32 and rl.shape[1:] == logits.shape[1:] and nf == logits.shape[0]
        Nf, Nr = fake.shape[0], rl.shape[0]
        fp, fs, fl = _v(fake)
        rp, rs, rld = _v(rl)
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        sg = torch.empty((), device=logits.device, dtype=torch.float32)
        sd = torch.empty((), device=logits.device, dtype=torch.float32)
        ga = torch.empty((Nf + Nr, M, 1), device=logits.device, dtype=torch.float32)
        gb = torch.empty_like(ga)
        check(lib.cape_gan_bce_fwd_bwd(fp, fs, fl, rp, rs, rld, Nf, Nr, M, float(smooth), float(lam), _ptr(out), _ptr(sg), _ptr(sd),
                                       _ptr(ga), _ptr(gb), _stream()), "cape_gan_bce_fwd_bwd")
        ctx.save_for_backward(ga, gb)
        ctx.split = None if real is None else Nf
        ctx.mark_non_differentiable(out)
        # each sweep differentiates ONE of the two losses: the other output's gradient must arrive as None, not as a
        # materialised zero tensor (an extra multiply + add per sweep, and 0 * inf = NaN in the live term)
        ctx.set_materialize_grads(False)
        return sg, sd, out

    @staticmethod
    def backward(ctx, gg, gd, _gout):
        ga, gb = ctx.saved_tensors

        def times(t, g):
            if g is None:
                return None
            return t if (UNIT_GRAD is not None and g.data_ptr() == UNIT_GRAD.data_ptr()) else t * g

        a, b = times(ga, gg), times(gb, gd)
        tot = a if b is None else (b if a is None else a + b)
        if tot is None:
            return None, None, None, None, None
        if ctx.split is None:
            return tot, None, None, None, None
        return tot[:ctx.split], tot[ctx.split:], None, None, None


# --------------------------------------------------------------------------------------------
# functional front-ends with the reference's operator names
# --------------------------------------------------------------------------------------------
def chebyshev5(x, W, ops, bias=None, activation=None, cond=None, W_affine=None, cond_in=None, grad_bufs=(None, None),
               bias_grad_buf=None, coef=None):
    """Graph convolution (lib/models.py:69-103) with optional fused bias+activation
    (``activation`` in b1leakyrelu/b1relu/b1tanh/b2relu), affine branch, rank-1 input condition
    (``cond_in``, or its coefficients ``coef`` precomputed by CondCoefFn) and materialised output condition
    concat (``cond``)."""
    if activation is None:
        act, bmode = "none", (_lib.BIAS_NONE if bias is None else
                              (_lib.BIAS_VERTEX if bias.shape[1] > 1 else _lib.BIAS_CHANNEL))
    else:
        act, bmode = _ACT_OF[activation]
    if ops.fused:
        return ChebConvFn.apply(x, W, bias, W_affine, cond_in, cond, ops, act, bmode, "twopass", grad_bufs[0], grad_bufs[1],
                                bias_grad_buf, coef)
    assert W_affine is None and coef is None
    if cond_in is not None:
        x = ConcatCondFn.apply(x, cond_in)
    plan = None
    if FUSED_RECURRENCE and x.is_cuda and x.dtype == torch.float32 and W.shape[0] == x.shape[2] * ops.K:
        plan = ops.patch_plan(x.shape[2], W.shape[1])
    if plan is not None:
        # recurrence on chip; a bias / activation of the layer runs as the standalone element-wise operator
        y = ChebConvFusedFn.apply(x, W, ops, plan)
        if bias is not None or act != "none":
            y = BiasActFn.apply(y, bias, act, bmode)
    else:
        y = ChebConvRecurrenceFn.apply(x, W, bias, ops, act, bmode)
    if cond is not None:
        y = ConcatCondFn.apply(y, cond)
    return y


class CondCoefFn(torch.autograd.Function):
    """Rank-1 condition coefficients of every consumer of one condition vector, one launch (csrc/cond.hip).

    ``layers``: list of dicts  W [(Ch+Cc)*K, F], Wa (None or [Ch+Cc, F]), Ch, K, gW, gWa  -- gW / gWa are the
    gradient-bucket views of W / Wa.  Returns one contiguous coef [N, K (+1), F] per layer (ChebConvFn's
    ``coef`` input).  The weights are deliberately NOT autograd inputs: backward writes the gradient rows of the
    condition channels (rows >= Ch*K of gW, >= Ch of gWa) straight into the bucket views -- disjoint from the rows
    ChebConvFn's weight-gradient kernel writes -- and returns only d(cond)."""

    @staticmethod
    def _descr(layers, N, coefs=None, dcoefs=None, grads=False):
        arr = (_lib.CapeCondLayer * len(layers))()
        for i, ly in enumerate(layers):
            W, Wa, Ch, K = ly["W"], ly["Wa"], int(ly["Ch"]), int(ly["K"])
            F = int(W.shape[1])
            d = arr[i]
            d.K, d.F = K, F
            d.w = W.data_ptr() + 4 * Ch * K * F
            d.w_aff = None if Wa is None else Wa.data_ptr() + 4 * Ch * F
            d.coef = None if coefs is None else coefs[i].data_ptr()
            d.dcoef = None if dcoefs is None else dcoefs[i].data_ptr()
            d.gw = d.gw_aff = None
            if grads:
                gW, gWa = ly["gW"], ly["gWa"]
                assert gW.shape == W.shape and gW.is_contiguous() and (Wa is None or (gWa.shape == Wa.shape and gWa.is_contiguous()))
                d.gw = gW.data_ptr() + 4 * Ch * K * F
                d.gw_aff = None if Wa is None else gWa.data_ptr() + 4 * Ch * F
        return arr

    @staticmethod
    def forward(ctx, cond, layers):
        _lib.require_gpu()
        cond = cond.contiguous()
        N, Cc = cond.shape
        coefs = []
        for ly in layers:
            W, Wa, K = ly["W"], ly["Wa"], int(ly["K"])
            assert W.is_contiguous() and W.shape[0] == (int(ly["Ch"]) + Cc) * K and (Wa is None or Wa.is_contiguous())
            coefs.append(torch.empty((N, K + (0 if Wa is None else 1), W.shape[1]), device=cond.device, dtype=torch.float32))
        arr = CondCoefFn._descr(layers, N, coefs=coefs)
        check(lib.cape_cond_coef_fwd(C.c_void_p(cond.data_ptr()), Cc, N, Cc, arr, len(layers), _stream()), "cape_cond_coef_fwd")
        ctx.layers = layers
        ctx.save_for_backward(cond)
        return tuple(coefs)

    @staticmethod
    def backward(ctx, *dcoefs):
        flush_deferred()                 # the dcoef reductions of the decoder layers were queued, finish them now
        (cond,) = ctx.saved_tensors
        layers = ctx.layers
        N, Cc = cond.shape
        dcoefs = [torch.zeros((N, int(ly["K"]) + (0 if ly["Wa"] is None else 1), ly["W"].shape[1]), device=cond.device)
                  if d is None else d.contiguous() for d, ly in zip(dcoefs, layers)]
        arr = CondCoefFn._descr(layers, N, dcoefs=dcoefs, grads=True)
        dcond = torch.empty_like(cond) if ctx.needs_input_grad[0] else None
        check(lib.cape_cond_coef_bwd(C.c_void_p(cond.data_ptr()), Cc, N, Cc, arr, len(layers), _ptr(dcond), Cc, 0, _stream()),
              "cape_cond_coef_bwd")
        return dcond, None


class CondNetsFn(torch.autograd.Function):
    """Both condition networks (reference lib/models.py:479-511, :284-290) in one launch per direction (csrc/condnet.hip):
    ycat = [ leaky_relu(c1 W1 + b1) W2 + b2 | c2 Wc + bc ].  ``gbufs``: the six gradient-bucket views (W1, b1, W2, b2, Wc,
    bc) or None -- with views the backward kernel writes the bucket directly."""

    @staticmethod
    def forward(ctx, c1, c2, W1, b1, W2, b2, Wc, bc, gbufs, copies=1):
        """``copies`` 2: returns (ycat, ycat_b), the same values in two buffers -- a consumer that reads ycat_b alone hands its
        gradient to the backward kernel directly (which differentiates the sum) instead of through an element-wise add."""
        _lib.require_gpu()
        c1, c2 = c1.contiguous(), c2.contiguous()
        N, in1 = c1.shape
        hid, out1, in2, out2 = int(W1.shape[1]), int(W2.shape[1]), int(c2.shape[1]), int(Wc.shape[1])
        assert all(t.is_contiguous() and t.dtype == torch.float32 for t in (W1, b1, W2, b2, Wc, bc, c1, c2))
        assert tuple(W1.shape) == (in1, hid) and tuple(W2.shape) == (hid, out1) and tuple(Wc.shape) == (in2, out2)
        h = torch.empty((N, hid), device=c1.device, dtype=torch.float32)
        ycat = torch.empty((N, out1 + out2), device=c1.device, dtype=torch.float32)
        ycat_b = torch.empty_like(ycat) if copies == 2 else None
        check(lib.cape_condnet_fwd(_ptr(c1), in1, _ptr(c2), in2, _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(Wc), _ptr(bc),
                                   _ptr(h), _ptr(ycat), _ptr(ycat_b), N, in1, hid, out1, in2, out2, _stream()), "cape_condnet_fwd")
        _trace_sign(h, "leaky")
        ctx.gbufs, ctx.dims = gbufs, (N, in1, hid, out1, in2, out2)
        ctx.save_for_backward(c1, c2, W2, h)
        ctx.shapes = [tuple(t.shape) for t in (W1, b1, W2, b2, Wc, bc)]
        ctx.set_materialize_grads(False)
        return ycat if ycat_b is None else (ycat, ycat_b)

    @staticmethod
    def backward(ctx, dycat, dycat_b=None):
        c1, c2, W2, h = ctx.saved_tensors
        N, in1, hid, out1, in2, out2 = ctx.dims
        if dycat is None and dycat_b is None:
            return (None,) * 10
        rows = lambda t: None if t is None else (t if t.stride(1) == 1 and (N == 1 or t.stride(0) >= out1 + out2) else t.contiguous())
        dycat, dycat_b = rows(dycat), rows(dycat_b)
        ld = lambda t: 0 if t is None else (int(t.stride(0)) if N > 1 else out1 + out2)
        outs = []
        for i, shp in enumerate(ctx.shapes):
            v = None if ctx.gbufs is None else ctx.gbufs[i]
            if v is not None and v.is_contiguous() and v.numel() == int(np.prod(shp)):
                outs.append(v.view(shp))
            else:
                outs.append(torch.empty(shp, device=c1.device, dtype=torch.float32))
        check(lib.cape_condnet_bwd(_ptr(c1), in1, _ptr(c2), in2, _ptr(W2), _ptr(h), _ptr(dycat), ld(dycat), _ptr(dycat_b),
                                   ld(dycat_b), *[_ptr(t) for t in outs], N, in1, hid, out1, in2, out2, _stream()),
              "cape_condnet_bwd")
        return (None, None) + tuple(outs) + (None, None)


def poolwT(x, fwd_csr, bwd_csr):
    return SparseOpFn.apply(x, fwd_csr, bwd_csr)


class BiasActFn(torch.autograd.Function):
    """Standalone act(x + bias): b1leakyrelu / b1relu / b1tanh / b2relu (lib/models.py:105-127),
    used where the activation cannot be fused into the producing conv (encoder res_block)."""

    @staticmethod
    def forward(ctx, x, bias, act, bias_mode):
        x = as_act(x)
        y = bias_act_fwd(x, bias, bias_mode, act)
        _trace_sign(y, act)
        ctx.act, ctx.bias_mode = act, bias_mode
        ctx.save_for_backward(y)
        ctx.bshape = None if bias is None else tuple(bias.shape)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = as_act(g)
        dz = act_bwd(g, y, ctx.act) if ctx.act != "none" else g
        dB = None
        if ctx.bshape is not None and ctx.needs_input_grad[1]:
            dB = torch.empty(ctx.bshape, device=g.device, dtype=torch.float32)
            colsum(dz, dB, per_vertex=(ctx.bias_mode == _lib.BIAS_VERTEX))
        return dz, dB, None, None


def brelu(x, bias, activation):
    act, bmode = _ACT_OF[activation]
    return BiasActFn.apply(x, bias, act, bmode)


def _parr(tensors):
    """HOST array of device pointers (None -> NULL)."""
    return (C.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


class FcLongFn(torch.autograd.Function):
    """y_m = x @ W_m + b_m for 1 or 2 matrices sharing a [N <= 64, in] input with a very long ``in`` (encoder
    fc_mean / fc_var, reference lib/models.py:555-560): weight-streaming kernels of csrc/fc.hip -- two launches
    forward, one backward (dW_m, db_m written into the gradient bucket views, one dx for both matrices)."""

    @staticmethod
    def forward(ctx, x, gbufs, *wb):
        _lib.require_gpu()
        x = x.contiguous()
        N, kin = x.shape
        Ws, bs = list(wb[0::2]), list(wb[1::2])
        nmat, out = len(Ws), int(Ws[0].shape[1])
        assert all(W.is_contiguous() and tuple(W.shape) == (kin, out) for W in Ws)
        ys = [torch.empty((N, out), device=x.device, dtype=torch.float32) for _ in Ws]
        need = int(lib.cape_fc_long_workspace_bytes(N, kin, out, nmat))
        if need < 0:
            check(need, "cape_fc_long_workspace_bytes")
        ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        _log_launch("fc_long_fwd", 2 * N * kin * out * nmat, 4 * (kin * out * nmat + N * kin + N * out * nmat),
                    lambda: check(lib.cape_fc_long_fwd(C.c_void_p(x.data_ptr()), kin, N, kin, out, nmat, _parr(Ws), _parr(bs), _parr(ys),
                                                       C.c_void_p(ws.data_ptr()), need, _stream()), "cape_fc_long_fwd"))
        ctx.gbufs, ctx.nmat, ctx.has_b = gbufs, nmat, [b is not None for b in bs]
        ctx.save_for_backward(x, *Ws)
        return tuple(ys)

    @staticmethod
    def backward(ctx, *gs):
        x, *Ws = ctx.saved_tensors
        N, kin = x.shape
        out, nmat = int(Ws[0].shape[1]), ctx.nmat
        gs = [torch.zeros((N, out), device=x.device) if g is None else g.contiguous() for g in gs]
        gbufs = ctx.gbufs or [(None, None)] * nmat
        dWs, dbs = [], []
        for m in range(nmat):
            dWs.append(_grad_buffer(Ws[m], gbufs[m][0]) if ctx.needs_input_grad[2 + 2 * m] else None)
            if ctx.has_b[m] and ctx.needs_input_grad[3 + 2 * m]:
                gb = gbufs[m][1]
                dbs.append(gb if (gb is not None and gb.numel() == out and gb.is_contiguous()) else
                           torch.empty(out, device=x.device, dtype=torch.float32))
            else:
                dbs.append(None)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _log_launch("fc_long_bwd", 4 * N * kin * out * nmat, 4 * (2 * kin * out * nmat + 2 * N * kin + N * out * nmat),
                    lambda: check(lib.cape_fc_long_bwd(C.c_void_p(x.data_ptr()), kin, N, kin, out, nmat, _parr(Ws), _parr(gs), _parr(dWs),
                                                       _parr(dbs), _ptr(dx), kin, _stream()), "cape_fc_long_bwd"))
        grads = [dx, None]
        for m in range(nmat):
            grads += [dWs[m], dbs[m]]
        return tuple(grads)


class FcWideFn(torch.autograd.Function):
    """y = act(x @ W + b) for a [N <= 64, in <= 200] input and a very wide output (decoder fc1, reference
    lib/models.py:579-583, bias and leaky-ReLU fused): one launch forward, three backward (csrc/fc.hip)."""

    @staticmethod
    def forward(ctx, x, W, b, act, gW, gb):
        _lib.require_gpu()
        x = x.contiguous()
        N, kin = x.shape
        out = int(W.shape[1])
        assert W.is_contiguous() and W.shape[0] == kin
        y = torch.empty((N, out), device=x.device, dtype=torch.float32)
        _log_launch("fc_wide_fwd", 2 * N * kin * out, 4 * (kin * out + N * kin + N * out),
                    lambda: check(lib.cape_fc_wide_fwd(C.c_void_p(x.data_ptr()), kin, N, kin, out, C.c_void_p(W.data_ptr()), _ptr(b),
                                                       _lib.ACT[act], C.c_void_p(y.data_ptr()), out, _stream()), "cape_fc_wide_fwd"))
        _trace_sign(y, act)
        ctx.act, ctx.gW, ctx.gb, ctx.has_b = act, gW, gb, b is not None
        ctx.save_for_backward(x, W, y)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W, y = ctx.saved_tensors
        N, kin = x.shape
        out = int(W.shape[1])
        g = g.contiguous()
        dW = _grad_buffer(W, ctx.gW) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_b and ctx.needs_input_grad[2]:
            gb = ctx.gb
            db = gb if (gb is not None and gb.numel() == out and gb.is_contiguous()) else torch.empty(out, device=x.device, dtype=torch.float32)
        dx = ws = None
        need = 0
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            need = int(lib.cape_fc_wide_bwd_workspace_bytes(N, kin, out))
            ws = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
        _log_launch("fc_wide_bwd", 4 * N * kin * out, 4 * (2 * kin * out + 2 * N * kin + 2 * N * out),
                    lambda: check(lib.cape_fc_wide_bwd(C.c_void_p(x.data_ptr()), kin, C.c_void_p(g.data_ptr()), out, C.c_void_p(y.data_ptr()),
                                                       out, _lib.ACT[ctx.act], N, kin, out, C.c_void_p(W.data_ptr()), _ptr(dW), _ptr(db),
                                                       _ptr(dx), kin, _ptr(ws), need, _stream()), "cape_fc_wide_bwd"))
        return dx, dW, db, None, None, None


_FC_LONG = 8192       # a side at least this long takes the weight-streaming kernels


def _fc_long_ok(x, kernels):
    return x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and x.shape[1] >= _FC_LONG and x.shape[0] * kernels[0].shape[1] * len(kernels) <= 8192


def dense(x, kernel, bias, activation=None, grad_bufs=(None, None)):
    """tf.layers.dense: act(x @ kernel + bias), activation in (None, 'leaky_relu')."""
    kin, kout = kernel.shape
    act = "none" if activation is None else "leaky"
    if _fc_long_ok(x, [kernel]):
        y = FcLongFn.apply(x, [grad_bufs], kernel, bias)[0]
    elif x.is_cuda and x.dim() == 2 and x.shape[0] <= 64 and kout >= _FC_LONG and kin <= 200 and x.shape[0] * kin <= 12288:
        return FcWideFn.apply(x, kernel, bias, act, grad_bufs[0], grad_bufs[1])
    else:
        y = torch.addmm(bias, x, kernel)
    if activation == 'leaky_relu':
        y = torch.nn.functional.leaky_relu(y, 0.2)
        _trace_sign(y, "leaky")
    return y


def dense_pair(x, k0, b0, k1, b1, grad_bufs=((None, None), (None, None))):
    """Two dense layers on the same input (encoder fc_mean / fc_var): one pass over x when it is long."""
    if _fc_long_ok(x, [k0, k1]) and k0.shape == k1.shape:
        return FcLongFn.apply(x, list(grad_bufs), k0, b0, k1, b1)
    return dense(x, k0, b0, grad_bufs=grad_bufs[0]), dense(x, k1, b1, grad_bufs=grad_bufs[1])


def _ranges_arg(ranges):
    flat = [int(v) for r in ranges for v in r]
    return (C.c_int64 * max(len(flat), 1))(*flat), len(ranges)


def flat_workspace(device):
    return torch.empty(int(lib.cape_flat_workspace_bytes()) // 4, device=device, dtype=torch.float32)


def flat_gradnorm(g, w, ranges, coef, sumsq_out, ws, grad_scale=1.0):
    """sumsq_out <- sum (grad_scale*g + coef*w on ``ranges``)^2 over a flat bucket (two deterministic launches)."""
    _lib.require_gpu()
    arr, nr = _ranges_arg(ranges)
    _log_launch("flat_gradnorm", 0, 4 * g.numel(),
                lambda: check(lib.cape_flat_gradnorm(C.c_void_p(g.data_ptr()), _ptr(w), g.numel(), arr, nr, float(coef), float(grad_scale),
                                                     C.c_void_p(sumsq_out.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel() * 4,
                                                     _stream()), "cape_flat_gradnorm"))
    return sumsq_out


def flat_momentum_update(w, g, m, momentum, clip, sumsq, neg_lr, ranges, coef, grad_scale=1.0):
    """clip-by-global-norm + momentum update of a flat bucket in one launch (csrc/optim.hip)."""
    _lib.require_gpu()
    arr, nr = _ranges_arg(ranges)
    _log_launch("flat_momentum_update", 0, 5 * 4 * w.numel(),
                lambda: check(lib.cape_flat_momentum_update(C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                                            w.numel(), float(momentum), float(clip), C.c_void_p(sumsq.data_ptr()),
                                                            C.c_void_p(neg_lr.data_ptr()), arr, nr, float(coef), float(grad_scale), _stream()),
                              "cape_flat_momentum_update"))


def flat_adam_update(w, g, m, v, beta1, beta2, eps, clip, sumsq, neg_lr, state, ranges, coef, grad_scale=1.0):
    """clip-by-global-norm + Adam update of a flat bucket in one launch (csrc/optim.hip; reference lib/models.py:447-449);
    ``state``: int32[2] device tensor {number of updates so far, 0}, advanced by the launch."""
    _lib.require_gpu()
    assert state.dtype == torch.int32 and state.numel() >= 2 and v.shape == w.shape
    arr, nr = _ranges_arg(ranges)
    _log_launch("flat_adam_update", 0, 7 * 4 * w.numel(),
                lambda: check(lib.cape_flat_adam_update(C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                                        C.c_void_p(v.data_ptr()), w.numel(), float(beta1), float(beta2), float(eps),
                                                        float(clip), C.c_void_p(sumsq.data_ptr()), C.c_void_p(neg_lr.data_ptr()),
                                                        C.c_void_p(state.data_ptr()), arr, nr, float(coef), float(grad_scale), _stream()),
                              "cape_flat_adam_update"))


def sumsq_ranges(x, ranges, scale, ws):
    """scale * sum of x^2 over element ranges of a flat buffer -> 0-dim tensor."""
    _lib.require_gpu()
    arr, nr = _ranges_arg(ranges)
    out = torch.empty((), device=x.device, dtype=torch.float32)
    check(lib.cape_sumsq_ranges(C.c_void_p(x.data_ptr()), arr, nr, float(scale), C.c_void_p(out.data_ptr()),
                                C.c_void_p(ws.data_ptr()), ws.numel() * 4, _stream()), "cape_sumsq_ranges")
    return out


class VaeSampleKLFn(torch.autograd.Function):
    """z = mean + exp(0.5*logvar)*eps  and  kl = mean_n( -0.5 * sum_j(1 + logvar - mean^2 - exp(logvar)) )
    (reference lib/models.py:193-196, :371-372): one launch forward, one backward (csrc/optim.hip) instead of the
    ~25 an op-by-op autograd tape replays on these [N, nz] tensors."""

    @staticmethod
    def forward(ctx, mean, logvar, eps, cond=None):
        """``cond`` [N, Cc]: returned z is [z | cond] (the decoder's input, lib/models.py:296 tf.concat) from the same launch."""
        _lib.require_gpu()
        mean, logvar, eps = mean.contiguous(), logvar.contiguous(), eps.contiguous()
        N, nz = mean.shape
        Cc = 0 if cond is None else int(cond.shape[1])
        if cond is not None and cond.stride(1) != 1:
            cond = cond.contiguous()
        z = torch.empty((N, nz + Cc), device=mean.device, dtype=torch.float32)
        kl = torch.empty((), device=mean.device, dtype=torch.float32)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        check(lib.cape_vae_sample_kl_fwd(p(mean), p(logvar), p(eps), p(z), nz + Cc, p(kl), N, nz, p(cond),
                                         0 if cond is None else int(cond.stride(0)), Cc, _stream()), "cape_vae_sample_kl_fwd")
        ctx.save_for_backward(mean, logvar, eps)
        ctx.Cc = Cc
        return z, kl

    @staticmethod
    def backward(ctx, gz, gkl):
        mean, logvar, eps = ctx.saved_tensors
        N, nz = mean.shape
        dmean, dlv = torch.empty_like(mean), torch.empty_like(mean)
        if gz is not None and gz.stride(1) != 1:
            gz = gz.contiguous()
        gkl = None if gkl is None else gkl.contiguous()
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        check(lib.cape_vae_sample_kl_bwd(p(mean), p(logvar), p(eps), p(gz), 0 if gz is None else int(gz.stride(0)), p(gkl), p(dmean), p(dlv),
                                         N, nz, _stream()), "cape_vae_sample_kl_bwd")
        dcond = gz[:, nz:] if (ctx.Cc and gz is not None) else None          # (a view: the condition's own gradient, no copy)
        return dmean, dlv, None, dcond
