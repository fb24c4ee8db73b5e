#!/usr/bin/env python3
"""bench.py -- meshes/sec of the CAPE-affineconv_nz64 training step on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of 16 synthetic meshes PER GPU: the full
CAPE-affineconv_nz64 Mesh-CVAE (reference configs/CAPE-affineconv_nz64_pose32_clotype32_male.yaml)
forward + backward + gradient clipping + Momentum update (BASELINE.json configs[2]); with
``--gan`` the mesh-patch discriminator passes and its update are included as well (the reference's
adversarial step).  Inputs are resident in HBM before the timed region; weights are the reference
initialisers (random), data is synthetic N(0,1) displacements on the real SMPL mesh hierarchy.
Prints ONE JSON line (rank 0).  Data-parallel (weak scaling): per-GPU batch fixed, one flat
gradient all-reduce per step over RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, chip table (fp32 matrix = vector peak)
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same table, dense bf16
# gemm_split_kernel computes every fp32 multiply-add as six bf16 MFMA products (exact three-way operand split): its
# ceiling in ALGORITHMIC fp32 flops is the dense bf16 peak / 6
BF16X6_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
# gemm_h2_kernel / dw_h2_kernel: three fp16 MFMA products per multiply-add (two-piece operands); dense fp16 peak = dense bf16 peak
F16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
HBM_PEAK_GBS = 8000.0


def build_model(batch, local_rank, config, act_dtype='fp32'):
    from cape_amd.configs import cape_params
    from cape_amd.load_data import load_graph_mtx
    from cape_amd.models import CAPE
    L, D, U, p, L_d, D_d, _ = load_graph_mtx(None, load_for_demo=True)
    # lr schedule as main.py:67 builds it for the male dataset (README.md:53: 31 036 training meshes, the last
    # 100 held out as validation, lib/load_data.py:64-65): decay_steps = decay_every * n_train / batch_size, and
    # the warm-up (lr_warmup: 1) spans 8 * decay_steps steps -- i.e. the timed steps run at the small learning
    # rates a real run starts with (a 0.008 step on N(0,1) data diverges from the reference initialisers).
    decay_steps = 2 * (31036 - 100) / 16
    params = cape_params(config, p=p, batch_size=batch, name='bench', decay_steps=decay_steps, act_dtype=act_dtype)
    model = CAPE(L=L, D=D, U=U, L_d=L_d, D_d=D_d, device='cuda:%d' % local_rank, **params)
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='train')
    return model


def synthetic_batch(model, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    B, M = model.batch_size, model.input_num_verts
    r = lambda *s: torch.randn(*s, generator=g)
    x = r(B, M, 3)
    clo = torch.eye(4)[torch.arange(B) % 4]
    return dict(data_g=x, gt=x + 0.1 * r(B, M, 3), data_d=r(B, M, 3), cond_g=0.5 * r(B, 126), cond_d=0.5 * r(B, 126),
                cond2_g=clo, cond2_d=clo.roll(1, 0), eps=r(B, int(model.nz)))


# SURVEY.md section 8(d), config 3 (affine-nz64 generator fwd+bwd at N = 16, reference formulation, fp32):
# 256.5 GFLOP and 5.75 GB algorithmic per step -> per mesh 16.03 GFLOP and 359.6 MB
STEP_ALG_GFLOP_PER_MESH = 16.03
STEP_ALG_MB_PER_MESH = 359.6
# one discriminator pass (SURVEY 8d row "3 + one D pass"): fwd+bwd 21.9 GFLOP, 0.57 GB at N = 16; the adversarial step
# runs D on the real and on the generated batch
DPASS_ALG_GFLOP_PER_MESH = 1.37
DPASS_ALG_MB_PER_MESH = 35.5


def _csrc_fingerprint():
    """sha256 over the kernel sources: PMC traffic figures are only attached to a bench line when they were collected
    from exactly this code (tools/pmc_merge.py stamps the same fingerprint into the summary)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cape_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def _pmc_entry(kernel):
    """The committed PMC summary's record of ``kernel`` (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ counters in separate
    passes, tools/collect_profiles.sh; counters cannot be read from inside this process), or {} when the summary was
    collected from different kernel code."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        if pmc.get("_meta", {}).get("csrc_sha") != _csrc_fingerprint():
            return {}
        key = next((k for k in pmc if k.replace(" ", "") == kernel.replace(" ", "")), None)
        return {} if key is None else pmc[key]
    except Exception:
        return {}


def _pmc_step_bytes(kernel, launches_per_step):
    """HBM bytes of ONE step according to the committed counter pass: sum over every kernel of that run of bytes per dispatch x
    dispatches, divided by the number of steps the run made (= dispatches of ``kernel`` / its launches per step; the run's few
    set-up dispatches -- fills, initialisers -- are included, < 1 %).  None when the summary belongs to other kernel sources."""
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
        if pmc.get("_meta", {}).get("csrc_sha") != _csrc_fingerprint():
            return None
        key = next((k for k in pmc if k.replace(" ", "") == kernel.replace(" ", "")), None)
        steps = pmc[key]["dispatches"] / float(launches_per_step)
        if steps < 1 or abs(steps - round(steps)) > 1e-6:
            return None
        return int(sum(v["hbm_bytes_per_dispatch"] * v["dispatches"] for k, v in pmc.items() if k != "_meta" and v.get("hbm_bytes_per_dispatch")) / steps)
    except Exception:
        return None


def _pmc_traffic(kernel):
    """HBM bytes per launch (1024 * (2 * FETCH_SIZE + WRITE_SIZE): the guide's gfx950 correction) or None."""
    return _pmc_entry(kernel).get("hbm_bytes_per_dispatch")


def _mfma_products(kernel):
    """MFMA products the kernel executes per algorithmic multiply-add, and the dense peak of the pipe it runs them on."""
    if kernel.startswith(("gemm_h2_kernel", "gemm_h2x_kernel", "dw_h2_kernel")):
        return 3, BF16_MFMA_PEAK_TFLOPS
    if kernel.startswith(("gemm_split_kernel", "dw_split_kernel")):
        return (2 if "unsigned short" in kernel else 6), BF16_MFMA_PEAK_TFLOPS
    return 1, FP32_MFMA_PEAK_TFLOPS


def kernel_roofline(runner):
    """Per-launch HIP-event timing of EVERY library launch (contractions fwd / dX / dW, slab reductions, sparse
    operators, backward-prep, dense layers, loss, optimiser) in three eager passes of the same step (graph replays
    cannot be bracketed per kernel; kernels and shapes are identical); returns the roofline object of the kernel
    with the largest total time and the per-kernel table."""
    from cape_amd import ops
    ops.LAUNCH_LOG = []
    torch.cuda.synchronize()
    reps = 3
    for _ in range(reps):
        runner._fwd_bwd()
        runner._update()
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    # an event pair around NOTHING still measures the record-to-record gap of the stream (a few us): calibrate
    # it on empty brackets and subtract it, so that the per-launch figure is the kernel's own duration
    empty = []
    for _ in range(50):
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        a1.record()
        empty.append((a0, a1))
    torch.cuda.synchronize()
    bracket_ms = float(np.median([a0.elapsed_time(a1) for a0, a1 in empty]))
    agg = {}
    for name, flops, byts, e0, e1 in log:
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += max(e0.elapsed_time(e1) - bracket_ms, 1e-4) * 1e-3
        a[2] += flops
        a[3] += byts
    if not agg:
        return None, {}
    dom = max(agg, key=lambda k: agg[k][1])
    n, t, fl, by = agg[dom]
    table = {k: dict(launches=v[0] // reps, avg_us=1e6 * v[1] / v[0], total_us=1e6 * v[1] / reps, tflops=v[2] / v[1] / 1e12,
                     alg_gbs=v[3] / v[1] / 1e9) for k, v in agg.items()}
    for k, row in table.items():
        # MFMA utilisation, two ways: executed matrix-pipe flops (algorithmic x products per multiply-add) over the dense peak
        # of that pipe at the maximum clock -- live, from this run's timing -- and, when the committed counter summary belongs
        # to these kernel sources, SQ_VALU_MFMA_BUSY_CYCLES over SIMD-cycles (tools/pmc_merge.py), i.e. against the clock the
        # kernel actually ran at
        if row["tflops"] > 0.5:
            prod, pipe_peak = _mfma_products(k)
            row["mfma_util_of_dense_peak"] = row["tflops"] * prod / pipe_peak
            pm = _pmc_entry(k).get("mfma_util")
            if pm is not None:
                row["mfma_util_pmc"] = pm
    split = dom.startswith(("gemm_split_kernel", "dw_split_kernel"))
    h2 = dom.startswith(("gemm_h2_kernel", "gemm_h2x_kernel", "dw_h2_kernel"))   # fp16 two-piece operands: three fp16 MFMA products per multiply-add
    one_product = split and "unsigned short" in dom          # bf16 storage: one bf16 MFMA product per multiply-add
    mfma_peak = BF16_MFMA_PEAK_TFLOPS if one_product else BF16X6_PEAK_TFLOPS if split else F16X3_PEAK_TFLOPS if h2 else \
        FP32_MFMA_PEAK_TFLOPS
    # the bound of THIS kernel: time at the HBM peak vs time at its matrix-pipe peak for its algorithmic work
    t_hbm, t_mfma = by / n / (HBM_PEAK_GBS * 1e9), fl / n / (mfma_peak * 1e12)
    bound = "mfma" if t_mfma >= t_hbm else "hbm"
    if bound == "mfma":
        achieved, peak, unit = fl / t / 1e12, mfma_peak, "TFLOP/s"
        basis = "bf16 operands, fp32 accumulate: dense bf16 MFMA peak" if one_product else \
            ("fp32 result on the bf16 MFMA pipe, 6 bf16 products per multiply-add: dense bf16 peak 2500 / 6; "
             "achieved counts algorithmic fp32 flops (x6 = bf16 MFMA flops executed)") if split else \
            ("fp32 result on the fp16 MFMA pipe, 3 fp16 products per multiply-add on two-piece operands: dense fp16 peak "
             "2500 / 3; achieved counts algorithmic fp32 flops (x3 = fp16 MFMA flops executed)") if h2 \
            else "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32)"
    else:
        achieved, peak, unit = by / t / 1e9, HBM_PEAK_GBS, "GB/s"
        basis = "HBM3E 8 TB/s; achieved counts algorithmic bytes (operands touched once)"
    roof = dict(bound=bound, kernel=dom, achieved=round(achieved, 2), peak=round(peak, 1), unit=unit,
                frac=round(achieved / peak, 4), traffic=_pmc_traffic(dom), peak_basis=basis,
                launches_per_step=n // reps, avg_launch_us=round(1e6 * t / n, 2),
                alg_flop_per_launch=fl / n, alg_bytes_per_launch=by / n,
                tflops=round(fl / t / 1e12, 2),
                hbm_alg_gbs=round(by / t / 1e9, 1), hbm_frac=round(by / t / 1e9 / HBM_PEAK_GBS, 4),
                share_of_logged_time=round(t / sum(v[1] for v in agg.values()), 4))
    prod, pipe_peak = _mfma_products(dom)
    # scalars at the top level of the object (a parser that drops nested dicts still sees them):
    #   mfma_util            = executed MFMA flops (algorithmic x products per multiply-add) / dense peak of the pipe at the maximum
    #                          clock, from this run's live timing
    #   mfma_util_pmc        = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) from the committed counter pass of the
    #                          same kernel sources (null otherwise): against the clock the kernel actually ran at
    #   frac_of_pipe_ceiling = algorithmic flops / (dense peak / products per multiply-add): 2500 / 3 = 833 TFLOP/s for the
    #                          two-piece kernels -- the ceiling of THIS arithmetic on its pipe (equals `frac` when the bound is mfma)
    roof["mfma_util"] = round(fl / t / 1e12 * prod / pipe_peak, 4)
    roof["mfma_util_pmc"] = _pmc_entry(dom).get("mfma_util")
    roof["frac_of_pipe_ceiling"] = round(fl / t / 1e12 / (pipe_peak / prod), 4)
    roof["products_per_multiply_add"], roof["pipe_peak_tflops"] = prod, pipe_peak
    return roof, table


# SURVEY 8(d) config 4 (non-affine GraphCMR/group-norm generator nz18, fp32): 757 GFLOP / 22.0 GB at N = 32 and one
# discriminator pass 35.2 GFLOP / 1.05 GB
CMR_ALG_GFLOP_PER_MESH, CMR_ALG_MB_PER_MESH = 23.66, 688.0
CMR_DPASS_ALG_GFLOP_PER_MESH, CMR_DPASS_ALG_MB_PER_MESH = 1.10, 32.8


def step_roofline(ms_per_step, batch, gan, bf16=False, cmr=False):
    """Step-level fraction: t_roof = max(alg. bytes / HBM peak, alg. flops / matrix-pipe peak of the dtype)
    of the reference formulation of the step, over the measured step time.  bf16 storage (SURVEY 8d cfg 5): half the
    bytes, dense bf16 MFMA peak -> HBM-bound."""
    if cmr:
        gf = CMR_ALG_GFLOP_PER_MESH + (2 * CMR_DPASS_ALG_GFLOP_PER_MESH if gan else 0.0)
        mb = CMR_ALG_MB_PER_MESH + (2 * CMR_DPASS_ALG_MB_PER_MESH if gan else 0.0)
    else:
        gf = STEP_ALG_GFLOP_PER_MESH + (2 * DPASS_ALG_GFLOP_PER_MESH if gan else 0.0)
        mb = (STEP_ALG_MB_PER_MESH + (2 * DPASS_ALG_MB_PER_MESH if gan else 0.0)) * (0.5 if bf16 else 1.0)
    t_mfma = batch * gf * 1e9 / ((BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS) * 1e12) * 1e3
    t_hbm = batch * mb * 1e6 / (HBM_PEAK_GBS * 1e9) * 1e3
    t_roof = max(t_mfma, t_hbm)
    return dict(t_roof_ms=round(t_roof, 4), t_mfma_ms=round(t_mfma, 4), t_hbm_ms=round(t_hbm, 4),
                bound="hbm" if t_hbm >= t_mfma else "mfma",
                frac=round(t_roof / ms_per_step, 4), frac_hbm=round(t_hbm / ms_per_step, 4),
                basis="SURVEY 8(d) reference-formulation work per mesh: %.2f GFLOP, %.1f MB fwd+bwd; %s MFMA peak %.1f TF, "
                      "HBM %.0f GB/s" % (gf, mb, "bf16" if bf16 else "fp32", BF16_MFMA_PEAK_TFLOPS if bf16 else FP32_MFMA_PEAK_TFLOPS,
                                         HBM_PEAK_GBS))


def cpu_baseline(batch=16, iters=5, budget_s=150.0):
    """The CPU oracle (numpy/torch restatement of the reference's TF1 graph, reference op order, fp32)
    timed on this host: forward+backward of the same CVAE step at the benchmarked batch (BASELINE.md section 3:
    same batch, >= 5 timed passes after one untimed pass that builds the variables).  ``budget_s`` bounds the
    sample on a slow host: timing stops early (never below 2 passes) once it is spent."""
    from cape_amd.load_data import load_graph_mtx, load_pack
    from oracle.torch_twin import TwinCAPE
    from oracle.configs import cape_params as oracle_params
    L, D, U, p, L_d, D_d, _ = load_graph_mtx(None, load_for_demo=True)
    pack = load_pack()
    P = oracle_params('affine_nz64', batch)
    twin = TwinCAPE(L, D, U, L_d, D_d, p=p, dtype=np.float64, tdtype=torch.float32, verts_ref=pack['template_verts'],
                    vpe=pack['edges_smpl'], **P)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((batch, 6890, 3)).astype(np.float32)
    cond = rng.standard_normal((batch, 126)).astype(np.float32)
    clo = np.eye(4, dtype=np.float32)[np.arange(batch) % 4]
    eps = rng.standard_normal((batch, P['nz'])).astype(np.float32)
    times = []
    t_begin = time.time()
    for it in range(iters + 1):
        t0 = time.time()
        y, y2 = twin.cond_embeddings(cond, clo)
        xh, zm, zl = twin.generator(x, y, y2, eps)
        ls = twin.losses(xh, x, zm, zl)
        torch.autograd.grad(ls['loss_g'], [v for n, v in twin.params.items() if not n.startswith('discriminator')],
                            allow_unused=True)
        if it:                        # first pass builds the variables
            times.append(time.time() - t0)
        if len(times) >= 2 and time.time() - t_begin > budget_s:
            break
    t = float(np.median(times))
    return dict(value=round(batch / t, 3), unit="meshes/s", cores=int(torch.get_num_threads()), kind="port",
                sample="torch-CPU fp32 restatement of the TF1 graph (reference op order), CVAE fwd+bwd, batch %d, "
                       "median of %d timed passes (%.1f s of CPU work)" % (batch, len(times), sum(times)))


def exact_fp32_run(args, env_over=None, note=None):
    """The same step with every contraction on the exact-fp32 MFMA (CAPE_H2=0 CAPE_GEMM_BF16X6=0 CAPE_DW_BF16X6=0; the library reads the
    knobs once per process, hence a child process): reported NEXT TO the headline so that both arithmetic paths are measured by
    the same bench invocation.  Never replaces ``value``; any failure is reported as a string instead of aborting the line.
    ``env_over`` / ``note``: another knob setting measured the same way (the six-product bf16 split of round 3)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--steps', str(min(args.steps, 30)), '--warmup', str(min(args.warmup, 5)),
           '--batch', str(args.batch), '--config', args.config, '--no-cpu-baseline', '--no-roofline', '--no-ab', '--no-extras']
    if args.gan:
        cmd.append('--gan')
    try:
        env = dict(os.environ, **(env_over or dict(CAPE_H2='0', CAPE_GEMM_BF16X6='0', CAPE_DW_BF16X6='0')))
        out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=180, check=True)
        line = [l for l in out.stdout.decode().splitlines() if l.startswith('{')][-1]
        r = json.loads(line)
        return {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
                "note": note or "same step, CAPE_H2=0 CAPE_GEMM_BF16X6=0 CAPE_DW_BF16X6=0: all contractions on v_mfma_f32_32x32x2_f32"}
    except Exception as e:                                  # noqa: BLE001 -- the comparison is optional
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def _timed_replays(runner, steps, warmup):
    for _ in range(warmup):
        runner.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        runner.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


def measure_extra_step(config, batch, gan, dtype, steps=20, warmup=3, want_roofline=True):
    """One of BASELINE.json's other configs on THIS GPU, measured like the headline (captured step, inputs resident):
    ms per step, step-level roofline fraction and the dominant kernel with its own roofline fraction."""
    from cape_amd.runtime import GraphedTrainStep
    model = build_model(batch, torch.cuda.current_device(), config, act_dtype=dtype)
    runner = GraphedTrainStep(model, with_gan=gan)
    runner.load_batch(**synthetic_batch(model, seed=4321))
    torch.cuda.synchronize()
    runner.capture()
    ms = _timed_replays(runner, steps, warmup)
    out = dict(ms_per_step=round(ms, 4), meshes_per_s=round(1e3 * batch / ms, 1), batch=batch, steps=steps)
    sr = step_roofline(ms, batch, gan, dtype == 'bf16', cmr=config.startswith("CAPE_nz18"))
    out["step_roofline_frac"], out["step_roofline_bound"], out["t_roof_ms"] = sr["frac"], sr["bound"], sr["t_roof_ms"]
    if want_roofline:
        roof, _ = kernel_roofline(runner)
        if roof is not None:
            out["dominant_kernel"] = dict(kernel=roof["kernel"], bound=roof["bound"], frac=roof["frac"], achieved=roof["achieved"],
                                          unit=roof["unit"], avg_launch_us=roof["avg_launch_us"], launches_per_step=roof["launches_per_step"])
    del runner, model
    torch.cuda.empty_cache()
    return out


def measure_config1(iters=50):
    """BASELINE.json configs[1]: single Chebyshev K = 6 graph-conv layer fwd+bwd, 64 x 6890 x 16 -> 32 (reference
    lib/models.py:69-103 with the explicit recurrence of :88-96), HIP-graph replay, against SURVEY 8(d)'s roofline
    (9.6 GFLOP / 198 MB algorithmic: 60.8 us at the fp32-MFMA peak, 24.8 us at 8 TB/s)."""
    from cape_amd import ops
    from cape_amd.graph import ConvOperators
    from cape_amd.load_data import load_graph_mtx
    L = load_graph_mtx(None, load_for_demo=True)[0]
    dev = torch.device('cuda', torch.cuda.current_device())
    N, Cin, Fout, K = 64, 16, 32, 6
    dops = ops.DeviceConvOps(ConvOperators(L[0], K), dev)
    g = torch.Generator(device='cpu').manual_seed(0)
    x = torch.randn(N, 6890, Cin, generator=g).to(dev).requires_grad_(True)
    W = (0.1 * torch.randn(Cin * K, Fout, generator=g)).clamp_(-0.2, 0.2).to(dev).requires_grad_(True)
    dy = torch.randn(N, 6890, Fout, generator=g).to(dev)

    def step():
        y = ops.chebyshev5(x, W, dops)
        torch.autograd.grad(y, [x, W], dy)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        graph.replay()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / iters
    ops.LAUNCH_LOG = []
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    agg = {}
    for name, _, _, e0, e1 in log:
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e-3
    nnz = int(L[0].nnz)
    X, Y, Wt = N * 6890 * Cin, N * 6890 * Fout, Cin * K * Fout
    flops = 3 * (2 * N * 6890 * Cin * K * Fout + (K - 1) * 2 * nnz * Cin * N + (K - 2) * 2 * 6890 * Cin * N)
    byts = 4 * (3 * X + 2 * Y + 3 * Wt) + 2 * (8 * nnz + 4 * 6891)
    t_roof = max(flops / (FP32_MFMA_PEAK_TFLOPS * 1e12), byts / (HBM_PEAK_GBS * 1e9))
    return dict(workload="single Chebyshev K=6 layer fwd+bwd, 64x6890x16->32", ms_per_step=round(1e3 * t, 4),
                meshes_per_s=round(N / t, 1), alg_gflop=round(flops / 1e9, 2), alg_mb=round(byts / 1e6, 1),
                tflops=round(flops / t / 1e12, 2), alg_gbs=round(byts / t / 1e9, 1), t_roof_ms=round(1e3 * t_roof, 4),
                step_roofline_frac=round(t_roof / t, 4), step_roofline_bound="mfma (fp32)" if flops / (FP32_MFMA_PEAK_TFLOPS * 1e12) >= byts / (HBM_PEAK_GBS * 1e9) else "hbm",
                kernels={k: dict(launches=v[0] // 3, avg_us=round(1e6 * v[1] / v[0], 2)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])})


# xGMI message model of SURVEY 8(e): every GPU has 7 point-to-point links of ~153 GB/s; with direct reduce-scatter +
# all-gather every peer pair uses its own link, so an all-reduce of S bytes over N GPUs moves 2 * S / N per link
XGMI_LINK_GBS = 153.0
COLLECTIVE_LATENCY_MS = 0.03          # per all-reduce launch (RCCL launch + first/last hop); one-rank RCCL measures 0.01
SPLIT_STEP_OVERHEAD_MS = 0.10         # two-graph step + collective launches, measured with a one-rank RCCL group (dp_selftest)


def modelled_scaling(ms_by_batch, grad_bytes, late_fraction, phase2_share=0.35):
    """MODELLED (not measured) 1/2/4/8-GPU table, as SURVEY 8(e) asks for when one GPU is reachable: per-GPU step times are
    the ones measured on THIS GPU at the per-GPU batch of each row; the exchange is the message model above, with the
    overlap cape_amd/runtime.py implements: the early (1 - late_fraction) of the gradient bucket is reduced while backward
    phase 2 (encoder convolutions: ~phase2_share of the step) runs, the late part after it."""
    def comm_ms(nbytes, n):
        return 0.0 if n == 1 else 1e3 * 2.0 * nbytes / n / (XGMI_LINK_GBS * 1e9) + COLLECTIVE_LATENCY_MS

    def row(n, b):
        t = ms_by_batch[b]
        if n == 1:
            return t, 0.0, 0.0
        early = comm_ms((1.0 - late_fraction) * grad_bytes, n)
        late = comm_ms(late_fraction * grad_bytes, n)
        exposed = max(0.0, early - phase2_share * t) + late
        return t + SPLIT_STEP_OVERHEAD_MS + exposed, early + late, exposed

    out = dict(label="MODELLED, not measured: single-GPU step times measured in this run + xGMI message model (SURVEY 8e)",
               assumptions=dict(grad_bucket_mb=round(grad_bytes / 1e6, 1), late_fraction=round(late_fraction, 4),
                                xgmi_link_gbs=XGMI_LINK_GBS, all_reduce="direct reduce-scatter + all-gather, 2*S/N bytes per link",
                                collective_latency_ms=COLLECTIVE_LATENCY_MS, split_step_overhead_ms=SPLIT_STEP_OVERHEAD_MS,
                                overlap="early part hidden behind backward phase 2 (%.0f %% of the step)" % (100 * phase2_share)),
               weak=[], strong=[])
    b0 = max(ms_by_batch)
    t1 = ms_by_batch[b0]
    for n in (1, 2, 4, 8):
        t, comm, exposed = row(n, b0)
        out["weak"].append(dict(n_gpus=n, per_gpu_batch=b0, ms_per_step=round(t, 4), meshes_per_s=round(1e3 * n * b0 / t, 1),
                                comm_ms=round(comm, 4), exposed_comm_ms=round(exposed, 4), efficiency=round(t1 / t, 4)))
        b = b0 // n
        if b in ms_by_batch:
            t, comm, exposed = row(n, b)
            out["strong"].append(dict(n_gpus=n, per_gpu_batch=b, ms_per_step=round(t, 4), meshes_per_s=round(1e3 * b0 / t, 1),
                                      comm_ms=round(comm, 4), exposed_comm_ms=round(exposed, 4), efficiency=round(t1 / (n * t), 4)))
    return out


def extra_measurements(args, headline_ms, model):
    """Time-boxed: the other BASELINE configs on this GPU and the inputs of the modelled scaling table."""
    t_begin = time.time()
    extras, budget = {}, float(os.environ.get("CAPE_BENCH_EXTRAS_BUDGET_S", "150"))
    st = model._opt_state['g']
    grad_bytes = 4 * int(st['flat_grad'].numel())
    late_fraction = 1.0 - float(st['split_off']) / float(st['flat_grad'].numel())

    def guarded(name, fn):
        if time.time() - t_begin > budget:
            extras[name] = {"skipped": "time box of %.0f s spent" % budget}
            return None
        try:
            extras[name] = fn()
            return extras[name]
        except Exception as e:                                  # noqa: BLE001 -- never lose the headline line
            extras[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
            return None

    guarded("config4_bf16_shard16", lambda: measure_extra_step(args.config, 16, False, 'bf16'))
    guarded("config3_nz18_gan_b32", lambda: measure_extra_step("CAPE_nz18_pose24_clotype8_male", 32, True, 'fp32', steps=10, warmup=2))
    guarded("config1_k6_layer", measure_config1)
    ms_by_batch = {args.batch: headline_ms}
    for b in (args.batch // 2, args.batch // 4, args.batch // 8):
        if b >= 1:
            r = guarded("strong_scaling_input_batch%d" % b, lambda b=b: measure_extra_step(args.config, b, args.gan, 'fp32', want_roofline=False))
            if r is not None and "ms_per_step" in r:
                ms_by_batch[b] = r["ms_per_step"]
    return extras, modelled_scaling(ms_by_batch, grad_bytes, late_fraction)


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks here (a harness valid
    for world_size 1..8 however it is invoked), one process per GPU, rendezvous on 127.0.0.1, and hand their output through.
    Refuses to run when the box has fewer than N devices unless the ranks are told to share one (CAPE_FORCE_DEVICE, the
    1-GPU functional check over gloo) -- a silent one-rank line under an `n_gpus: N` request is the failure this prevents."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus and "CAPE_FORCE_DEVICE" not in os.environ:
        raise SystemExit("--gpus %d but only %d HIP device(s) visible (set CAPE_FORCE_DEVICE=<id> with CAPE_DIST_BACKEND=gloo "
                         "to let the ranks share one device for a functional check)" % (args.gpus, have))
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=16, help='meshes per GPU per step')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong scaling (SURVEY 8e: 16 -> 16/8/4/2 per GPU): the GLOBAL batch is fixed and sharded over the '
                         'ranks (per-GPU batch = global / world); default 0 = weak scaling with --batch per GPU')
    ap.add_argument('--config', default='CAPE-affineconv_nz64_pose32_clotype32_male')
    ap.add_argument('--dtype', choices=('fp32', 'bf16'), default='fp32',
                    help="storage type of the mesh activations: fp32 (the headline, parity path) or bf16 (BASELINE configs[4]: "
                         "bf16 activations / bf16 operands with fp32 accumulation and fp32 master weights)")
    ap.add_argument('--gan', action='store_true', help='include the discriminator passes/update (adversarial step)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extras', action='store_true',
                    help='skip the time-boxed extra measurements (BASELINE configs[1], [3], [4] on this GPU and the modelled '
                         '1/2/4/8 scaling table); they run only at N = 1 for the default fp32 headline')
    ap.add_argument('--no-ab', action='store_true',
                    help='skip the short exact-fp32-MFMA comparison run (a child process with CAPE_GEMM_BF16X6=0)')
    ap.add_argument('--host-inputs', action='store_true',
                    help='hand every step a fresh batch of HOST numpy arrays (as fit() does): the PCIe-inclusive rate '
                         'quoted in DESIGN.md; never the headline value')
    args = ap.parse_args()
    relaunch_under_torchrun(args)

    from cape_amd import dist as cdist
    from cape_amd.runtime import GraphedTrainStep
    import torch.distributed as tdist
    # CAPE_DIST_BACKEND=gloo + CAPE_FORCE_DEVICE=0 let the N>1 code path be exercised on a 1-GPU box
    world, rank, local = cdist.init_from_env(backend=os.environ.get("CAPE_DIST_BACKEND"))
    if "CAPE_FORCE_DEVICE" in os.environ:
        local = int(os.environ["CAPE_FORCE_DEVICE"])
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch %d is not divisible by the %d ranks" % (args.global_batch, world))
        args.batch = args.global_batch // world
    model = build_model(args.batch, local, args.config, act_dtype=args.dtype)
    hook = None
    dp = dict(world=world, backend=(tdist.get_backend() if world > 1 else None), ranks=cdist.rank_inventory(model.device))
    if world > 1:
        model.sync_variables(src=0)
        hook = cdist.GradAverager()
        st_g = model._opt_state['g']
        nelem = int(st_g['flat_grad'].numel())
        dp.update(bucket_mb=round(4 * nelem / 1e6, 2), early_fraction=round(float(st_g['split_off']) / nelem, 4),
                  mean="folded into the optimiser kernels (grad_scale = 1/world): the exchange leaves the SUM")
        if os.environ.get("CAPE_DP_PROBE", "1") != "0":
            # every form of the exchange timed on the real bucket (standalone: nothing to hide behind), MAX over ranks; the
            # fastest one that worked carries the timed steps unless CAPE_DP_COLLECTIVE pins the choice
            probe = cdist.probe_collectives(hook, nelem, model.device)
            dp["collective_probe_ms"] = probe
            ok = {k: v for k, v in probe.items() if isinstance(v, float)}
            if ok and "CAPE_DP_COLLECTIVE" not in os.environ:
                hook.mode = min(ok, key=ok.get)
        dp["collective"] = hook.mode
    runner = GraphedTrainStep(model, with_gan=args.gan, grad_hook=hook, use_graph=not args.no_graph)
    runner.load_batch(**synthetic_batch(model, seed=1234 + rank))
    torch.cuda.synchronize()
    runner.capture()

    host = None
    if args.host_inputs:               # pageable numpy arrays, one distinct batch per step
        host = [{k: v.numpy() for k, v in synthetic_batch(model, seed=99 + 1000 * rank + i).items()}
                for i in range(min(args.steps, 8))]

    def one_step(i):
        if host is not None:
            runner.load_batch(**host[i % len(host)])
        runner.step()

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    torch.cuda.synchronize()
    if world > 1:
        tdist.barrier()
    torch.cuda.synchronize()
    elapsed = cdist.max_over_ranks(time.perf_counter() - t0, model.device)

    loss = float(runner.losses['loss_g']) if 'loss_g' in runner.losses else float('nan')
    if world > 1:
        # what the exchange costs the step: the same split step (graphs A1 / A2 / B) with the collectives switched off, AFTER the
        # timed region (the replicas drift apart from here on; nothing below reads the variables)
        hook.enabled = False
        torch.cuda.synchronize()
        tdist.barrier()
        t1 = time.perf_counter()
        for i in range(min(args.steps, 20)):
            one_step(i)
        torch.cuda.synchronize()
        t_off = cdist.max_over_ranks((time.perf_counter() - t1) / min(args.steps, 20), model.device)
        hook.enabled = True
        dp["ms_per_step_exchange_off"] = round(1e3 * t_off, 4)
        dp["exposed_exchange_ms"] = round(1e3 * (elapsed / args.steps - t_off), 4)
        probe_ms = dp.get("collective_probe_ms", {}).get(dp["collective"])
        if isinstance(probe_ms, float):
            dp["hidden_exchange_ms"] = round(max(0.0, probe_ms - dp["exposed_exchange_ms"]), 4)
    roof, table = (None, {})
    if not args.no_roofline:
        roof, table = kernel_roofline(runner)
    if rank != 0:
        return
    ms = 1e3 * elapsed / args.steps
    result = {
        "metric": "meshes/sec fwd+bwd, CAPE-affineconv nz64 @ batch 16, 1/2/4/8 MI355X",
        "value": round(args.batch * world * args.steps / elapsed, 2), "unit": "meshes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": "f32" if args.dtype == 'fp32' else "bf16", "data": "synthetic",
        "config": {"workload": "%s Mesh-CVAE%s: fwd+bwd+clip+momentum update, batch %d per GPU, "
                               "6890-vertex SMPL hierarchy%s"
                               % (args.config.replace("_pose32_clotype32_male", ""), " + mesh-patch discriminator (adversarial step)" if args.gan else "",
                                  args.batch, " (BASELINE configs[2])" if (args.config.startswith("CAPE-affineconv_nz64") and args.dtype == 'fp32') else
                                  " (BASELINE configs[4], one GPU's shard)" if args.config.startswith("CAPE-affineconv_nz64") else
                                  " (BASELINE configs[3], one GPU)" if (args.config.startswith("CAPE_nz18") and args.gan) else ""),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "graph_replay": runner._gA is not None,
                   "inputs": "host numpy per step (PCIe-inclusive)" if args.host_inputs else "resident in HBM",
                   "arithmetic": "bf16 activation storage (BASELINE configs[4] per-GPU shard): bf16 operands, one bf16 MFMA product "
                                 "per multiply-add, fp32 accumulate, fp32 master weights / dense layers / losses / optimiser"
                   if args.dtype == 'bf16' else
                                 "fp32 in/out/accumulate; contractions over >= 256 channels (>= 128 with F >= 128; forward incl. the "
                                 "affine DUAL form, data gradient, weight gradient) as 3 fp16 MFMA products per multiply-add on a "
                                 "two-piece fp16 split (hi + lo, 22 significand bits) of each operand, scaled by a power of two per "
                                 "activation row / weight column (fp32-class accuracy: rms error ~0.7x that of an fp32 FMA chain, "
                                 "tests/test_h2_numerics.py; not bit-identical; inf / NaN operands stay in their rows); shorter "
                                 "contractions as 6 bf16 products on an exact 3-way bf16 split; exact-fp32 MFMA for odd-channel / "
                                 "packed launches and the dense layers",
                   "final_loss_g": loss, "dp": dp},
        "roofline": roof,
        "step_roofline": step_roofline(ms, args.batch, args.gan, args.dtype == 'bf16', cmr=args.config.startswith("CAPE_nz18"))
        if args.config.startswith(("CAPE-affineconv_nz64", "CAPE_nz18")) else None,
    }
    if table:
        result["kernels"] = {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in table.items()}
        # HBM bytes the step moves according to the committed counter pass (per kernel: bytes per dispatch x launches per step),
        # next to the algorithmic bytes of SURVEY 8(d); null while the counter summary belongs to other kernel sources
        # (the launch log names sparse / dense / optimiser launches by operation, the counter summary by kernel symbol: the
        # step total is taken over the counter run's own dispatches, the per-kernel match below is reported beside it)
        cb = [(_pmc_traffic(k), v["launches"]) for k, v in table.items()]
        have = [(b, n) for b, n in cb if b is not None]
        result["counter_bytes_per_step"] = _pmc_step_bytes(roof["kernel"], roof["launches_per_step"]) if roof else None
        result["counter_bytes_per_step_matched"] = dict(bytes=int(sum(b * n for b, n in have)), kernels_with_counters=len(have),
                                                        kernels=len(cb)) if have else None
        if result.get("step_roofline"):
            result["step_roofline"]["counter_bytes_per_step"] = result["counter_bytes_per_step"]
    if world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(batch=args.batch)
    else:
        result["cpu_baseline"] = None
    if world == 1 and not args.no_ab and not args.host_inputs and args.dtype == 'fp32' and os.environ.get("CAPE_GEMM_BF16X6", "1") != "0":
        result["exact_fp32_mfma"] = exact_fp32_run(args)
        # the strict-fp32 figure next to the headline, where a parser of `config` sees it: `value` runs long contractions on a
        # 22-bit two-piece fp16 split (fp32-class accuracy, see `arithmetic`), this is the same step with nothing narrower than
        # the reference's fp32 multiplier anywhere
        result["config"]["exact_fp32_mfma_meshes_per_s"] = result["exact_fp32_mfma"].get("value")
        result["config"]["exact_fp32_mfma_ms_per_step"] = result["exact_fp32_mfma"].get("ms_per_step")
        if result["exact_fp32_mfma"].get("ms_per_step"):
            result["exact_fp32_mfma"]["step_roofline"] = step_roofline(result["exact_fp32_mfma"]["ms_per_step"], args.batch, args.gan)
        if os.environ.get("CAPE_H2", "1") != "0":
            result["bf16x6_split"] = exact_fp32_run(args, dict(CAPE_H2='0'),
                                                    "same step, CAPE_H2=0: every eligible contraction as 6 bf16 products (the arithmetic of round 3)")
    if (world == 1 and not args.no_extras and not args.host_inputs and args.dtype == 'fp32' and not args.global_batch
            and args.config.startswith("CAPE-affineconv_nz64") and not args.no_graph):
        result["extra_configs"], result["scaling_model"] = extra_measurements(args, ms, model)
    print(json.dumps(result))


if __name__ == '__main__':
    main()
