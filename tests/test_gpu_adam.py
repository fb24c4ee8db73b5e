"""tf.train.AdamOptimizer on the device (reference lib/models.py:447-449, selectable through config_parser.py:41): the fused
clip + Adam kernel on the flat buckets (csrc/optim.hip ``cape_flat_adam_update``) against the TensorFlow formula in float64

    t <- t + 1;  lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    m <- beta1 m + (1 - beta1) g;  v <- beta2 v + (1 - beta2) g^2;  var <- var - lr_t m / (sqrt(v) + epsilon)

behind tf.clip_by_global_norm(5.0) (:461) with the dense kernels' regulariser gradient folded in, and the whole training step
with ``optimizer='adam'`` -- eager, and captured into a HIP graph (the step count is a device counter: replays must apply the
bias correction of steps 1, 2, 3, ... and not that of the captured step)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B1, B2, EPS = 0.9, 0.999, 1e-8
# TensorFlow holds beta1 / beta2 as float32 tensors and forms (1 - beta) in float32 (1 - 0.999f = 0.00100005): the float64
# reference below uses the same rounded constants, as the kernel does
B1F, B2F = float(np.float32(B1)), float(np.float32(B2))
C1F, C2F = float(np.float32(1) - np.float32(B1)), float(np.float32(1) - np.float32(B2))


def _adam_reference(w, g, m, v, t, lr, clip, ranges, coef):
    """One update in float64 (the formula of the docstring); returns the new (w, m, v)."""
    ge = g.copy()
    for b, e in ranges:
        ge[b:e] += coef * w[b:e]
    norm = np.sqrt((ge * ge).sum())
    ge *= clip / max(norm, clip)
    m = B1F * m + C1F * ge
    v = B2F * v + C2F * ge * ge
    lr_t = lr * np.sqrt(1 - B2F ** t) / (1 - B1F ** t)
    return w - lr_t * m / (np.sqrt(v) + EPS), m, v


@pytest.mark.parametrize("gscale,ranges", [(1.0, []), (1e-3, [(1024, 5120), (40000, 65536)]), (30.0, [(0, 4096)])],
                         ids=["clipped", "unclipped_reg", "clipped_reg"])
def test_adam_kernel_matches_tf_formula(gscale, ranges):
    from cape_amd import ops
    dev = torch.device("cuda:0")
    n = 1 << 17
    rng = np.random.default_rng(3)
    w = rng.standard_normal(n) * 0.1
    w64, m64, v64 = w.copy(), np.zeros(n), np.zeros(n)
    coef, lr, clip = 0.25, 3e-3, 5.0
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=dev)
    dw, dm, dv = f32(w), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    w64 = dw.cpu().numpy().astype(np.float64)
    state = torch.zeros(2, device=dev, dtype=torch.int32)
    sumsq = torch.zeros((), device=dev)
    neg_lr = torch.full((), -lr, device=dev)
    ws = ops.flat_workspace(dev)
    worst = 0.0
    for step in range(1, 7):
        g = rng.standard_normal(n) * gscale * (1.0 + 0.3 * step)
        g[::97] = 0.0                                           # exact zeros: sqrt(v) + eps carries the division
        dg = f32(g)
        g64 = dg.cpu().numpy().astype(np.float64)
        ops.flat_gradnorm(dg, dw, ranges, coef, sumsq, ws)
        ops.flat_adam_update(dw, dg, dm, dv, B1, B2, EPS, clip, sumsq, neg_lr, state, ranges, coef)
        w_new, m64, v64 = _adam_reference(w64, g64, m64, v64, step, lr, clip, ranges, coef)
        upd = np.abs(w_new - w64).max()
        err_w = np.abs(dw.cpu().numpy() - w_new).max()
        # the device holds w, m, v in fp32: one rounding of w (6e-8 |w|) plus the update's own fp32 evaluation (1e-5 of it)
        assert err_w <= 1e-5 * upd + 1.2e-7 * np.abs(w_new).max(), (step, err_w, upd)
        assert np.abs(dm.cpu().numpy() - m64).max() <= 2e-6 * np.abs(m64).max()
        assert np.abs(dv.cpu().numpy() - v64).max() <= 2e-6 * np.abs(v64).max()
        worst = max(worst, err_w / upd)
        w64 = dw.cpu().numpy().astype(np.float64)               # follow the device's fp32 state (no drift accumulation)
        m64, v64 = dm.cpu().numpy().astype(np.float64), dv.cpu().numpy().astype(np.float64)
        assert state.cpu().tolist() == [step, 0]
    print("adam kernel vs float64 TF formula: worst error / update = %.2e over 6 steps" % worst)


def test_adam_kernel_rejects_bad_arguments():
    from cape_amd import ops
    dev = torch.device("cuda:0")
    z = torch.zeros(64, device=dev)
    state = torch.zeros(2, device=dev, dtype=torch.int32)
    s = torch.zeros((), device=dev)
    with pytest.raises(RuntimeError):
        ops.flat_adam_update(z, z, z, z, 1.0, B2, EPS, 5.0, s, s, state, [], 0.0)       # beta1 = 1: division by zero in lr_t
    with pytest.raises(RuntimeError):
        ops.flat_adam_update(z, z, z, z, B1, B2, 0.0, 5.0, s, s, state, [], 0.0)       # epsilon must be positive
    with pytest.raises(RuntimeError):
        ops.flat_adam_update(z[1:], z[1:], z[1:], z[1:], B1, B2, EPS, 5.0, s, s, state, [], 0.0)   # misaligned bucket


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hip_graph"])
def test_adam_training_steps_match_manual_adam(graph, mesh_ops):
    """Three training steps of the affine model with optimizer='adam' (regularised dense kernels, clip active) through the
    step runner, against a manual float64 Adam applied to the device's own gradient buckets of each step.  Under graph
    capture the SAME graph is replayed three times: the device step counter must advance (bias corrections of t = 1, 2, 3)."""
    from test_gpu_model import _build, _inputs
    from cape_amd.runtime import GraphedTrainStep
    N = 2
    P, twin, model = _build("affine_nz64", mesh_ops, N, dict(regularization=0.5, lr_warmup=False, decay_steps=1000, optimizer='adam', lr=1e-4))
    assert model.optimizer == 'adam'
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    runner = GraphedTrainStep(model, with_gan=False, use_graph=graph)
    assert runner.use_graph == graph
    runner.load_batch(data_g=x, cond_g=cond, cond2_g=clo, gt=gt, data_d=xd, cond_d=cond_d, cond2_d=clo_d, eps=eps)
    model.set_learning_rates(('g',))
    if graph:
        runner.capture(warmup=1, preserve_state=True)
        assert model.adam_steps('g') == 0                       # the warm-up pass' update was rolled back, counter included
    st = model._opt_state['g']
    ranges = None
    w64 = st['flat'].detach().cpu().numpy().astype(np.float64)
    m64, v64 = np.zeros_like(w64), np.zeros_like(w64)
    lr = model._lr_at(model.lr_g, 0)
    coef = model.regularization ** 2
    for step in range(1, 4):
        runner.step()
        torch.cuda.synchronize()
        if ranges is None:
            ranges = model._reg_ranges()
            assert ranges, "the affine model regularises its dense kernels"
        g64 = st['flat_grad'].detach().cpu().numpy().astype(np.float64)
        w_new, m64, v64 = _adam_reference(w64, g64, m64, v64, step, lr, 5.0, ranges, coef)
        got = st['flat'].detach().cpu().numpy().astype(np.float64)
        assert np.isfinite(g64).all() and np.isfinite(got).all(), step
        upd = np.abs(w_new - w64).max()
        err = np.abs(got - w_new).max()
        assert upd > 0 and err <= 2e-5 * upd + 1.2e-7 * np.abs(w_new).max(), (step, err, upd)
        assert model.adam_steps('g') == step
        w64 = got
        m64 = st['m'].detach().cpu().numpy().astype(np.float64)
        v64 = st['v'].detach().cpu().numpy().astype(np.float64)
    assert np.isfinite(float(runner.losses['loss_g']))


def test_adam_state_round_trips_through_checkpoints(tmp_path, mesh_ops):
    from test_gpu_model import _build, _inputs
    N = 2
    P, twin, model = _build("affine_nz64", mesh_ops, N, dict(lr_warmup=False, decay_steps=1000, optimizer='adam', lr=1e-4))
    model.project_dir = str(tmp_path)
    x, gt, xd, cond, cond_d, clo, clo_d, eps = _inputs(N, P["nz"])
    t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
    args = (t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d))
    for _ in range(2):
        model.train_step(*args, eps=t(eps))
    assert model.adam_steps('g') == 2 and model.adam_steps('d') == 2
    fn = model.save_checkpoint(2)
    v_before = model._opt_state['g']['v'].clone()
    model._set_adam_steps('g', 0)
    model._opt_state['g']['v'].zero_()
    model.restore(fn)
    assert model.adam_steps('g') == 2 and model.adam_steps('d') == 2
    assert bool(torch.isfinite(v_before).all()) and torch.equal(model._opt_state['g']['v'], v_before)
