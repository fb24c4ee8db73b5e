"""The data-parallel path with the REAL model on a GPU box (the CPU test tests/test_dist_cpu.py covers
the exchange logic on a stand-in network).  One GPU is enough: two ranks share device 0 and exchange over gloo; the
two-phase backward + early/late bucket exchange + HIP-graph replay are exactly what `bench.py --gpus N` runs over RCCL."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    return env


def _two_rank_equals_global(gan, tmp_path, backend, env, port):
    B, steps = 2, 2
    out = str(tmp_path / "dp.npz")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "dp_equiv_worker.py"), out, str(B), str(steps), "1" if gan else "0",
           backend]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    dp = np.load(out)
    assert int(dp["split"]) == 1                       # the two-phase backward was the path taken
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dp_equiv_worker as W
    model = W.build(2 * B, 'cuda:0')
    init = {g: model._opt_state[g]['flat'].detach().clone() for g in ('g', 'd')}
    W.run(model, W.global_batch(2 * B, int(model.nz)), steps, gan, None)
    for grp in (('g', 'd') if gan else ('g',)):
        ref = model._opt_state[grp]['flat'].detach().cpu().numpy()
        got = dp["flat_" + grp]
        moved = np.abs(ref - init[grp].cpu().numpy()).max()
        err = np.abs(got - ref).max()
        print(grp, "largest update %.3e, dp-vs-global difference %.3e" % (moved, err))
        assert moved > 0 and err <= 2e-3 * moved + 1e-7, (grp, err, moved)


@pytest.mark.parametrize("gan", [False, True], ids=["cvae", "adversarial"])
def test_two_rank_step_equals_global_batch_step(gan, tmp_path):
    """2 ranks x B meshes (sum of the rank gradients, 1 / world and the clip inside the optimiser kernels, replicas
    bit-identical) must reproduce the single-process step on the 2B-mesh global batch: every loss term is a batch mean, so the
    averaged gradient IS the global gradient.  Tolerance: fp32 summation order only (kernel tile selection differs between
    batch B and 2B).  Two ranks on device 0 over gloo."""
    _two_rank_equals_global(gan, tmp_path, "gloo", _env(CAPE_FORCE_DEVICE="0"), 29533)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two HIP devices: RCCL over xGMI between them")
@pytest.mark.parametrize("mode", ["allreduce", "rsag", "direct"])
def test_two_rank_rccl_step_equals_global_batch_step(mode, tmp_path):
    """The same equivalence over the real transport -- one process per GPU, backend "nccl" (RCCL), every form of the exchange
    (cape_amd.dist.MODES) -- wherever two devices are visible; skipped on the one-GPU boxes of the test tier."""
    env = _env(CAPE_DP_COLLECTIVE=mode)
    env.pop("CAPE_FORCE_DEVICE", None)
    _two_rank_equals_global(False, tmp_path, "nccl", env, 29541)


def test_split_step_with_one_rank_rccl_group():
    """The split runner against the real collective backend (a one-rank RCCL group, collectives forced on): graph A1 ->
    async all-reduce -> graph A2 -> all-reduce -> wait -> graph B must replay and equal the single-graph step."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_selftest.py")], env=_env(), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert '"split": true' in r.stdout.decode()


def test_bench_gpus2_without_torchrun_launches_two_ranks():
    """`python bench.py --gpus 2` invoked the way the driver invokes `--gpus 1` (no torchrun, no WORLD_SIZE) must start the
    two ranks itself and report n_gpus 2 (harness valid for world_size 1..8); here both share device 0 over gloo."""
    import json
    env = _env(CAPE_DIST_BACKEND="gloo", CAPE_FORCE_DEVICE="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--no-cpu-baseline", "--no-extras", "--no-ab", "--no-roofline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 4
    # the line proves what ran: both ranks with their device and backend, every exchange form timed (or its refusal named),
    # the form used, and what the exchange cost the step
    dp = out["config"]["dp"]
    assert dp["world"] == 2 and dp["backend"] == "gloo" and sorted(r["rank"] for r in dp["ranks"]) == [0, 1]
    assert all(r["device"] == 0 and r["backend"] == "gloo" for r in dp["ranks"]) and len({r["pid"] for r in dp["ranks"]}) == 2
    assert set(dp["collective_probe_ms"]) == {"allreduce", "rsag", "direct"} and isinstance(dp["collective_probe_ms"]["allreduce"], float)
    assert dp["collective"] in dp["collective_probe_ms"] and "exposed_exchange_ms" in dp and dp["bucket_mb"] > 60


def test_bench_gpus_refuses_more_ranks_than_devices():
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CAPE_FORCE_DEVICE"):
        env.pop(k, None)
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0 and b"HIP device(s) visible" in r.stderr
