"""Per-launch table of the gather-GEMM launches of one training step (HIP-event timed, eager)."""
import sys, torch
sys.path.insert(0, '.')
import bench
from cape_amd import ops
from cape_amd.runtime import GraphedTrainStep
cfg = next((a.split('=', 1)[1] for a in sys.argv if a.startswith('config=')), 'CAPE-affineconv_nz64_pose32_clotype32_male')
batch = int(next((a.split('=', 1)[1] for a in sys.argv if a.startswith('batch=')), 16))
model = bench.build_model(batch, 0, cfg)
r = GraphedTrainStep(model, with_gan=('gan' in sys.argv), use_graph=False)
r.load_batch(**bench.synthetic_batch(model, 1234))
for _ in range(2):
    r._fwd_bwd()
orig = ops.gconv_fwd
def wrapped(entries, y, **kw):
    N, Mo, F = y.shape
    desc = "Mo%5d F%4d src[%s]%s" % (Mo, F, ",".join("%d%s" % (int(e.get("C", e["x"].shape[2])), "g" if (e.get("csr") is not None and not e["csr"].identity) else "") for e in entries), " dual" if any(e.get("w2") is not None for e in entries) else "")
    ops._last_desc = desc
    if "rep" in sys.argv and ops.LAUNCH_LOG is not None:
        for _ in range(2):
            orig(entries, y, **kw)
        ops._last_desc = desc + " (3rd rep)"
    return orig(entries, y, **kw)
ops.gconv_fwd = wrapped
orig_dw = ops.gconv_dw
def wrapped_dw(entries, dz, accumulate=False, dz2=None, **kw):
    N, Mo, F = dz.shape
    if ops.LAUNCH_LOG is None:
        return orig_dw(entries, dz, accumulate, dz2, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig_dw(entries, dz, accumulate, dz2, **kw)
    e1.record()
    Cs = [int(e.get("C", e["x"].shape[2])) for e in entries]
    fl = sum(2.0 * N * Mo * c * F for c in Cs)
    by = 4.0 * N * Mo * (sum(Cs) + F)
    ops.LAUNCH_LOG.append(("dW(+reduce)", fl, by, e0, e1, "Mo%5d F%4d src[%s]" % (Mo, F, ",".join(map(str, Cs)))))
ops.gconv_dw = wrapped_dw
log = []
ops.LAUNCH_LOG = []
torch.cuda.synchronize()
# monkeypatch _log_launch to keep descriptions
orig_log = ops._log_launch
def ll(name, flops, byts, fn):
    out = orig_log(name, flops, byts, fn)
    ops.LAUNCH_LOG[-1] = ops.LAUNCH_LOG[-1] + (getattr(ops, "_last_desc", ""),)
    return out
ops._log_launch = ll
r._fwd_bwd()
torch.cuda.synchronize()
tot = 0
for name, fl, by, e0, e1, desc in ops.LAUNCH_LOG:
    t = e0.elapsed_time(e1) * 1e3
    tot += t
    print("%-40s %-36s %8.1f us %6.1f TF %7.0f GB/s" % (name.replace("gconv_fwd_kernel", "fwd"), desc, t, fl / t / 1e6, by / t / 1e3))
print("total %.1f us over %d launches" % (tot, len(ops.LAUNCH_LOG)))
