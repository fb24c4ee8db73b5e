"""Which aten operators launch the at::native kernels of one (eager) training step?  They are the sub-10-us dispatches of the
step sequence: fills, adds, copies.  Prints operator, calls, device time and -- when the profiler delivers Python stacks (it
does not on every build) -- the first frames inside this repository; together with the ordered dispatch list of
tools/rocpd_step_seq.py (neighbouring kernels) that locates them.
    python tools/attribute_torch_kernels.py [--gan] [--config NAME] [--batch N]"""
import os, sys, argparse, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cape_amd.runtime import GraphedTrainStep

ap = argparse.ArgumentParser()
ap.add_argument('--gan', action='store_true')
ap.add_argument('--config', default=None)
ap.add_argument('--batch', type=int, default=16)
a = ap.parse_args()
model = bench.build_model(a.batch, 0, a.config or 'CAPE-affineconv_nz64_pose32_clotype32_male')
runner = GraphedTrainStep(model, with_gan=a.gan, use_graph=False)
runner.load_batch(**bench.synthetic_batch(model, seed=1))
for _ in range(2):
    runner.step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.key_averages(group_by_stack_n=12):
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.key.startswith("aten::") or dt <= 0:
        continue
    frames = [f for f in (ev.stack or []) if ("cape_amd/" in f or "bench.py" in f) and "attribute_torch_kernels" not in f]
    where = " <- ".join(x.split("/")[-1].strip() for x in frames[:3]) if frames else "?"
    r = rows.setdefault((ev.key, where), [0, 0.0])
    r[0] += ev.count
    r[1] += dt
print("%-28s %5s %9s  %s" % ("op", "calls", "device us", "first repo frame"))
for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print("%-28s %5d %9.1f  %s" % (name, n, us, where))
