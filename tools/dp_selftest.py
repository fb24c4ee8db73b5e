"""Single-GPU self-test of the data-parallel step with the REAL collective backend: a one-rank RCCL process group, the
collectives forced on, the split step runner (graph A1 -> async all-reduce -> graph A2 -> all-reduce -> wait -> graph B).
Checks that the sequence runs under HIP-graph replay, that the variables equal the plain single-graph step's, and prints
both step times (one rank: the all-reduce is a device copy, so the difference is the cost of the split itself)."""
import json
import os
import sys
import time

import torch
import torch.distributed as tdist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                                   # noqa: E402
from cape_amd import dist as cdist                             # noqa: E402
from cape_amd.runtime import GraphedTrainStep                  # noqa: E402

import socket                                                  # noqa: E402
with socket.socket() as _s:                                    # a free port: a fixed one collides with a lingering run
    _s.bind(("127.0.0.1", 0))
    _port = _s.getsockname()[1]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
tdist.init_process_group("nccl", rank=0, world_size=1)
res = {}
finals = {}
for name, hook in (("single_graph", None), ("split_rccl_1rank", cdist.GradAverager(always=True))):
    model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
    runner = GraphedTrainStep(model, with_gan=('gan' in sys.argv), grad_hook=hook)
    runner.load_batch(**bench.synthetic_batch(model, 1234))
    torch.cuda.synchronize()
    runner.capture()
    for _ in range(5):
        runner.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        runner.step()
    torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
    finals[name] = model._opt_state['g']['flat'].detach().clone()
# every form of the exchange against the real backend (one rank: each is a device copy, the point is that RCCL accepts the calls)
res["collective_probe_ms"] = cdist.probe_collectives(cdist.GradAverager(always=True), int(finals["single_graph"].numel()), torch.device("cuda:0"))
assert all(isinstance(v, float) for v in res["collective_probe_ms"].values()), res["collective_probe_ms"]
d = (finals["single_graph"] - finals["split_rccl_1rank"]).abs().max().item()
res["max_abs_param_diff"] = d
res["split"] = bool(runner.split)
print(json.dumps(res))
assert d <= 1e-5 * finals["single_graph"].abs().max().item(), d
tdist.destroy_process_group()
