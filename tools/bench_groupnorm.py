"""Per-call time of the group-norm operator (forward, backward) at the GraphCMR decoder's shapes:
   python tools/bench_groupnorm.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cape_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
for V, C in [(6890, 96), (6890, 64), (3445, 96), (1723, 160), (862, 288), (431, 288)]:
    x = torch.randn(N, V, C, device=dev, requires_grad=True)
    g, b = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    gy = torch.randn(N, V, C, device=dev)
    G = ops.group_count(N, C)
    def fwd():
        return ops.GroupNormFn.apply(x, g, b, G, 1e-5, 1)
    for _ in range(3):
        fwd().backward(gy)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    R = 20
    for _ in range(R):
        e[0].record(); y = fwd(); e[1].record(); y.backward(gy); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    mb = N * V * C * 4 / 1e6
    print("N %d V %5d C %3d (%.0f MB): forward %6.1f us  backward %6.1f us  (eager, includes launch overhead)" % (N, V, C, mb, 1e3 * tf / R, 1e3 * tb / R))
