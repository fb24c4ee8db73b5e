"""BASELINE.json configs[1]: single Chebyshev K=6 graph-conv layer fwd+bwd, batch 64 x 6890 x 16 -> 32,
HIP-graph replay timing; prints algorithmic GB/s and TFLOP/s against the rooflines of SURVEY section 8d."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cape_amd import ops
from cape_amd.graph import ConvOperators
from cape_amd.load_data import load_graph_mtx
from tools.bench_gconv import timeit

L, D, U, p, Ld, Dd, Ud = load_graph_mtx(None, True)
dev = torch.device('cuda:0')
N, Cin, Fout, K = 64, 16, 32, 6
dops = ops.DeviceConvOps(ConvOperators(L[0], K), dev)
x = torch.randn(N, 6890, Cin, device=dev, requires_grad=True)
W = (0.1 * torch.randn(Cin * K, Fout, device=dev)).requires_grad_(True)
dy = torch.randn(N, 6890, Fout, device=dev)
def step():
    y = ops.chebyshev5(x, W, dops)
    torch.autograd.grad(y, [x, W], dy)
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
if os.environ.get("CAPE_CONFIG2_EAGER"):          # counter passes: a few eager steps, no graph
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t = timeit(step, iters=5)
else:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    t = timeit(g.replay, iters=50)
nnz = 41328
X, Y, Wt = N * 6890 * Cin, N * 6890 * Fout, Cin * K * Fout
flops = 3 * (2 * N * 6890 * Cin * K * Fout + (K - 1) * 2 * nnz * Cin * N + (K - 2) * 2 * 6890 * Cin * N)
byts = 4 * (3 * X + 2 * Y + 3 * Wt) + 2 * (8 * nnz + 4 * 6891)
print(json.dumps(dict(workload="single Chebyshev K=6 layer fwd+bwd, 64x6890x16->32 (BASELINE configs[1])",
                      form="recurrence on chip (cheb_fused)" if ops.FUSED_RECURRENCE else "materialised K-stack", ms=round(t * 1e3, 4),
                      meshes_per_s=round(N / t, 1), alg_gflop=round(flops / 1e9, 2), alg_mb=round(byts / 1e6, 1),
                      tflops=round(flops / t / 1e12, 2), alg_gbs=round(byts / t / 1e9, 1),
                      roofline_us=dict(fp32_mfma=round(flops / 157.3e12 * 1e6, 1), hbm=round(byts / 8e12 * 1e6, 1)),
                      frac_of_mfma_roofline=round(flops / 157.3e12 / t, 4))))
