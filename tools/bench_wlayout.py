import sys, torch
sys.path.insert(0, '.')
from cape_amd import ops
from tools.bench_gconv import timeit
dev = torch.device('cuda:0')
N, M, C, F, K = 16, 862, 512, 512, 2
x0 = torch.randn(N, M, C, device=dev); x1 = torch.randn(N, M, C, device=dev)
W = torch.randn(C * K, F, device=dev) * 0.1
Wk = W.view(C, K, F).permute(1, 0, 2).contiguous()          # [K, C, F]
bias = torch.zeros(1, 1, F, device=dev)
y = ops.alloc_act(N, M, F, dev)
fl = 2.0 * N * M * C * K * F
def run(tag, ent, **kw):
    t = timeit(lambda: ops.gconv_fwd(ent, y, **kw))
    print("%-42s %7.1f us %6.1f TF" % (tag, t * 1e6, fl / t / 1e12))
run("interleaved W (row stride K*F), no epilogue", [dict(x=x0, csr=None, w=(W, 0, K * F, 1)), dict(x=x1, csr=None, w=(W, F, K * F, 1))])
run("interleaved W, bias+leaky", [dict(x=x0, csr=None, w=(W, 0, K * F, 1)), dict(x=x1, csr=None, w=(W, F, K * F, 1))], bias=bias, bias_mode=1, act="leaky")
run("per-k contiguous W, no epilogue", [dict(x=x0, csr=None, w=(Wk, 0, F, 1)), dict(x=x1, csr=None, w=(Wk, C * F, F, 1))])
run("per-k contiguous W, bias+leaky", [dict(x=x0, csr=None, w=(Wk, 0, F, 1)), dict(x=x1, csr=None, w=(Wk, C * F, F, 1))], bias=bias, bias_mode=1, act="leaky")
Wp = torch.randn(C * K, F + 32, device=dev) * 0.1
run("interleaved W padded ld=F+32", [dict(x=x0, csr=None, w=(Wp, 0, K * (F + 32), 1)), dict(x=x1, csr=None, w=(Wp, F + 32, K * (F + 32), 1))])
