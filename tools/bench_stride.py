import sys, torch
sys.path.insert(0, '.')
from cape_amd import ops
from tools.bench_gconv import timeit
dev = torch.device('cuda:0')
N, M, Cin, Fout = 16, 862, 1024, 512
for pad in (0, 4, 32, 64):
    buf = torch.randn(N, M, Cin + pad, device=dev)
    x = buf[:, :, :Cin]
    W = torch.randn(Cin, Fout, device=dev) * 0.1
    y = ops.alloc_act(N, M, Fout, dev)
    ent = [dict(x=x, csr=None, w=(W, 0, Fout, 1))]
    t = timeit(lambda: ops.gconv_fwd(ent, y))
    print("ld=%d  %.1f us  %.1f TF" % (Cin + pad, t * 1e6, 2.0 * N * M * Cin * Fout / t / 1e12))
