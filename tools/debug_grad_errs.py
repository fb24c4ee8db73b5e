import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from cape_amd.load_data import load_graph_mtx, load_pack
import test_gpu_model as T
L, D, U, p, Ld, Dd, Ud = load_graph_mtx(None, True)
mesh_ops = dict(L=L, D=D, U=U, p=p, L_d=Ld, D_d=Dd, U_d=Ud, pack=load_pack())
cfg = sys.argv[1] if len(sys.argv) > 1 else 'cmr_nz18'
N = 2
P, twin, model = T._build(cfg, mesh_ops, N, None)
inp = T._inputs(N, P['nz'])
xh, zm, zl, dr, df, ls = T._run_twin(twin, *inp)
model.load_variables(twin.vs.vars)
t = lambda a: torch.tensor(a, dtype=torch.float32, device=model.device)
x, gt, xd, cond, cond_d, clo, clo_d, eps = inp
out = model.forward_losses(t(x), t(cond), t(clo), t(gt), t(xd), t(cond_d), t(clo_d), eps=t(eps))
names = model._g_names
tg = torch.autograd.grad(ls['loss_g'], [twin.params[n] for n in names], allow_unused=True)
hg = torch.autograd.grad(out['loss_g'], [model._vars[n] for n in names], allow_unused=True)
_, twin32 = T._twin(cfg, mesh_ops, N, None, tdtype=torch.float32)
ls32 = T._run_twin(twin32, *inp)[-1]
cg = torch.autograd.grad(ls32['loss_g'], [twin32.params[n] for n in names], allow_unused=True)
rows = []
for n, a, b, c in zip(names, tg, hg, cg):
    if a is None: continue
    rows.append((T.rel_err(b.cpu().numpy(), a.numpy()), T.rel_err(c.numpy(), a.numpy()), n, float(a.abs().max())))
rows.sort(reverse=True)
for r in rows[:25]: print('%.3e  e32=%.3e  %s  max|g|=%.3e' % r)
