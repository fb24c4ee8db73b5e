import sys, torch, numpy as np
sys.path.insert(0, '.')
import bench
from cape_amd import ops
from cape_amd.runtime import GraphedTrainStep
model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
r = GraphedTrainStep(model, with_gan=False, use_graph=False)
r.load_batch(**bench.synthetic_batch(model, 1234))
for _ in range(2):
    r._fwd_bwd(); r._update()
b = r.buf
def lat(mode):
    ops.MODE = mode
    with torch.no_grad():
        out = model.forward_losses(b['data_g'], b['cond_g'], b['cond2_g'], b['gt'], eps=b['eps'], with_gan=False)
    ops.MODE = 'twopass'
    return float(out['latent']), float(out['z_logvar'].max()), float(out['z_logvar'].min()), float(out['z_mean'].abs().max())
for i in range(22):
    if i >= 17:
        print(i, 'twopass', lat('twopass'), 'fused', lat('fused'))
    r.step()
    torch.cuda.synchronize()
    print(i, 'step latent', float(r.losses['latent']))
# dump step-19 state for the CPU oracle
