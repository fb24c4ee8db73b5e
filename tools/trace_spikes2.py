import sys, torch
sys.path.insert(0, '.')
import bench
from cape_amd.runtime import GraphedTrainStep
model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
r = GraphedTrainStep(model, with_gan=False, use_graph=False)
r.load_batch(**bench.synthetic_batch(model, 1234))
torch.cuda.synchronize()
# same pre-roll as capture(): two fwd/bwd/update passes with lr = 0 (momentum accumulates)
for _ in range(2):
    r._fwd_bwd(); r._update()
vals = []
for i in range(30):
    r.step()
    torch.cuda.synchronize()
    vals.append(float(r.losses['latent']))
print('eager+preroll', ' '.join('%.0f' % v for v in vals))
