import sys, torch
sys.path.insert(0, '.')
import bench
from cape_amd.runtime import GraphedTrainStep
use_graph = 'eager' not in sys.argv
model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
r = GraphedTrainStep(model, with_gan=False, use_graph=use_graph)
r.load_batch(**bench.synthetic_batch(model, 1234))
torch.cuda.synchronize()
r.capture()
vals = []
for i in range(60):
    r.step()
    torch.cuda.synchronize()
    vals.append(float(r.losses['latent']))
print('graph' if use_graph else 'eager', ' '.join('%.0f' % v for v in vals))
