import sys, torch
sys.path.insert(0, '.')
import bench
from cape_amd.runtime import GraphedTrainStep
model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
r = GraphedTrainStep(model, with_gan=('gan' in sys.argv), use_graph=False)
r.load_batch(**bench.synthetic_batch(model, 1234))
for i in range(30):
    r.step()
    torch.cuda.synchronize()
    st = model._opt_state['g']
    print(i, {k: float(v) for k, v in r.losses.items()}, 'gnorm', float(torch.linalg.vector_norm(st['flat_grad'])), 'lr', -float(st['neg_lr']))
