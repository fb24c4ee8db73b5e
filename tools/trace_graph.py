import sys, torch
sys.path.insert(0, '.')
import bench
from cape_amd.runtime import GraphedTrainStep
model = bench.build_model(16, 0, 'CAPE-affineconv_nz64_pose32_clotype32_male')
r = GraphedTrainStep(model, with_gan=False, use_graph=True)
r.load_batch(**bench.synthetic_batch(model, 1234))
torch.cuda.synchronize()
r.capture()
st = model._opt_state['g']
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    r.step()
    torch.cuda.synchronize()
    print(i, {k: float(v) for k, v in r.losses.items()}, 'gnorm', float(torch.linalg.vector_norm(st['flat_grad'])),
          'lr', -float(st['neg_lr']), 'nan params', int(torch.isnan(st['flat']).sum()), 'nan m', int(torch.isnan(st['m']).sum()),
          'nan grad', int(torch.isnan(st['flat_grad']).sum()))
