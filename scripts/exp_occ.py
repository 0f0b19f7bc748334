"""Experiment: forward kernel rate vs LDS footprint (does a second workgroup fit per CU?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from nerfies_amd import models
dev = torch.device('cuda:0')
for F in (8, 6, 4):
  class Cfg(bench.Cfg):
    num_nerf_point_freqs = F
  model, fp = models.construct_nerf(0, Cfg, 1024, [0,1,2,3],[0,1],[0,1,2,3], 0.0206, 0.826, device=dev)
  batch = bench.synthetic_batch(1024, 100, dev)
  for train in (False, True):
    for _ in range(3): model.apply({'params': fp}, batch, {}, rngs={'coarse': 1, 'fine': 2}, train=train)
    model.profile_enable(True)
    for _ in range(10): model.apply({'params': fp}, batch, {}, rngs={'coarse': 1, 'fine': 2}, train=train)
    torch.cuda.synchronize()
    pr = model.profile_read(); model.profile_enable(False)
    print('F=%d' % F, 'train' if train else 'eval ', {e['name']: round(e['ms']/e['launches'], 4) for e in pr if 'mlp' in e['name']},
          {e['name']: round(e['flops_per_launch']/(e['ms']/e['launches']*1e-3)/1e12,1) for e in pr if 'mlp' in e['name']})
