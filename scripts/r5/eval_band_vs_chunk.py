#!/usr/bin/env python
"""What one of 8 ranks does for a 960 x 540 frame under the two tile_parallel modes of evaluation.render_image, timed on ONE GPU (no
collective: the launches only).  'band': the rank renders its contiguous band of 8 whole 8192-ray chunks.  'chunk' (the reference's
order, evaluation.py:61-92): every one of the frame's 64 chunks is cut 8 ways, i.e. the rank replays 64 launches of 1024 rays.
Same rays, same hipGraph renderer (GraphedChunkRenderer), SE3 warp on as eval.py renders; fp32 and the bf16 mode.
  python scripts/r5/eval_band_vs_chunk.py > gpurun_out/r5n/eval_band_vs_chunk.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from nerfies_amd import evaluation, models, training
dev = torch.device('cuda', 0)
H, W, WORLD, CHUNK = 540, 960, 8, 8192
n_frame = H * W
n_band = -(-(-(-n_frame // CHUNK)) // WORLD) * CHUNK           # 8 chunks of 8192 = 65536 rays per rank
out = {'frame': [W, H], 'world': WORLD, 'rays_per_rank': n_band, 'modes': {}}
for bf16 in (False, True):
  model, fp = models.construct_nerf(0, bench.CfgEvalWarp, CHUNK, list(range(256)), [0, 1], list(range(256)), 0.0206, 0.826, device=dev)
  state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=8.0)
  res = {}
  for name, chunk in (('band', CHUNK), ('chunk', CHUNK // WORLD)):
    rays = {k: v.reshape(n_band // 64, 64, 3) for k, v in bench.synthetic_batch(n_band, 7, dev).items() if k in ('origins', 'directions')}
    rays['metadata'] = {'warp': torch.full((n_band // 64, 64, 1), 3, dtype=torch.int32, device=dev)}
    fn = evaluation.GraphedChunkRenderer(model, bf16=bf16)
    render = lambda: evaluation.render_image(state, rays, fn, chunk=chunk)
    render(); render()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
      render()
    torch.cuda.synchronize()
    res[name] = {'launch_rays': chunk, 'launches': n_band // chunk, 'ms': 1e3 * (time.perf_counter() - t0) / 3}
  res['chunk_over_band'] = res['chunk']['ms'] / res['band']['ms']
  out['modes']['bf16' if bf16 else 'f32'] = res
  print('bf16' if bf16 else 'f32', res, file=sys.stderr, flush=True)
out['csrc_sha16'] = bench.kernel_source_sha()
print(json.dumps(out))
