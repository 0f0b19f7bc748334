"""RCCL on the one GPU a box has: a world-size-1 `nccl` process group loads librccl and pushes the step's fused
[grad | stats] all-reduce (training.py:266-267), render_image's packed all_gather_into_tensor (eval.py:339) and bench.py's
all-reduce timing through it, on the library's stream, so the first multi-GPU run only adds ranks to a path that has
already executed.  A one-rank sum / gather is the identity: checked BITWISE on the step's own fused buffer and on a rendered
chunk; the training run as a whole is compared with a run without torch.distributed to float32 summation order (two runs of
the same step differ in the order of their float atomics, with or without a collective in between).
Runs in a child process: a process group cannot be re-initialised inside the pytest process."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.environ['NRF_ROOT']); sys.path.insert(0, os.path.join(os.environ['NRF_ROOT'], 'tests'))
import torch, torch.distributed as dist
import helpers as H
from oracle import nerfies_oracle as O
from nerfies_amd import training, evaluation
use_dist = sys.argv[1] == '1'
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
if use_dist:
  os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = sys.argv[2]
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
spec = O.ModelSpec(num_coarse_samples=32, num_fine_samples=32, num_nerf_point_freqs=6, use_stratified_sampling=True, use_warp=True,
                   num_warp_freqs=4, use_camera_metadata=True)
oparams = O.init_params(spec, seed=3, trained_like=True, dtype=torch.float32)
B = 96
batch = H.gpu_batch(O.synthetic_batch(B, seed=4, dtype=torch.float32))
batch['background_points'] = (torch.rand(512, 3, generator=torch.Generator().manual_seed(1)) - 0.5).to(dev)
model, fp = H.gpu_model(spec, oparams, B)
state = training.TrainState(optimizer=training.Optimizer(fp), warp_alpha=2.5)
sp = training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, background_loss_weight=1.0)
# the gradient at the initial parameters (before Adam's normalised updates amplify rounding-level differences between two runs)
grad0, _ = model.loss_and_grad(state.optimizer.target, batch, warp_extra=state.warp_extra, rngs={'coarse': 5, 'fine': 6},
                               elastic={'weight': 0.01, 'reduce_method': 'weight'})
grad_abs0 = float(grad0.double().abs().sum())
# Every step starts from parameters BOTH runs can reproduce exactly (the initial ones, then seeded perturbations of them written
# over the Adam-updated values): a step's reported losses are functions of the parameters it starts from, so the two processes are
# compared step by step at float-atomic-order tolerance instead of along two trajectories that Adam's normalised update drives apart
base = fp.flat.clone()
pert = torch.randn(base.numel(), generator=torch.Generator().manual_seed(11)).to(dev)
def resync(k):
  fp.flat.copy_(base * (1.0 + 0.01 * k * pert))
key, losses = 7, []
for k in range(3):
  if k:
    resync(k)
  state, stats, key = training.train_step(model, key, state, batch, sp, use_elastic_loss=True, elastic_reduce_method='weight',
                                          use_background_loss=True)
  losses.append([float(stats['coarse']['loss/total']), float(stats['fine']['loss/total']), float(stats['background_loss'])])
# render_image: 5 x 7 = 35 rays in chunks of 16 (ragged tail), through the packed gather
rays = {k: v[:35].reshape(5, 7, -1) for k, v in batch.items() if k in ('origins', 'directions')}
rays['metadata'] = {k: v[:35].reshape(5, 7, 1) for k, v in batch['metadata'].items()}
fn = lambda k0, k1, params, r, extra: model.apply({'params': params}, r, extra)
img = evaluation.render_image(state, rays, fn, chunk=16)
out = {'params_sum': float(fp.flat.double().sum()), 'params_abs': float(fp.flat.double().abs().sum()),
       'params_bits': int(fp.flat.view(torch.int32).to(torch.int64).sum().item()), 'losses': losses,
       'rgb_bits': int(img['rgb'].contiguous().view(torch.int32).to(torch.int64).sum().item()), 'rgb_shape': list(img['rgb'].shape),
       'rccl': None}
out['grad_abs'] = grad_abs0
# the whole step from ONE hipGraph: with the process group its all-reduce (RCCL, a stream operation) is captured with the rest
gstep = training.GraphedTrainStep(model, state, batch, sp, use_elastic_loss=True, elastic_reduce_method='weight', use_background_loss=True)
glosses = []
for k in (21, 22, 23):
  resync(k - 17)
  st = gstep(k)
  glosses.append([float(st['coarse']['loss/total']), float(st['fine']['loss/total']), float(st['background_loss'])])
out['graph_losses'] = glosses
out['graph_split'] = bool(gstep.split)
out['graph_params_abs'] = float(fp.flat.double().abs().sum())
if use_dist:
  fused = state.optimizer._gs.clone()
  before = fused.clone()
  dist.all_reduce(fused)                                    # the step's own collective, on the step's own buffer
  out['allreduce_identity'] = bool(torch.equal(fused, before))
  packed = img['rgb'].reshape(-1, 3).contiguous()
  gathered = torch.empty_like(packed)
  dist.all_gather_into_tensor(gathered, packed)             # render_image's collective
  out['allgather_identity'] = bool(torch.equal(gathered, packed))
  v = torch.cuda.nccl.version()
  out['rccl'] = list(v) if isinstance(v, (tuple, list)) else v
  out['backend'] = dist.get_backend()
  dist.destroy_process_group()
print('RESULT ' + json.dumps(out))
'''


def _run(use_dist, port):
  env = dict(os.environ, NRF_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0')
  r = subprocess.run([sys.executable, '-c', CHILD, '1' if use_dist else '0', str(port)], env=env, capture_output=True, text=True,
                     timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][-1]
  return json.loads(line[7:])


def test_train_step_and_render_through_a_one_rank_rccl_communicator():
  plain = _run(False, 0)
  rccl = _run(True, 29517)
  assert rccl['backend'] == 'nccl' and rccl['rccl'], rccl
  assert rccl['allreduce_identity'] and rccl['allgather_identity']        # bitwise: a one-rank sum / gather is the identity
  assert plain['rgb_shape'] == rccl['rgb_shape'] == [5, 7, 3]
  # the same three steps in two separate processes, every step from parameters both runs reproduce exactly (the child re-syncs them:
  # round 5 compared two free-running Adam trajectories and had to allow 5-10 %): the runs differ by the order of their float
  # atomics only
  for k, (ra, rb) in enumerate(zip(plain['losses'], rccl['losses'])):
    for a, b in zip(ra, rb):
      assert abs(a - b) <= 1e-6 + 5e-4 * abs(a), (k, plain['losses'], rccl['losses'])
  assert abs(plain['grad_abs'] - rccl['grad_abs']) <= 1e-4 * plain['grad_abs']     # at the initial parameters
  assert abs(plain['params_abs'] - rccl['params_abs']) <= 2e-5 * plain['params_abs']
  # three more steps replayed from the captured step: ONE graph in both runs (the RCCL all-reduce is inside it), re-synced likewise
  assert plain['graph_split'] is False and rccl['graph_split'] is False
  for a, b in zip(sum(plain['graph_losses'], []), sum(rccl['graph_losses'], [])):     # steps 4-6 of the two trajectories
    assert abs(a - b) <= 1e-6 + 5e-4 * abs(a), (plain['graph_losses'], rccl['graph_losses'])
  # ONE Adam step behind re-synced parameters (the moments carry six steps of float-atomic ordering noise: lr-sized moves of the
  # entries whose gradient is rounding-level)
  assert abs(plain['graph_params_abs'] - rccl['graph_params_abs']) <= 2e-5 * plain['graph_params_abs']


def test_bench_line_through_rccl(tmp_path):
  """bench.py BENCH_FORCE_DIST=1: the headline step, the burn-in agreement and the grad_allreduce_us measurement on a one-rank
  RCCL communicator; the line reports the RCCL version."""
  env = dict(os.environ, BENCH_FORCE_DIST='1', MASTER_PORT='29519', HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('NCCL_DEBUG', 'NCCL_DEBUG_FILE', 'NCCL_DEBUG_SUBSYS'):   # an inherited console log would be left alone by the bench (no capture)
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--burn-in-s', '0',
                      '--no-cpu-baseline', '--rays-per-gpu', '128'], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
  assert d['dist_backend'] == 'nccl' and d['rccl_ranks'] == 1 and d['rccl_version'] and d['grad_allreduce_us'] > 0
  assert d['config']['rays_per_gpu'] == 128 and d['value'] > 0
  # the pre-flight record: what RCCL logged while this (one-rank) communicator came up
  pf = d['rccl_preflight']
  assert pf and pf['log_lines'] > 0 and pf['init_complete'] and 1 in pf['nranks_seen'], pf


def test_bench_gpus_2_launches_itself_and_checks_itself():
  """`python bench.py --gpus 2 ...` exactly as the driver types it, NOT under torch.distributed.run (round 4: rc 1).  On the
  one-GPU lease the two ranks share cuda:0 over gloo (the line says `oversubscribed`); on a box with >= 2 GPUs the same command
  runs RCCL.  One line carries both curves: weak (1024 rays per GPU) and the nested strong-scaling record (512 rays per GPU,
  eager + hipGraph), each with bit-identical replicas."""
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BENCH_FORCE_DIST', 'BENCH_DIST_BACKEND', 'BENCH_SAME_DEVICE'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--burn-in-s', '0'],
                     env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, lines                      # rank 0 prints ONE JSON line
  d = json.loads(lines[0])
  import torch
  assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['replica_param_checksums_agree'] is True
  assert d['config']['global_batch'] == 2048 and d['scaling'] == 'weak' and d['value'] > 0
  # the statistics a step returns are pmean'ed over the ranks (training.py:267): every rank reports the same mean loss
  assert len(d['per_rank_final_loss_fine']) == 2 and d['per_rank_final_loss_fine'][0] == d['per_rank_final_loss_fine'][1] > 0
  assert d['grad_allreduce_us'] > 0 and d['grad_allreduce_exposed_us'] is not None
  s = d['strong_scaling']
  assert s['global_batch'] == 1024 and s['rays_per_gpu'] == 512
  for k in ('eager', 'graph'):
    assert s[k]['value'] > 0 and s[k]['replica_param_checksums_agree'] is True
  if torch.cuda.device_count() < 2:
    assert d['dist_backend'] == 'gloo' and '2 ranks on 1' in d['oversubscribed'] and 'OVERSUBSCRIBED' in d['config']['workload']
  else:
    assert d['dist_backend'] == 'nccl' and d['oversubscribed'] is None and d['rccl_version']


def test_default_bench_line_carries_its_certificates():
  """`python bench.py` as the driver runs it (fewer steps, no CPU baseline here): the ONE line carries the config-E parity record
  (HIP vs float64 oracle, within the north star's 1e-3), the secondary lines of the other BASELINE configs and the sustained run,
  top-level and as a compact copy inside `config` (the driver's record keeps `config` whole)."""
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'BENCH_FORCE_DIST', 'BENCH_DIST_BACKEND', 'BENCH_SAME_DEVICE'):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '5', '--warmup', '2', '--burn-in-s', '0.5', '--no-cpu-baseline',
                      '--sustained-s', '1.0'], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
  d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
  ep = d['eval_parity']
  assert ep['rays'] >= 256 and ep['pass'] and ep['max_abs_rgb'] <= 1e-3 and ep['max_abs_depth'] <= 1e-3 and ep['psnr_vs_oracle_db'] > 60, ep
  modes = {(x['mode'], x['dtype']): x for x in d['secondary']}
  x3 = ep['bf16x3']   # round 6: the split-bf16 (float32-emulating) chains on the same rays, same oracle
  assert x3['pass'] and x3['max_abs_rgb'] <= 2e-4 and x3['max_abs_rgb_vs_f32_path'] <= 2e-4 and x3['psnr_vs_oracle_db'] > 60, x3
  assert set(modes) == {('vrig', 'f32'), ('fullhd', 'bf16'), ('eval_warp', 'f32'), ('eval_x3', 'bf16x3 (fp32-emulating)'),
                        ('eval_warp_x3', 'bf16x3 (fp32-emulating)')}, d['secondary']
  assert modes[('eval_x3', 'bf16x3 (fp32-emulating)')]['value'] > 1.5 * 3.0e5   # > 1.5 x what the float32 chains reach on this chunk (308 k rays/s)
  for x in d['secondary']:
    assert 'error' not in x and x['value'] > 0 and 0 < x['roofline']['frac'] < 1, x
  assert d['sustained']['seconds'] >= 0.97 and 0.8 < d['sustained']['vs_headline'] < 1.25, d['sustained']
  c = d['config']['certified_in_this_run']
  assert c['eval_parity']['pass'] and len(c['secondary']) == 5 and c['sustained']['value'] > 0
  assert list(d)[-3:] == ['eval_parity', 'secondary', 'sustained']     # the tail of the line
