"""Training driver: the command-line surface of google/nerfies' train.py (train.py:43-51 flags, :100-321 main) on
the MI355X path.

  python train.py --base_folder EXP --data_dir CAPTURE --gin_configs configs/my.gin [--gin_bindings "A.b = 1" ...]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...     (one process per GPU)

One process drives one GPU; ranks hold the same HBM-resident ray table and take their own slice of each global
batch; gradients meet in one all-reduce (RCCL) inside training.train_step.  Tensorboard is replaced by a JSON-lines
scalar log under <exp>/summaries/train."""
import argparse
import dataclasses
import os
import sys

import torch
import torch.distributed as dist

from nerfies_amd import checkpoints, configs, datasets, models, schedules, training, utils
from nerfies_amd import gin_lite as gin


def parse_flags(argv=None):
  p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  p.add_argument('--base_folder', required=True, help='where to store ckpts and logs')
  p.add_argument('--data_dir', default=None, help='input data directory.')
  p.add_argument('--gin_bindings', action='append', default=None, help='Gin parameter bindings.')
  p.add_argument('--gin_configs', action='append', default=[], help='Gin config files.')
  p.add_argument('--max_steps', type=int, default=None, help='stop early (smoke runs); default TrainConfig.max_steps')
  p.add_argument('--bf16', nargs='?', const='all', default=None, choices=['all', 'mlp', 'x3', 'x3mlp'],
                 help="bfloat16 MFMA operands (NRF_FLAG_BF16; no reference counterpart; fp32 master weights, posenc, exp_se3, compositing, "
                      "loss, Adam).  --bf16 / --bf16 all: the NeRF MLPs AND the SE3 warp trunk (BASELINE configs[3]; a warped point moves by "
                      "~5e-4 of its displacement; no systematic held-out-PSNR cost at the vrig preset's posenc widths over a 6000-step schedule, profiles/r05_bf16_warp_gap.json); --bf16 mlp: the NeRF MLPs only, the warp trunk stays float32 (NRF_FLAG_WARP_F32; about half the speed with the warp on; for captures whose deformation is large against the scene).  train.py "
                      "trains with a bfloat16 activation / gradient stash (~6x the fp32 step without the warp), eval.py renders with "
                      "bfloat16 operands (~1e-2 on colour, < 0.01 dB held-out PSNR).  --bf16 x3 (eval.py only): the NeRF MLPs in split-bf16 "
                      "arithmetic (NRF_FLAG_BF16X3: every float32 operand as a bf16 pair, three bf16 MFMAs per product, float32 accumulate) -- "
                      "float32-emulating, ~1e-6 of the float32 render on colour without the warp, ~1e-5 with it, at ~3x its speed (--bf16 x3mlp: the SE3 trunk stays on the float32 kernels)")
  p.add_argument('--graph', action='store_true', help='replay the whole train step (loss + gradient, all-reduce, Adam) from ONE hipGraph '
                 '(training.GraphedTrainStep): the reference jits the step into one XLA executable (train.py:254-262); worth it for small '
                 'per-GPU batches, where the ~25 launches of a step take about as long as the kernels')
  flags = p.parse_args(argv)
  flags.bf16 = {'all': True, 'mlp': 'mlp', 'x3': 'x3', 'x3mlp': 'x3mlp', None: False}[flags.bf16]   # the value models / training take (False, True, 'mlp', 'x3')
  return flags


def init_distributed():
  """torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment."""
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local = int(os.environ.get('LOCAL_RANK', '0'))
  torch.cuda.set_device(local)
  if world > 1 and not dist.is_initialized():
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))
  return (dist.get_rank() if world > 1 else 0), world, torch.device('cuda', local)


def make_datasource(flags, exp_config, model_config):
  spec = exp_config.datasource_spec or {'type': exp_config.datasource_type, 'data_dir': flags.data_dir}
  return datasets.from_config(
      spec, image_scale=exp_config.image_scale, use_appearance_id=model_config.use_appearance_metadata,
      use_camera_id=model_config.use_camera_metadata, use_warp_id=model_config.use_warp,
      use_time=model_config.warp_metadata_encoder_type == 'time',   # train.py:172, eval.py:290
      random_seed=exp_config.random_seed, **exp_config.datasource_kwargs)


def main(argv=None):
  flags = parse_flags(argv)
  if flags.bf16 in ('x3', 'x3mlp'):
    raise SystemExit('--bf16 x3 is an inference mode (eval.py): the training chains stash float32 or bfloat16 activations')
  gin.parse_config_files_and_bindings(config_files=flags.gin_configs, bindings=flags.gin_bindings, skip_unknown=True)
  exp_config, model_config, train_config = configs.ExperimentConfig(), configs.ModelConfig(), configs.TrainConfig()
  rank, world, device = init_distributed()
  log = (lambda *a: print(*a, flush=True)) if rank == 0 else (lambda *a: None)

  exp_dir = flags.base_folder if not exp_config.subname else os.path.join(flags.base_folder, exp_config.subname)
  summary_dir, checkpoint_dir = os.path.join(exp_dir, 'summaries', 'train'), os.path.join(exp_dir, 'checkpoints')
  if rank == 0:
    os.makedirs(checkpoint_dir, exist_ok=True)
    config_str = gin.operative_config_str()
    with open(os.path.join(exp_dir, 'config.gin'), 'w') as f:
      f.write(config_str)
  if train_config.batch_size % world != 0:
    raise ValueError('Batch size must be divisible by the number of devices.')

  datasource = make_datasource(flags, exp_config, model_config)
  train_iter = datasource.create_iterator(datasource.train_ids, flatten=True, shuffle=True,
                                          batch_size=train_config.batch_size, device=device)
  points = None
  if train_config.use_background_loss:      # train.py:177-189: per-device slices of the shuffled point cloud
    points = torch.from_numpy(datasource.load_points(shuffle=True)).to(device)
    per = min(len(points) // world, train_config.background_points_batch_size)
    points_pos = rank * per

  lr_sched = schedules.from_config(train_config.lr_schedule)
  warp_alpha_sched = schedules.from_config(train_config.warp_alpha_schedule)
  time_alpha_sched = schedules.from_config(train_config.time_alpha_schedule)
  elastic_sched = schedules.from_config(train_config.elastic_loss_weight_schedule)

  model, params = models.construct_nerf(
      exp_config.random_seed, model_config, batch_size=train_config.batch_size, appearance_ids=datasource.appearance_ids,
      camera_ids=datasource.camera_ids, warp_ids=datasource.warp_ids, near=datasource.near, far=datasource.far,
      use_warp_jacobian=train_config.use_elastic_loss, use_weights=train_config.use_elastic_loss, device=device)
  state = training.TrainState(optimizer=training.Optimizer(params), warp_alpha=warp_alpha_sched(0),
                              time_alpha=time_alpha_sched(0))
  scalar_params = training.ScalarParams(
      learning_rate=lr_sched(0), elastic_loss_weight=elastic_sched(0),
      warp_reg_loss_weight=train_config.warp_reg_loss_weight, warp_reg_loss_alpha=train_config.warp_reg_loss_alpha,
      warp_reg_loss_scale=train_config.warp_reg_loss_scale, background_loss_weight=train_config.background_loss_weight)
  state = checkpoints.restore_checkpoint(checkpoint_dir, state)
  init_step = state.optimizer.step + 1
  writer = utils.ScalarLog(summary_dir) if rank == 0 else None
  if writer:
    writer.text('gin/train', config_str, 0)

  max_steps = train_config.max_steps if flags.max_steps is None else min(flags.max_steps, train_config.max_steps)
  key = exp_config.random_seed + rank            # per-device keys (train.py:270-271)
  tracker = utils.TimeTracker()
  tracker.tic('data', 'total')
  log(f'Starting training at step {init_step}: {datasource.__class__.__name__}, {world} GPU(s), '
      f'batch {train_config.batch_size} rays')
  step = init_step - 1
  gstep = None      # --graph: built on the first batch (the capture needs static input buffers of the batch's shapes)
  step_flags = dict(use_elastic_loss=train_config.use_elastic_loss, elastic_reduce_method=train_config.elastic_reduce_method,
                    elastic_loss_type=train_config.elastic_loss_type, use_background_loss=train_config.use_background_loss,
                    use_warp_reg_loss=train_config.use_warp_reg_loss)
  for step in range(init_step, max_steps + 1):
    batch = next(train_iter)
    if points is not None:
      if points_pos + per > len(points):
        points_pos = rank * per
      batch['background_points'] = points[points_pos:points_pos + per]
      points_pos += per * world
    tracker.toc('data')
    scalar_params = dataclasses.replace(scalar_params, learning_rate=lr_sched(step),
                                                 elastic_loss_weight=elastic_sched(step))
    state.warp_alpha, state.time_alpha = warp_alpha_sched(step), time_alpha_sched(step)
    with tracker.record_time('train_step'):
      if flags.graph:
        if gstep is None:
          gstep = training.GraphedTrainStep(model, state, batch, scalar_params, bf16=flags.bf16, **step_flags)
        stats = gstep(key, scalar_params, warp_alpha=state.warp_alpha, time_alpha=state.time_alpha, batch=batch)
        key = training._step_keys(key)[0]
      else:
        state, stats, key = training.train_step(model, key, state, batch, scalar_params, bf16=flags.bf16, **step_flags)
      if step % train_config.print_every == 0 or step % train_config.log_every == 0:
        torch.cuda.synchronize(device)            # only when the numbers are read
    tracker.toc('total')
    if step % train_config.print_every == 0:
      log(f'step={step}, warp_alpha={state.warp_alpha:.04f}, time_alpha={state.time_alpha:.04f}, '
          f'{tracker.summary_str("last")}')
      for lv in ('coarse', 'fine'):
        log(f'\t{lv} metrics: ' + ', '.join(f'{k}={float(v):.04f}' for k, v in stats[lv].items()))
    if step % train_config.save_every == 0 and rank == 0:
      checkpoints.save_checkpoint(checkpoint_dir, state, step, keep=5)
    if step % train_config.log_every == 0 and writer:
      writer.scalar('params/learning_rate', scalar_params.learning_rate, step)
      writer.scalar('params/warp_alpha', state.warp_alpha, step)
      writer.scalar('params/elastic_loss/weight', scalar_params.elastic_loss_weight, step)
      for lv in ('coarse', 'fine'):
        for k, v in stats[lv].items():
          writer.scalar(f'{k}/{lv}', float(v), step)
      if 'background_loss' in stats:
        writer.scalar('loss/background', float(stats['background_loss']), step)
      for k, v in tracker.summary('mean').items():
        writer.scalar(f'time/{k}', v, step)
      tracker.reset()
    tracker.tic('data', 'total')
  if rank == 0 and step >= init_step and step % train_config.save_every != 0:
    checkpoints.save_checkpoint(checkpoint_dir, state, step, keep=5)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()
  return state


if __name__ == '__main__':
  main(sys.argv[1:])
