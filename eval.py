"""Evaluation driver: the command-line surface of google/nerfies' eval.py (eval.py:43-58 flags, :65-420) on the
MI355X path: restores the latest checkpoint, renders strided subsets of the train / val items and the test camera
path in chunks (hipGraph-captured forward, rays generated on the GPU), writes rgb / depth PNGs and mse / psnr.

  python eval.py --base_folder EXP --data_dir CAPTURE --gin_configs EXP/config.gin [--gin_bindings "EvalConfig.eval_once = True"]

Multiscale SSIM is reported for frames of at least 176 px per side (five scales), from a restatement of
tf.image.ssim_multiscale's published algorithm (nerfies_amd/evaluation.py)."""
import functools
import os
import shutil
import sys
import time

import numpy as np
import torch

from nerfies_amd import checkpoints, configs, evaluation, models, training, utils, visualization as viz
from nerfies_amd import gin_lite as gin
import train as train_driver


def process_batch(*, batch, rng, state, tag, item_id, step, writer, render_fn, save_dir, datasource):
  """Renders one frame, writes its images, returns its metrics (eval.py:65-152)."""
  item_id = item_id.replace('/', '_')
  render = render_fn(state, batch, rng=rng)
  rgb = render['rgb'].cpu().numpy()
  depth_exp, depth_med = render['depth'].cpu().numpy(), render['med_depth'].cpu().numpy()
  if save_dir:
    os.makedirs(save_dir, exist_ok=True)
    colorize_depth = functools.partial(viz.colorize, cmin=datasource.near, cmax=datasource.far, invert=True)
    viz.save_image(os.path.join(save_dir, f'rgb_{item_id}.png'), viz.image_to_uint8(rgb))
    viz.save_image(os.path.join(save_dir, f'depth_expected_viz_{item_id}.png'), viz.image_to_uint8(colorize_depth(depth_exp)))
    viz.save_depth(os.path.join(save_dir, f'depth_expected_{item_id}.png'), depth_exp)
    viz.save_image(os.path.join(save_dir, f'depth_median_viz_{item_id}.png'), viz.image_to_uint8(colorize_depth(depth_med)))
    viz.save_depth(os.path.join(save_dir, f'depth_median_{item_id}.png'), depth_med)
  out = {}
  if 'rgb' in batch:
    m = evaluation.image_metrics(render['rgb'], batch['rgb'])
    out = {k: float(v) for k, v in m.items()}
    if _rank() == 0:
      print(f'\t[{tag}] {item_id}: ' + ', '.join(f'{k}={v:.04f}' for k, v in out.items()), flush=True)
  return out


def _rank():
  import torch.distributed as dist
  return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def process_iterator(tag, item_ids, iterator, rng, state, step, render_fn, writer, save_dir, datasource):
  """eval.py:155-217."""
  save_dir = os.path.join(save_dir, f'{step:08d}', tag) if save_dir else None
  meters = {}
  for i, (item_id, batch) in enumerate(zip(item_ids, iterator)):
    if tag == 'test':      # a test camera has no metadata of its own: one id per table, drawn from the step (eval.py:171-199)
      g = np.random.RandomState(step)
      md = {}
      for name, ids in (('appearance', datasource.appearance_ids), ('warp', datasource.warp_ids), ('camera', datasource.camera_ids)):
        if ids:
          # a raw id VALUE, as random.choice(datasource.*_ids) gives (the embedding row: tables have max(ids)+1 rows,
          # models.py:121-131) -- the same convention training.train_step's background ids follow
          md[name] = torch.full(batch['origins'][..., :1].shape, int(ids[g.randint(len(ids))]), dtype=torch.int32,
                                device=batch['origins'].device)
      if getattr(datasource, 'use_time', False):
        # eval.py:189-194: timestamp ~ U[0, 1) from the step's key, then jnp.full(shape, timestamp, dtype=uint32) -- the
        # reference's integer cast truncates it to 0; kept (drop-in behaviour), as a float32 tensor the TimeEncoder takes
        timestamp = float(np.uint32(g.uniform(0.0, 1.0)))
        md['time'] = torch.full(batch['origins'][..., :1].shape, timestamp, dtype=torch.float32, device=batch['origins'].device)
      batch['metadata'] = md
    stats = process_batch(batch=batch, rng=rng, state=state, tag=tag, item_id=item_id, step=step, writer=writer,
                          render_fn=render_fn, save_dir=save_dir, datasource=datasource)
    for k, v in stats.items():
      meters.setdefault(k, utils.ValueMeter()).update(v)
  for k, m in meters.items():
    writer.scalar(f'metrics-eval/{k}/{tag}', m.reduce('mean'), step)
  return {k: m.reduce('mean') for k, m in meters.items()}


def delete_old_renders(render_dir, max_renders):
  for path in sorted(os.listdir(render_dir))[:-max_renders]:
    shutil.rmtree(os.path.join(render_dir, path))


def main(argv=None):
  flags = train_driver.parse_flags(argv)
  gin.parse_config_files_and_bindings(config_files=flags.gin_configs, bindings=flags.gin_bindings, skip_unknown=True)
  exp_config = configs.ExperimentConfig()
  model_config = configs.ModelConfig(use_stratified_sampling=False)        # eval.py:239: explicit kwarg beats the binding
  train_config, eval_config = configs.TrainConfig(), configs.EvalConfig()
  rank, world, device = train_driver.init_distributed()
  exp_dir = flags.base_folder if not exp_config.subname else os.path.join(flags.base_folder, exp_config.subname)
  summary_dir, renders_dir = os.path.join(exp_dir, 'summaries', 'eval'), os.path.join(exp_dir, 'renders')
  checkpoint_dir = os.path.join(exp_dir, 'checkpoints')
  os.makedirs(renders_dir, exist_ok=True)
  datasource = train_driver.make_datasource(flags, exp_config, model_config)

  train_eval_ids = utils.strided_subset(datasource.train_ids, eval_config.num_train_eval)
  val_eval_ids = utils.strided_subset(datasource.val_ids, eval_config.num_val_eval)
  test_cameras = datasource.load_test_cameras(count=eval_config.num_test_eval)

  model, params = models.construct_nerf(
      20200823, model_config, batch_size=eval_config.chunk, appearance_ids=datasource.appearance_ids,
      camera_ids=datasource.camera_ids, warp_ids=datasource.warp_ids, near=datasource.near, far=datasource.far,
      use_warp_jacobian=False, use_weights=False, device=device)
  init_state = training.TrainState(optimizer=training.Optimizer(params))
  renderer = evaluation.GraphedChunkRenderer(model, bf16=flags.bf16)   # hipGraph replay per chunk
  render_fn = functools.partial(evaluation.render_image, model_fn=renderer, device_count=world, chunk=eval_config.chunk)
  writer = utils.ScalarLog(summary_dir, enabled=rank == 0)   # summaries, meters and prints on process 0 only
  last_step, results = 0, {}
  while True:
    if checkpoints.latest_checkpoint(checkpoint_dir) is None:
      if eval_config.eval_once:
        raise FileNotFoundError(f'no checkpoint under {checkpoint_dir}')
      time.sleep(10)
      continue
    state = checkpoints.restore_checkpoint(checkpoint_dir, init_state)
    step = state.optimizer.step
    if step <= last_step:
      time.sleep(10)
      continue
    save_dir = renders_dir if eval_config.save_output and rank == 0 else None
    common = dict(rng=0, state=state, step=step, render_fn=render_fn, writer=writer, save_dir=save_dir, datasource=datasource)
    results['val'] = process_iterator('val', val_eval_ids, datasource.create_iterator(val_eval_ids, batch_size=0, repeat=False, device=device), **common)
    results['train'] = process_iterator('train', train_eval_ids, datasource.create_iterator(train_eval_ids, batch_size=0, repeat=False, device=device), **common)
    if test_cameras:
      frames = (evaluation.rays_from_camera(c, None, device) for c in test_cameras)
      results['test'] = process_iterator('test', [f'{i:03d}' for i in range(len(test_cameras))], frames, **common)
    if save_dir:
      delete_old_renders(renders_dir, eval_config.max_render_checkpoints)
    if eval_config.eval_once or step >= train_config.max_steps:
      break
    last_step = step
  return results


if __name__ == '__main__':
  main(sys.argv[1:])
