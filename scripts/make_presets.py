#!/usr/bin/env python
"""Writes configs/*.gin: the six runnable presets of the reference (configs/{test_local,test_vrig,gpu_quarterhd,gpu_quarterhd_4gpu,
gpu_fullhd,gpu_vrig_paper}.gin), FLATTENED -- every preset is one self-contained file of bindings (the reference builds them from
defaults.gin -> warp_defaults.gin -> preset through includes and macros; SURVEY.md A.7 lists the effective values).  The binding
names and values are the contract `train.py --gin_configs configs/<preset>.gin` depends on; tests/test_configs_gin.py checks, where
/root/reference exists, that every shipped preset resolves to exactly the configuration the reference's own file resolves to.

    python scripts/make_presets.py        # rewrites configs/
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sched(kind, **kw):
  return {'type': kind, **kw}


def lr(v0, v1, steps):
  return sched('exponential', initial_value=v0, final_value=v1, num_steps=steps)


def warp_alpha(final, steps=80000):
  return sched('linear', initial_value=0.0, final_value=final, num_steps=steps)


def elastic_decay(w0):
  # held for 50 k steps, then eased down to 1e-8 over 100 k (milestones are durations: schedules.py:159-172)
  return sched('piecewise', schedules=[(50000, ('constant', w0)), (100000, ('cosine_easing', w0, 1e-8, 100000))])


def warp_preset(*, scale, batch, chunk, max_steps, lr0, lr1, coarse, fine, point_freqs, every):
  """gpu_quarterhd / gpu_quarterhd_4gpu / gpu_fullhd: warp_defaults.gin + the preset's macros."""
  return {
      'ExperimentConfig': dict(image_scale=scale, random_seed=12345),
      'ModelConfig': dict(use_viewdirs=True, use_stratified_sampling=True, sigma_activation='@nn.softplus',
                          use_appearance_metadata=True, use_warp=True, warp_field_type='se3', num_warp_freqs=8, num_warp_features=8,
                          num_nerf_point_freqs=point_freqs, nerf_trunk_width=256, nerf_trunk_depth=8,
                          num_coarse_samples=coarse, num_fine_samples=fine),
      'TrainConfig': dict(batch_size=batch, max_steps=max_steps, lr_schedule=lr(lr0, lr1, max_steps),
                          warp_alpha_schedule=warp_alpha(8), use_elastic_loss=True,
                          elastic_loss_weight_schedule=elastic_decay(0.01), use_background_loss=True, background_loss_weight=1.0,
                          print_every=every[0], log_every=every[1], save_every=every[2]),
      'EvalConfig': dict(eval_once=False, save_output=True, chunk=chunk),
  }


PRESETS = {
    'test_local.gin': ('sanity-check run on one GPU (not a quality setting)', {
        'ExperimentConfig': dict(image_scale=4),
        'ModelConfig': dict(num_coarse_samples=64, num_fine_samples=64, use_viewdirs=True, use_stratified_sampling=True,
                            use_appearance_metadata=True, use_warp=True, warp_field_type='se3', num_warp_features=3,
                            num_warp_freqs=8, sigma_activation='@nn.softplus'),
        'TrainConfig': dict(max_steps=200000, lr_schedule=lr(0.001, 0.0001, 250000), batch_size=1024,
                            warp_alpha_schedule=warp_alpha(8.0), use_elastic_loss=True,
                            elastic_loss_weight_schedule=elastic_decay(0.01), use_background_loss=False, background_loss_weight=1.0,
                            print_every=10, log_every=100, save_every=1000),
        'EvalConfig': dict(eval_once=False, save_output=True, chunk=8192),
    }),
    'test_vrig.gin': ('validation-rig smoke test: 8 x 128 trunk, camera code, deterministic sampling', {
        'ExperimentConfig': dict(image_scale=8, random_seed=12345),
        'ModelConfig': dict(use_viewdirs=True, use_stratified_sampling=False, sigma_activation='@nn.softplus',
                            use_appearance_metadata=False, use_camera_metadata=True, camera_metadata_dims=2, use_warp=True,
                            warp_field_type='se3', num_warp_freqs=8, num_warp_features=8, num_nerf_point_freqs=8,
                            nerf_trunk_width=128, nerf_trunk_depth=8, num_coarse_samples=64, num_fine_samples=64),
        'TrainConfig': dict(batch_size=1024, max_steps=250000, lr_schedule=lr(0.001, 0.0001, 250000),
                            warp_alpha_schedule=sched('constant', value=8), use_elastic_loss=True,
                            elastic_loss_weight_schedule=sched('constant', value=0.001), use_background_loss=True,
                            background_loss_weight=1.0, print_every=1, log_every=100, save_every=1000),
        'EvalConfig': dict(eval_once=False, save_output=True, chunk=8192, num_val_eval=None, num_train_eval=None),
    }),
    'gpu_quarterhd.gin': ('quarter-HD capture, 8 GPUs (README: "around 14 hours")',
                          warp_preset(scale=4, batch=6144, chunk=8096, max_steps=250000, lr0=0.001, lr1=0.0001, coarse=128, fine=128,
                                      point_freqs=8, every=(200, 500, 5000))),
    'gpu_quarterhd_4gpu.gin': ('quarter-HD capture on 4 GPUs: half the batch, twice the steps',
                               warp_preset(scale=4, batch=3072, chunk=4096, max_steps=500000, lr0=0.0007, lr1=0.00007, coarse=128,
                                           fine=128, point_freqs=8, every=(200, 500, 5000))),
    'gpu_fullhd.gin': ('full-HD capture, 8 GPUs (README: "around 3 days")',
                       warp_preset(scale=1, batch=4096, chunk=4096, max_steps=1000000, lr0=0.00075, lr1=0.000075, coarse=256, fine=256,
                                   point_freqs=10, every=(200, 500, 10000))),
    'gpu_vrig_paper.gin': ('the validation-rig configuration behind the paper table, 8 GPUs', {
        'ExperimentConfig': dict(image_scale=4, random_seed=12345),
        'ModelConfig': dict(use_viewdirs=True, use_stratified_sampling=True, sigma_activation='@nn.softplus',
                            use_appearance_metadata=False, use_camera_metadata=True, camera_metadata_dims=2, use_warp=True,
                            warp_field_type='se3', num_warp_freqs=6, num_warp_features=8, use_sample_at_infinity=True,
                            num_nerf_point_freqs=8, nerf_trunk_width=256, nerf_trunk_depth=8, num_coarse_samples=128,
                            num_fine_samples=128),
        'TrainConfig': dict(batch_size=6144, max_steps=250000, lr_schedule=lr(0.001, 0.0001, 250000),
                            warp_alpha_schedule=warp_alpha(6), use_elastic_loss=True, elastic_reduce_method='weight',
                            elastic_loss_weight_schedule=sched('constant', value=0.001), use_background_loss=True,
                            background_loss_weight=1.0, use_warp_reg_loss=False, warp_reg_loss_weight=1e-2, print_every=500,
                            log_every=500, histogram_every=1000, save_every=5000),
        'EvalConfig': dict(eval_once=False, save_output=True, chunk=8096, num_val_eval=None, num_train_eval=None),
    }),
}


def literal(v):
  if isinstance(v, str) and v.startswith('@'):
    return v
  if isinstance(v, dict):
    return '{' + ', '.join(f'{k!r}: {literal(x)}' for k, x in v.items()) + '}'
  if isinstance(v, (list, tuple)):
    body = ', '.join(literal(x) for x in v)
    return '[' + body + ']' if isinstance(v, list) else '(' + body + (',)' if len(v) == 1 else ')')
  return repr(v)


def main():
  out = os.path.join(ROOT, 'configs')
  os.makedirs(out, exist_ok=True)
  for name, (what, cfg) in PRESETS.items():
    lines = [f'# nerfies_amd preset "{name[:-4]}": {what}.',
             f'# Flattened equivalent of the reference\'s configs/{name} (generated by scripts/make_presets.py; do not edit by hand).',
             '# Override any line with --gin_bindings "TrainConfig.batch_size = 64".', '']
    for section in ('ExperimentConfig', 'ModelConfig', 'TrainConfig', 'EvalConfig'):
      for key, val in cfg.get(section, {}).items():
        lines.append(f'{section}.{key} = {literal(val)}')
      lines.append('')
    with open(os.path.join(out, name), 'w') as f:
      f.write('\n'.join(lines))
    print('wrote', name)


if __name__ == '__main__':
  main()
