"""Experiment configuration: the four gin-configurable config objects of nerfies/configs.py (ModelConfig :35-105,
ExperimentConfig :108-124, TrainConfig :127-190, EvalConfig :193-212), same field names and defaults, on top of
nerfies_amd.gin_lite (gin-config itself is not installed here).

The classes are generated from field tables (name -> default), grouped by what the HIP path does with them, rather
than written out as class bodies; `dataclasses.fields(ModelConfig)` etc. behave as usual.  Activations are plain names
('relu', 'softplus', ...) registered as configurables so that `@nn.softplus` in the presets resolves; the HIP path
accepts relu and softplus for sigma (nrf_create rejects the rest)."""
import dataclasses
from typing import Any

from . import gin_lite as gin

ScheduleDef = Any


class _Activation(str):
  """A named activation; str subclass so it can be passed straight to NerfModel(activation=...)."""
  __name__ = property(lambda self: str(self))

  def __call__(self, *a, **k):
    raise TypeError(f'{self!s} is a name: the activation itself runs inside the HIP kernels')


relu, softplus, tanh, sigmoid = (_Activation(n) for n in ('relu', 'softplus', 'tanh', 'sigmoid'))
for _a in (relu, softplus, tanh, sigmoid):          # configs.py:27-32 (+ relu, the dataclass default)
  gin.external_configurable(_a, name=str(_a), module='flax.nn')


def _config_class(name: str, doc: str, *groups):
  """name + groups of {field: default} -> a gin-configurable dataclass.  Mutable defaults (dict) become factories."""
  fields = []
  for group in groups:
    for key, default in group.items():
      if isinstance(default, dict):
        spec = dataclasses.field(default_factory=lambda d=default: dict(d))
      else:
        spec = dataclasses.field(default=default)
      fields.append((key, Any, spec))
  cls = dataclasses.make_dataclass(name, fields)
  cls.__doc__ = doc
  cls.__module__ = __name__
  return gin.configurable(cls)


# ---- ModelConfig: what construct_nerf hands to nrf_create ----
_SAMPLING = dict(num_coarse_samples=64, num_fine_samples=128, use_stratified_sampling=True, use_linear_disparity=False,
                 use_white_background=False, use_sample_at_infinity=True, noise_std=None)
_NERF_MLP = dict(nerf_trunk_depth=8, nerf_trunk_width=256, nerf_skips=(4,), nerf_rgb_branch_depth=1, nerf_rgb_branch_width=128,
                 activation=relu, sigma_activation=relu, alpha_channels=1, rgb_channels=3)
_ENCODERS = dict(num_nerf_point_freqs=10, num_nerf_viewdir_freqs=4, use_viewdirs=True)
_CONDITIONS = dict(use_trunk_condition=False, use_alpha_condition=False, use_rgb_condition=False,
                   use_appearance_metadata=False, appearance_metadata_dims=8, use_camera_metadata=False, camera_metadata_dims=2)
_WARP = dict(use_warp=False, num_warp_freqs=8, num_warp_features=8, warp_field_type='translation',
             warp_metadata_encoder_type='glo', warp_kwargs={})
ModelConfig = _config_class('ModelConfig', 'Parameters for the model (configs.py:35-105).', _SAMPLING, _NERF_MLP, _ENCODERS,
                            _CONDITIONS, _WARP)

# ---- ExperimentConfig ----
ExperimentConfig = _config_class(
    'ExperimentConfig', 'Experiment configuration (configs.py:108-124).',
    dict(subname=None, image_scale=4, random_seed=12345, datasource_type='nerfies', datasource_spec=None, datasource_kwargs={}))

# ---- TrainConfig: schedules, regularisers, bookkeeping intervals ----
_OPTIM = dict(batch_size=gin.REQUIRED, max_steps=1000000,
              lr_schedule={'type': 'exponential', 'initial_value': 0.001, 'final_value': 0.0001, 'num_steps': 1000000},
              warp_alpha_schedule={'type': 'linear', 'initial_value': 0.0, 'final_value': 8.0, 'num_steps': 80000},
              time_alpha_schedule=('constant', 0.0))
_REGULARISERS = dict(use_elastic_loss=False, elastic_loss_weight_schedule=('constant', 0.0), elastic_reduce_method='weight',
                     elastic_loss_type='log_svals', use_background_loss=False, background_loss_weight=0.0,
                     background_points_batch_size=16384, use_warp_reg_loss=False, warp_reg_loss_weight=0.0,
                     warp_reg_loss_alpha=-2.0, warp_reg_loss_scale=0.001)
_INTERVALS = dict(shuffle_buffer_size=5000000, save_every=10000, log_every=500, histogram_every=5000, print_every=25)
TrainConfig = _config_class('TrainConfig', 'Parameters for training (configs.py:127-190).', _OPTIM, _REGULARISERS, _INTERVALS)

# ---- EvalConfig ----
EvalConfig = _config_class(
    'EvalConfig', 'Parameters for evaluation (configs.py:193-212).',
    dict(eval_once=False, save_output=True, chunk=8192, max_render_checkpoints=3, num_val_eval=10, num_train_eval=10,
         num_test_eval=10))
