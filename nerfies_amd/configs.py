"""Experiment configuration: the four gin-configurable dataclasses of nerfies/configs.py (ModelConfig :35-105,
ExperimentConfig :108-124, TrainConfig :127-190, EvalConfig :193-212) with the same field names and defaults, on
top of nerfies_amd.gin_lite (gin-config itself is not installed here).  Activations are plain names ('relu',
'softplus', ...) registered as configurables so that `@nn.softplus` in the presets resolves; the HIP path accepts
relu and softplus for sigma (nrf_create rejects the rest)."""
import dataclasses
from typing import Any, Mapping, Optional, Tuple

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


@gin.configurable
@dataclasses.dataclass
class ModelConfig:
  """Parameters for the model (configs.py:35-105)."""
  use_linear_disparity: bool = False
  use_white_background: bool = False
  use_stratified_sampling: bool = True
  use_sample_at_infinity: bool = True
  noise_std: Optional[float] = None
  nerf_trunk_depth: int = 8
  nerf_trunk_width: int = 256
  nerf_rgb_branch_depth: int = 1
  nerf_rgb_branch_width: int = 128
  activation: Any = relu
  sigma_activation: Any = relu
  nerf_skips: Tuple[int, ...] = (4,)
  alpha_channels: int = 1
  rgb_channels: int = 3
  num_nerf_point_freqs: int = 10
  num_nerf_viewdir_freqs: int = 4
  num_coarse_samples: int = 64
  num_fine_samples: int = 128
  use_viewdirs: bool = True
  use_trunk_condition: bool = False
  use_alpha_condition: bool = False
  use_rgb_condition: bool = False
  use_appearance_metadata: bool = False
  appearance_metadata_dims: int = 8
  use_camera_metadata: bool = False
  camera_metadata_dims: int = 2
  use_warp: bool = False
  num_warp_freqs: int = 8
  num_warp_features: int = 8
  warp_field_type: str = 'translation'
  warp_metadata_encoder_type: str = 'glo'
  warp_kwargs: Mapping[str, Any] = dataclasses.field(default_factory=dict)


@gin.configurable
@dataclasses.dataclass
class ExperimentConfig:
  """Experiment configuration (configs.py:108-124)."""
  subname: Optional[str] = None
  image_scale: int = 4
  random_seed: int = 12345
  datasource_type: str = 'nerfies'
  datasource_spec: Optional[Mapping[str, Any]] = None
  datasource_kwargs: Mapping[str, Any] = dataclasses.field(default_factory=dict)


@gin.configurable
@dataclasses.dataclass
class TrainConfig:
  """Parameters for training (configs.py:127-190)."""
  batch_size: int = gin.REQUIRED
  lr_schedule: ScheduleDef = dataclasses.field(default_factory=lambda: {
      'type': 'exponential', 'initial_value': 0.001, 'final_value': 0.0001, 'num_steps': 1000000})
  max_steps: int = 1000000
  warp_alpha_schedule: ScheduleDef = dataclasses.field(default_factory=lambda: {
      'type': 'linear', 'initial_value': 0.0, 'final_value': 8.0, 'num_steps': 80000})
  time_alpha_schedule: ScheduleDef = ('constant', 0.0)
  use_elastic_loss: bool = False
  elastic_loss_weight_schedule: ScheduleDef = ('constant', 0.0)
  elastic_reduce_method: str = 'weight'
  elastic_loss_type: str = 'log_svals'
  use_background_loss: bool = False
  background_loss_weight: float = 0.0
  background_points_batch_size: int = 16384
  use_warp_reg_loss: bool = False
  warp_reg_loss_weight: float = 0.0
  warp_reg_loss_alpha: float = -2.0
  warp_reg_loss_scale: float = 0.001
  shuffle_buffer_size: int = 5000000
  save_every: int = 10000
  log_every: int = 500
  histogram_every: int = 5000
  print_every: int = 25


@gin.configurable
@dataclasses.dataclass
class EvalConfig:
  """Parameters for evaluation (configs.py:193-212)."""
  eval_once: bool = False
  save_output: bool = True
  chunk: int = 8192
  max_render_checkpoints: int = 3
  num_val_eval: Optional[int] = 10
  num_train_eval: Optional[int] = 10
  num_test_eval: Optional[int] = 10
