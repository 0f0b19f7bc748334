"""The gin subset (nerfies_amd/gin_lite.py) and the config dataclasses (nerfies_amd/configs.py).  gin-config is a
third-party dependency of the reference and is not installed here, so these tests encode the semantics the presets
rely on: includes, lazy macros, later-wins bindings, @configurable references, REQUIRED,
skip_unknown, explicit kwargs beat bindings, operative config round trip.  When the reference tree is present (build
container only) every preset under /root/reference/configs is also parsed and turned into schedules."""
import dataclasses
import os

import pytest

from nerfies_amd import configs, schedules
from nerfies_amd import gin_lite as gin

BASE = """
# base file
num_warp_freqs = 8
elastic_init_weight = 0.01
ANNEALED = {
  'type': 'linear',
  'initial_value': 0.0,
  'final_value': %num_warp_freqs,   # lazy
  'num_steps': 80000,
}
DECAY = {
  'type': 'piecewise',
  'schedules': [
    (50000, ('constant', %elastic_init_weight)),
    (100000, ('cosine_easing', %elastic_init_weight, 1e-8, 100000)),
  ]
}
LR = {'type': 'exponential', 'initial_value': %init_lr, 'final_value': %final_lr, 'num_steps': %max_steps}
ModelConfig.sigma_activation = @nn.softplus
ModelConfig.use_warp = False
ModelConfig.num_warp_freqs = %num_warp_freqs
TrainConfig.batch_size = %batch_size
TrainConfig.max_steps = %max_steps
TrainConfig.lr_schedule = %LR
TrainConfig.warp_alpha_schedule = %ANNEALED
TrainConfig.elastic_loss_weight_schedule = %DECAY
EvalConfig.chunk = %eval_batch_size
"""

TOP = """
include 'sub/base.gin'
max_steps = 250000
batch_size = 6144
eval_batch_size = 8096
init_lr = 0.001
final_lr = 0.0001
num_warp_freqs = 6          # overrides the base macro AFTER the include: ANNEALED must see 6
ModelConfig.use_warp = True # later binding wins
ModelConfig.nerf_skips = (4,)
ModelConfig.noise_std = None
ModelConfig.warp_field_type = 'se3'   # a '#' inside a string: 'a#b'
SomethingElse.value = 3
"""


@pytest.fixture(autouse=True)
def _clean():
  gin.clear_config()
  yield
  gin.clear_config()


def _write(tmp_path):
  (tmp_path / 'sub').mkdir()
  (tmp_path / 'sub' / 'base.gin').write_text(BASE)
  top = tmp_path / 'top.gin'
  top.write_text(TOP)
  return str(top)


def test_presets_style_file(tmp_path):
  top = _write(tmp_path)
  with pytest.raises(gin.GinError, match='SomethingElse'):
    gin.parse_config_files_and_bindings([top], None)
  gin.clear_config()
  gin.parse_config_files_and_bindings([top], ['TrainConfig.print_every = 7', "ExperimentConfig.subname = 'sweep'"],
                                      skip_unknown=True)
  m, t, e, x = configs.ModelConfig(), configs.TrainConfig(), configs.EvalConfig(), configs.ExperimentConfig()
  assert m.use_warp is True and m.num_warp_freqs == 6 and m.sigma_activation == 'softplus' and m.activation == 'relu'
  assert m.nerf_skips == (4,) and m.noise_std is None and m.warp_field_type == 'se3'
  assert t.batch_size == 6144 and t.max_steps == 250000 and t.print_every == 7 and e.chunk == 8096 and x.subname == 'sweep'
  assert t.warp_alpha_schedule['final_value'] == 6
  assert schedules.from_config(t.warp_alpha_schedule)(40000) == pytest.approx(3.0)
  lr = schedules.from_config(t.lr_schedule)
  assert lr(0) == pytest.approx(1e-3) and lr(250000) == pytest.approx(1e-4)
  el = schedules.from_config(t.elastic_loss_weight_schedule)
  assert el(10) == pytest.approx(0.01) and el(150000 - 1) < 1e-7
  # explicit keyword arguments beat bindings (eval.py:239)
  assert configs.EvalConfig(chunk=123).chunk == 123
  assert gin.query_parameter('ModelConfig.num_warp_freqs') == 6 and gin.query_parameter('%batch_size') == 6144


def test_operative_config_round_trip(tmp_path):
  gin.parse_config_files_and_bindings([_write(tmp_path)], None, skip_unknown=True)
  before = [dataclasses.asdict(c()) for c in (configs.ModelConfig, configs.TrainConfig, configs.EvalConfig)]
  text = gin.operative_config_str()
  assert 'ModelConfig.sigma_activation = @' in text and 'TrainConfig.batch_size = 6144' in text
  gin.clear_config()
  gin.parse_config(text)
  after = [dataclasses.asdict(c()) for c in (configs.ModelConfig, configs.TrainConfig, configs.EvalConfig)]
  assert before == after


def test_required_and_errors():
  with pytest.raises(gin.GinError, match='REQUIRED'):
    configs.TrainConfig()
  assert configs.TrainConfig(batch_size=4).batch_size == 4
  gin.parse_config('TrainConfig.batch_size = %nope')
  with pytest.raises(gin.GinError, match='never defined'):
    configs.TrainConfig()
  gin.clear_config()
  gin.parse_config('a = %b\nb = %a\nTrainConfig.batch_size = %a')
  with pytest.raises(gin.GinError, match='cycle'):
    configs.TrainConfig()
  gin.clear_config()
  with pytest.raises(gin.GinError, match='no parameter'):
    gin.parse_config('ModelConfig.not_a_field = 1')
    configs.ModelConfig()
  gin.clear_config()
  for bad in ('ModelConfig.use_warp = __import__("os")', 'ModelConfig.use_warp = 1 + 1', 'x = [1, 2', 'just words',
              "include 'missing.gin'", 'ModelConfig.sigma_activation = @nn.unknown_fn\n'):
    gin.clear_config()
    with pytest.raises(gin.GinError):
      gin.parse_config(bad)
      configs.ModelConfig()


def test_value_syntax():
  gin.parse_config("""
import nerfies.something
v1 = -1.5e-3
v2 = [1, (2, 3), {'k': None, 'f': True}]
v3 = "double # not a comment"
v4 = @nn.relu
ModelConfig.warp_kwargs = {'a': %v1, 'b': %v2}
""")
  assert gin.query_parameter('%v1') == -1.5e-3
  assert gin.query_parameter('%v2') == [1, (2, 3), {'k': None, 'f': True}]
  assert gin.query_parameter('%v3') == 'double # not a comment'
  assert gin.query_parameter('%v4') == 'relu'
  assert configs.ModelConfig().warp_kwargs == {'a': -1.5e-3, 'b': [1, (2, 3), {'k': None, 'f': True}]}


def test_model_config_defaults_match_reference_dataclass():
  """Field names and defaults of configs.py:35-212 (read off the reference; noted in SURVEY.md 2)."""
  m = configs.ModelConfig()
  assert (m.nerf_trunk_depth, m.nerf_trunk_width, m.nerf_rgb_branch_depth, m.nerf_rgb_branch_width) == (8, 256, 1, 128)
  assert (m.num_nerf_point_freqs, m.num_nerf_viewdir_freqs, m.num_coarse_samples, m.num_fine_samples) == (10, 4, 64, 128)
  assert m.warp_field_type == 'translation' and m.sigma_activation == 'relu' and m.use_sample_at_infinity
  t = configs.TrainConfig(batch_size=1)
  assert t.background_points_batch_size == 16384 and t.elastic_reduce_method == 'weight' and t.save_every == 10000
  assert configs.EvalConfig().chunk == 8192 and configs.ExperimentConfig().random_seed == 12345


REF_CONFIGS = '/root/reference/configs'
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIPPED = os.path.join(REPO, 'configs')
RUNNABLE = ('gpu_fullhd.gin', 'gpu_quarterhd.gin', 'gpu_quarterhd_4gpu.gin', 'gpu_vrig_paper.gin', 'test_local.gin', 'test_vrig.gin')


def _effective(path, cwd):
  """Every config dataclass as a dict after parsing `path` from `cwd` (the reference's presets include relative paths)."""
  import dataclasses
  old = os.getcwd()
  os.chdir(cwd)
  try:
    gin.clear_config()
    gin.parse_config_files_and_bindings([path], None, skip_unknown=True)
    return {c.__name__: dataclasses.asdict(c()) for c in (configs.ExperimentConfig, configs.ModelConfig, configs.TrainConfig, configs.EvalConfig)}
  finally:
    os.chdir(old)


def _check_preset(f, m, t):
  expect = {'gpu_vrig_paper.gin': (128, 128, 6144, 6), 'gpu_quarterhd.gin': (128, 128, 6144, 8),
            'gpu_fullhd.gin': (256, 256, 4096, 8), 'test_vrig.gin': (64, 64, 1024, 8), 'test_local.gin': (64, 64, 1024, 8),
            'gpu_quarterhd_4gpu.gin': (128, 128, 3072, 8)}
  assert m.sigma_activation == 'softplus' and m.use_warp and m.warp_field_type == 'se3'
  for name in ('lr_schedule', 'warp_alpha_schedule', 'elastic_loss_weight_schedule'):
    assert schedules.from_config(getattr(t, name))(1000) >= 0
  assert (m.num_coarse_samples, m.num_fine_samples, t.batch_size, m.num_warp_freqs) == expect[f], f
  # every preset's model is accepted by the HIP library (nrf_create is host-only) and reports its own leaf shapes
  from nerfies_amd import models
  model, fp = models.construct_nerf(0, m, t.batch_size, [0, 1, 2], [0, 1], [0, 1, 2], 0.1, 1.0, device='cpu')
  k = fp['nerf_mlps_fine']['MLP_0']['hidden_4']['kernel']
  assert k.shape == (m.nerf_trunk_width + 3 + 6 * m.num_nerf_point_freqs, m.nerf_trunk_width), f


def test_every_shipped_preset_parses():
  """configs/*.gin of THIS repo (scripts/make_presets.py): the presets a user passes to train.py / eval.py; runs anywhere."""
  assert sorted(os.listdir(SHIPPED)) == sorted(RUNNABLE)
  for f in RUNNABLE:
    gin.clear_config()
    gin.parse_config_files_and_bindings([os.path.join(SHIPPED, f)], None, skip_unknown=True)
    m, t = configs.ModelConfig(), configs.TrainConfig()
    configs.EvalConfig(), configs.ExperimentConfig()
    _check_preset(f, m, t)


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason='reference tree only exists in the build container')
def test_shipped_presets_equal_the_reference_presets():
  """Each shipped (flattened) preset resolves to EXACTLY the configuration the reference's include chain resolves to."""
  for f in RUNNABLE:
    ours = _effective(os.path.join(SHIPPED, f), REPO)
    ref = _effective(os.path.join('configs', f), '/root/reference')
    assert ours == ref, (f, {c: {k: (ours[c][k], ref[c][k]) for k in ours[c] if ours[c][k] != ref[c][k]} for c in ours})


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason='reference tree only exists in the build container')
def test_every_reference_preset_parses(monkeypatch):
  monkeypatch.chdir('/root/reference')        # presets include both 'warp_defaults.gin' and 'configs/warp_defaults.gin'
  for f in sorted(os.listdir(REF_CONFIGS)):
    if f in ('defaults.gin', 'warp_defaults.gin'):      # "Do not run this directly": macros left for the includer
      continue
    gin.clear_config()
    gin.parse_config_files_and_bindings([os.path.join('configs', f)], None, skip_unknown=True)
    m, t = configs.ModelConfig(), configs.TrainConfig()
    configs.EvalConfig(), configs.ExperimentConfig()
    _check_preset(f, m, t)
