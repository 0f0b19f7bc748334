"""Checkpoint I/O (nerfies_amd/checkpoints.py): flax-msgpack wire format, TrainState round trip, file rotation.
Runs without a GPU: the parameter layout comes from nrf_create/nrf_param_layout, which are host-only."""
import os
import types

import msgpack
import numpy as np
import pytest
import torch

from nerfies_amd import checkpoints, models, training


def _state(seed=0, **cfg):
  base = dict(num_coarse_samples=8, num_fine_samples=8, use_warp=True, warp_field_type='se3', use_camera_metadata=True,
              sigma_activation='softplus')
  base.update(cfg)
  model, fp = models.construct_nerf(seed, types.SimpleNamespace(**base), 0, [0, 1, 2], [0, 1], [0, 1, 2, 3], 0.1, 1.0,
                                    device='cpu')
  opt = training.Optimizer(fp)
  g = torch.Generator().manual_seed(seed + 1)
  opt.m.copy_(torch.randn(opt.m.shape, generator=g))
  opt.v.copy_(torch.rand(opt.v.shape, generator=g))
  opt.step = 1234 + seed
  return model, training.TrainState(optimizer=opt, warp_alpha=2.5, time_alpha=0.0)


def test_wire_format_is_flax_msgpack():
  """ndarray = ExtType(1, packb((shape, dtype name, bytes))), NumPy scalar = ExtType(3, ...), maps keyed by str."""
  d = {'a': {'kernel': np.arange(6, dtype=np.float32).reshape(2, 3)}, 'step': np.int32(7), 'n': 3, 'f': 0.5}
  raw = msgpack.unpackb(checkpoints.to_bytes(d), raw=False)
  ext = raw['a']['kernel']
  assert isinstance(ext, msgpack.ExtType) and ext.code == 1
  shape, dtype, buf = msgpack.unpackb(ext.data, raw=False)
  assert list(shape) == [2, 3] and dtype == 'float32' and buf == np.arange(6, dtype='<f4').tobytes()
  assert raw['step'].code == 3 and raw['n'] == 3 and raw['f'] == 0.5
  back = checkpoints.from_bytes(checkpoints.to_bytes(d))
  np.testing.assert_array_equal(back['a']['kernel'], d['a']['kernel'])
  assert back['step'] == 7 and back['step'].dtype == np.int32 and back['a']['kernel'].flags.writeable


def test_state_dict_has_the_reference_structure():
  _, state = _state()
  d = checkpoints.state_to_dict(state)
  assert set(d) == {'optimizer', 'warp_alpha', 'time_alpha'} and set(d['optimizer']) == {'state', 'target'}
  tgt = d['optimizer']['target']['model']
  assert tgt['nerf_mlps_coarse']['MLP_0']['hidden_4']['kernel'].shape[1] == 256
  assert tgt['warp_field']['branches_w']['logit']['kernel'].shape == (128, 3)
  ps = d['optimizer']['state']['param_states']['model']['nerf_mlps_fine']['MLP_0']['hidden_0']['bias']
  assert set(ps) == {'grad_ema', 'grad_sq_ema'} and ps['grad_ema'].shape == (256,)
  assert int(d['optimizer']['state']['step']) == 1234 and float(d['warp_alpha']) == 2.5


def test_save_restore_round_trip_and_rotation(tmp_path):
  ckpt = str(tmp_path / 'ckpt')
  _, state = _state(0)
  assert checkpoints.restore_checkpoint(ckpt, state) is state          # nothing there yet: fresh run
  for step in (100, 200, 300):
    state.optimizer.step = step
    checkpoints.save_checkpoint(ckpt, state, step, keep=2)
  assert sorted(os.listdir(ckpt)) == ['checkpoint_200', 'checkpoint_300']
  with pytest.raises(ValueError, match='outdated'):
    checkpoints.save_checkpoint(ckpt, state, 250)
  _, other = _state(5)
  assert not torch.equal(other.optimizer.target.flat, state.optimizer.target.flat)
  flat_ptr = other.optimizer.target.flat.data_ptr()
  checkpoints.restore_checkpoint(ckpt, other)
  assert other.optimizer.target.flat.data_ptr() == flat_ptr           # loaded into the existing buffers
  layout = state.optimizer.target.layout
  for name, off, shape in layout.entries:       # leaves only: the flat buffer has alignment gaps between them
    n = int(np.prod(shape))
    assert torch.equal(other.optimizer.target.flat[off:off + n], state.optimizer.target.flat[off:off + n]), name
    assert torch.equal(other.optimizer.m[off:off + n], state.optimizer.m[off:off + n]), name
    assert torch.equal(other.optimizer.v[off:off + n], state.optimizer.v[off:off + n]), name
  assert other.optimizer.step == 300 and other.warp_alpha == 2.5
  checkpoints.restore_checkpoint(ckpt, other, step=200)
  assert other.optimizer.step == 200
  with pytest.raises(ValueError, match='not found'):
    checkpoints.restore_checkpoint(ckpt, other, step=100)
  raw = checkpoints.restore_checkpoint(os.path.join(ckpt, 'checkpoint_300'), None)
  assert int(raw['optimizer']['state']['step']) == 300
  checkpoints.save_checkpoint(ckpt, state, 250, keep=2, overwrite=True)
  assert sorted(os.listdir(ckpt)) == ['checkpoint_200', 'checkpoint_250']


def test_restore_rejects_a_checkpoint_of_another_model(tmp_path):
  _, state = _state(0)
  checkpoints.save_checkpoint(str(tmp_path), state, 1)
  _, nowarp = _state(0, use_warp=False)
  checkpoints.restore_checkpoint(str(tmp_path), nowarp)               # a superset checkpoint loads (extra leaves ignored)
  _, bigger = _state(0, use_appearance_metadata=True)
  checkpoints.save_checkpoint(str(tmp_path), nowarp, 2)
  with pytest.raises(KeyError, match='does not match'):
    checkpoints.restore_checkpoint(str(tmp_path), bigger)
