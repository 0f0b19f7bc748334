"""Host-side pieces of the drivers that need no GPU: flag parsing, visualisation / image writers, the evaluation loop
with a stand-in renderer, meters and timers (eval.py:65-217, train.py:43-51, utils.py:370-465, visualization.py:150-219)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from nerfies_amd import utils, visualization as viz  # noqa: E402


def test_flags_match_the_reference_names():
  import train as train_driver
  f = train_driver.parse_flags(['--base_folder', '/tmp/x', '--data_dir', '/tmp/d', '--gin_configs', 'a.gin', '--gin_configs', 'b.gin',
                                '--gin_bindings', 'A.b = 1', '--gin_bindings', 'C.d = 2'])
  assert f.base_folder == '/tmp/x' and f.data_dir == '/tmp/d' and f.gin_configs == ['a.gin', 'b.gin']
  assert f.gin_bindings == ['A.b = 1', 'C.d = 2'] and f.max_steps is None and f.bf16 is False
  with pytest.raises(SystemExit):
    train_driver.parse_flags(['--data_dir', '/tmp/d'])            # base_folder is required (train.py:47)


def test_colorize_matches_matplotlib_and_saturates():
  import matplotlib
  x = np.linspace(-0.2, 1.2, 29).reshape(1, 29)
  got = viz.colorize(x, cmin=0.0, cmax=1.0, cmap='magma')
  cmap = matplotlib.colormaps['magma']
  inside = (x >= 0) & (x <= 1)
  want = np.asarray(cmap(np.linspace(0, 1, 256)))[:, :3]
  idx = np.clip(x, 0, 1) * 255
  lo = np.floor(idx).astype(int); hi = np.minimum(lo + 1, 255)
  interp = want[lo] + (want[hi] - want[lo]) * (idx - lo)[..., None]
  np.testing.assert_allclose(got[inside], interp[inside], atol=1e-12)
  assert (got[x > 1.0] == 1.0).all() and (got[x < 0.0] == 0.0).all()
  inv = viz.colorize(x, cmin=0.0, cmax=1.0, invert=True)
  np.testing.assert_allclose(inv[0, 5], viz.colorize(1.0 - x, cmin=0.0, cmax=1.0)[0, 5], atol=1e-12)
  assert (inv[x > 1.0] == 0.0).all() and (inv[x < 0.0] == 1.0).all()
  auto = viz.colorize(np.array([[2.0, 4.0]]))                     # cmin / cmax default to the data range
  np.testing.assert_allclose(auto[0, 0], want[0]); np.testing.assert_allclose(auto[0, 1], want[255])


def test_image_writers_round_trip(tmp_path):
  from PIL import Image
  img = np.random.default_rng(0).uniform(0, 1, (6, 9, 3)).astype(np.float32)
  u8 = viz.image_to_uint8(img)
  assert u8.dtype == np.uint8 and viz.image_to_uint8(u8) is u8
  viz.save_image(str(tmp_path / 'a.png'), u8)
  np.testing.assert_array_equal(np.asarray(Image.open(tmp_path / 'a.png')), u8)
  depth = np.random.default_rng(1).uniform(0.1, 900.0, (6, 9)).astype(np.float32)
  viz.save_depth(str(tmp_path / 'd.png'), depth)                  # 16-bit PNG of depth / 1000 (image_utils.py:164-165)
  back = np.asarray(Image.open(tmp_path / 'd.png')).astype(np.float64) / 65535 * 1000.0
  np.testing.assert_allclose(back, depth, atol=1000.0 / 65535)
  with pytest.raises(ValueError):
    viz.image_to_uint8(np.zeros((2, 2), np.int32))


def test_meters_timers_and_scalar_log(tmp_path):
  m = utils.ValueMeter()
  for v in (1.0, 2.0, 6.0):
    m.update(v)
  assert m.reduce('mean') == 3.0 and m.reduce('last') == 6.0 and abs(m.reduce('std') - np.std([1, 2, 6])) < 1e-12
  with pytest.raises(ValueError):
    m.reduce('median')
  t = utils.TimeTracker()
  t.tic('data', 'total'); t.toc('data')
  with t.record_time('train_step'):
    pass
  t.toc('total')
  s = t.summary('mean')
  assert set(s) == {'data', 'train_step', 'total', 'steps_per_sec'} and s['steps_per_sec'] > 0 and 'total=' in t.summary_str()
  assert utils.strided_subset(list(range(10)), 3) == [0, 3, 6, 9] and utils.strided_subset([1, 2], None) == [1, 2]
  assert abs(utils.compute_psnr(0.01) - 20.0) < 1e-12
  log = utils.ScalarLog(str(tmp_path / 'sum'))
  log.scalar('loss/rgb/fine', 0.5, 10); log.text('gin/train', 'x = 1', 0); log.close()
  rows = [json.loads(l) for l in open(tmp_path / 'sum' / 'scalars.jsonl')]
  assert rows[0] == {'tag': 'loss/rgb/fine', 'value': 0.5, 'step': 10} and rows[1]['text'] == 'x = 1'


def test_eval_loop_with_a_stand_in_renderer(tmp_path):
  """process_iterator / process_batch (eval.py:65-217): files written, metrics averaged, old renders rotated."""
  import eval as eval_driver
  h, w = 8, 12
  target = torch.rand(h, w, 3)

  def render_fn(state, batch, rng=0):
    return {'rgb': (batch['rgb'] * 0.9).clone(), 'depth': torch.full((h, w), 0.4), 'med_depth': torch.full((h, w), 0.5),
            'acc': torch.ones(h, w)}
  ds = type('DS', (), {'near': 0.1, 'far': 0.9, 'appearance_ids': (), 'warp_ids': (), 'camera_ids': ()})()
  frames = [{'rgb': target, 'origins': torch.zeros(h, w, 3)} for _ in range(2)]
  log = utils.ScalarLog(str(tmp_path / 'sum'))
  res = eval_driver.process_iterator('val', ['a/b', 'c'], iter(frames), 0, None, 120, render_fn, log, str(tmp_path / 'renders'), ds)
  mse = float(((target * 0.1) ** 2).mean())
  assert abs(res['mse'] - mse) < 1e-7 and abs(res['psnr'] - (-10 * np.log10(mse))) < 1e-4
  out = tmp_path / 'renders' / '00000120' / 'val'
  assert sorted(os.listdir(out)) == sorted(f'{k}_{i}.png' for i in ('a_b', 'c') for k in
                                          ('rgb', 'depth_expected', 'depth_expected_viz', 'depth_median', 'depth_median_viz'))
  rows = [json.loads(l) for l in open(tmp_path / 'sum' / 'scalars.jsonl')]
  assert {r['tag'] for r in rows} == {'metrics-eval/mse/val', 'metrics-eval/psnr/val'} and all(r['step'] == 120 for r in rows)
  for step in ('00000100', '00000110', '00000130'):
    os.makedirs(tmp_path / 'renders' / step)
  eval_driver.delete_old_renders(str(tmp_path / 'renders'), 2)
  assert sorted(os.listdir(tmp_path / 'renders')) == ['00000120', '00000130']


def test_replayed_ray_tree_must_match_the_captured_one():
  """evaluation.GraphedChunkRenderer copies new rays into the static buffers of its captured chunk: another key set, shape or
  dtype is refused with a message that says which (a missing sub-dict used to surface as a TypeError on None)."""
  import pytest
  import torch
  from nerfies_amd import evaluation
  from nerfies_amd import lib as L
  cap = {'origins': torch.zeros(8, 3), 'directions': torch.zeros(8, 3), 'metadata': {'warp': torch.zeros(8, 1, dtype=torch.int32)}}
  evaluation._check_same_tree(cap, {'origins': torch.ones(8, 3), 'directions': torch.ones(8, 3), 'metadata': {'warp': torch.ones(8, 1, dtype=torch.int32)}})
  with pytest.raises(L.NrfError, match='rays/metadata'):
    evaluation._check_same_tree(cap, {'origins': torch.ones(8, 3), 'directions': torch.ones(8, 3), 'metadata': {}})
  with pytest.raises(L.NrfError, match='rays has keys'):
    evaluation._check_same_tree(cap, {'origins': torch.ones(8, 3), 'directions': torch.ones(8, 3)})
  with pytest.raises(L.NrfError, match='rays/origins'):
    evaluation._check_same_tree(cap, {'origins': torch.ones(4, 3), 'directions': torch.ones(8, 3), 'metadata': {'warp': torch.ones(8, 1, dtype=torch.int32)}})
  with pytest.raises(L.NrfError, match='rays/metadata/warp'):
    evaluation._check_same_tree(cap, {'origins': torch.ones(8, 3), 'directions': torch.ones(8, 3), 'metadata': {'warp': torch.ones(8, 1)}})
  with pytest.raises(ValueError):
    evaluation.render_image(None, {'origins': torch.zeros(2, 2, 3)}, None, tile_parallel='rows')
