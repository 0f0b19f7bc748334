"""Host side of nerfies_amd.datasets (no GPU): the capture format of README.md:82-218 is read the way
datasets/nerfies.py:29-193 and datasets/core.py:76-105, 192-300 read it.  Ray generation is a GPU kernel:
tests/test_gpu_datasets.py."""
import json
import os

import numpy as np
import pytest

from nerfies_amd import datasets


@pytest.fixture()
def scene(tmp_path):
  d = str(tmp_path / 'capture')
  ids = datasets.write_synthetic_scene(d, num_frames=5, size=(16, 12), image_scale=2)
  return d, ids


def test_scene_ids_metadata(scene):
  d, ids = scene
  ds = datasets.NerfiesDataSource(d, image_scale=2, use_appearance_id=True, use_camera_id=True, use_warp_id=True)
  assert ds.train_ids == ids[:-1] and ds.val_ids == ids[-1:] and ds.all_ids == ids
  assert (ds.near, ds.far) == (0.05, 0.8) and ds.scene_scale == 0.1
  assert ds.appearance_ids == (0, 1, 2, 3) and ds.warp_ids == (0, 1, 2, 3) and ds.camera_ids == (0, 1)
  assert ds.item_metadata(ids[2]) == {'appearance': 2, 'camera': 0, 'warp': 2}
  strided = datasets.NerfiesDataSource(d, image_scale=2, use_warp_id=True, train_stride=2)
  assert strided.train_ids == ids[:-1][::2] and strided.warp_ids == (0, 2) and strided.appearance_ids == ()
  assert strided.item_metadata(ids[2]) == {'warp': 1}       # table row = position in the sorted training-id set
  assert datasets.from_config({'type': 'nerfies', 'data_dir': d}, image_scale=2).train_ids == ids[:-1]
  with pytest.raises(ValueError):
    datasets.from_config({'type': 'dynamic_scene', 'data_dir': d}, image_scale=2)


def test_use_time_gives_the_time_encoder_its_stamp(scene):
  """core.py:269-274, 298-303, 602-603: metadata['time'] = time_id / max(time_ids) * 2 - 1, time_id falling back to warp_id
  (nerfies.py:188-192); train.py:172 / eval.py:290 switch it on with warp_metadata_encoder_type == 'time'."""
  d, ids = scene
  ds = datasets.NerfiesDataSource(d, image_scale=2, use_warp_id=True, use_time=True)
  assert ds.use_time and ds.has_metadata and ds.time_ids == (0, 1, 2, 3)
  assert [ds.get_time(i) for i in ids[:4]] == [-1.0, -1.0 + 2 / 3, -1.0 + 4 / 3, 1.0]
  md = ds.item_metadata(ids[1])
  assert md['warp'] == 1 and md['time'] == pytest.approx(-1.0 + 2 / 3) and isinstance(md['time'], float)
  assert datasets.NerfiesDataSource(d, image_scale=2, use_warp_id=True).time_ids == ()
  import types
  import train as train_driver
  flags = types.SimpleNamespace(data_dir=d)
  exp = types.SimpleNamespace(datasource_spec=None, datasource_type='nerfies', image_scale=2, random_seed=0, datasource_kwargs={})
  for enc, want in (('time', True), ('glo', False)):
    mc = types.SimpleNamespace(use_appearance_metadata=False, use_camera_metadata=False, use_warp=True, warp_metadata_encoder_type=enc)
    assert train_driver.make_datasource(flags, exp, mc).use_time is want


def test_camera_is_rescaled_and_normalised(scene):
  d, ids = scene
  raw = json.load(open(os.path.join(d, 'camera', ids[1] + '.json')))
  ds = datasets.NerfiesDataSource(d, image_scale=2)
  c = ds.load_camera(ids[1])
  assert c.image_size.tolist() == [16, 12]                     # 32x24 capture at image_scale 2
  np.testing.assert_allclose(c.focal_length, raw['focal_length'] / 2)
  np.testing.assert_allclose(c.principal_point, np.array(raw['principal_point']) / 2)
  np.testing.assert_allclose(c.position, (np.array(raw['position']) - 5.0) * 0.1, atol=1e-6)
  np.testing.assert_allclose(c.radial_distortion, raw['radial_distortion'])
  half = ds.load_camera(ids[1], scale_factor=0.5)
  assert half.image_size.tolist() == [8, 6]


def test_rgb_points_items(scene):
  d, ids = scene
  ds = datasets.NerfiesDataSource(d, image_scale=2, use_warp_id=True)
  rgb = ds.load_rgb(ids[0])
  assert rgb.shape == (12, 16, 3) and rgb.dtype == np.float32 and 0 <= rgb.min() and rgb.max() <= 1
  np.testing.assert_allclose(rgb[3, 5], np.array([(5 * 7) % 256, (3 * 9) % 256, 8]) / 255.0)
  pts = ds.load_points()
  raw = np.load(os.path.join(d, 'points.npy'))
  assert pts.dtype == np.float32
  np.testing.assert_allclose(pts, (raw - 5.0) * 0.1, atol=1e-6)
  shuffled = ds.load_points(shuffle=True)
  assert sorted(map(tuple, shuffled.tolist())) == sorted(map(tuple, pts.tolist()))
  item = ds.get_item(ids[1])
  assert set(item) == {'rgb', 'camera', 'metadata'} and item['metadata'] == {'warp': 1}
  half = ds.get_item(ids[1], scale_factor=0.5)
  assert half['rgb'].shape == (6, 8, 3)
  np.testing.assert_allclose(half['rgb'][0, 0], item['rgb'][:2, :2].mean((0, 1)), rtol=1e-6)
  assert ds.load_test_cameras() == []


def test_rescale_image():
  img = np.arange(4 * 6 * 3, dtype=np.float32).reshape(4, 6, 3)
  assert datasets.rescale_image(img, 1.0) is img
  assert datasets.rescale_image(img, 2).shape == (8, 12, 3)
  np.testing.assert_allclose(datasets.rescale_image(img, 0.5)[1, 2], img[2:4, 4:6].mean((0, 1)))
  for bad in (0.0, 0.3):
    with pytest.raises(ValueError):
      datasets.rescale_image(img, bad)


def test_ray_table_batches_shard_like_the_reference():
  import torch
  n = 103
  cols = {'origins': torch.arange(n * 3, dtype=torch.float32).reshape(n, 3), 'rgb': torch.zeros(n, 3),
          'metadata/warp': torch.arange(n, dtype=torch.int32).reshape(n, 1)}
  table = datasets.RayTable(cols, n)
  it = table.batches(20, repeat=True)
  seen = [next(it) for _ in range(7)]
  assert all(b['origins'].shape == (20, 3) and b['metadata']['warp'].shape == (20, 1) for b in seen)
  assert seen[0]['metadata']['warp'][0, 0] == 0 and seen[4]['metadata']['warp'][0, 0] == 80
  assert seen[5]['metadata']['warp'][0, 0] == 0                   # wrapped: the 3-ray tail is dropped when repeating
  once = list(table.batches(20, repeat=False))
  assert [b['rgb'].shape[0] for b in once] == [20, 20, 20, 20, 20, 3]
  assert seen[1]['origins'].data_ptr() == cols['origins'][20:].data_ptr()    # views, no copies
  with pytest.raises(ValueError):
    next(datasets.RayTable(cols, n).batches(200))
