"""The nerfies on-disk capture format -> training / evaluation rays resident in HBM.

Reference: nerfies/datasets/nerfies.py:29-193 (NerfiesDataSource: scene.json, dataset.json, metadata.json,
camera/<id>.json, rgb/<scale>x/<id>.png, points.npy, camera-paths/<trajectory>/), nerfies/datasets/core.py:76-105
(load_camera), :192-300 (DataSource ids / metadata vocabulary), :392-447 (preload every ray, ONE global permutation,
batch), :110-160 (per-device reshape + prefetch), README.md:82-218 (format description).

MI355X-first layout: the reference preloads all rays on the host with NumPy, permutes them there and streams
batches to the devices every step.  Here each frame is decoded on the host (PNG -> uint8), uploaded once, turned
into rays by the camera kernel (nrf_camera_pixels_to_rays, one launch per frame) and appended to one flat ray table
in HBM -- 40 bytes per ray, so even a 1000-frame full-HD capture (2 G rays, 83 GB) fits one 288 GB GPU; the
global permutation is applied once on the device, after which a training batch is a contiguous slice and
`create_iterator` hands out views without any host traffic.  With torch.distributed initialised every rank holds
the same table (same seed => same permutation) and takes its own 1/world slice of each global batch, which is
the reference's reshape to (n_devices, batch / n_devices) (core.py:110-121)."""
import json
import os
from concurrent import futures
from typing import Any, Dict, Iterator, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import camera as cam


def load_scene_info(data_dir):
  """scene.json -> (center, scale, near, far) (datasets/nerfies.py:29-52)."""
  with open(os.path.join(data_dir, 'scene.json'), 'r') as f:
    sj = json.load(f)
  return np.array(sj['center']), sj['scale'], sj['near'], sj['far']


def load_camera(camera_path, scale_factor=1.0, scene_center=None, scene_scale=None) -> cam.Camera:
  """Camera JSON, rescaled image domain, position moved into the normalised scene frame (core.py:76-105)."""
  if not str(camera_path).endswith('.json'):
    raise ValueError('File must have extension .json.')
  camera = cam.Camera.from_json(camera_path)
  if scale_factor != 1.0:
    camera = camera.scale(scale_factor)
  if scene_center is not None:
    camera.position = camera.position - scene_center
  if scene_scale is not None:
    camera.position = camera.position * scene_scale
  return camera


def load_image(path) -> np.ndarray:
  """PNG -> float32 RGB in [0, 1] (datasets/nerfies.py:55-61; the reference decodes with cv2, here PIL)."""
  from PIL import Image
  with Image.open(path) as im:
    return np.asarray(im.convert('RGB'), dtype=np.uint8).astype(np.float32) / 255.0


def rescale_image(image: np.ndarray, scale_factor: float) -> np.ndarray:
  """image_utils.rescale_image (image_utils.py:73-101) for the integer cases the pipeline uses: 1/k is an area
  average over k x k blocks, k is pixel replication."""
  scale_factor = float(scale_factor)
  if scale_factor <= 0.0:
    raise ValueError('scale_factor must be a non-negative number.')
  if scale_factor == 1.0:
    return image
  h, w = image.shape[:2]
  if scale_factor.is_integer():
    k = int(scale_factor)
    return np.repeat(np.repeat(image, k, axis=0), k, axis=1)
  inv = 1.0 / scale_factor
  if inv.is_integer() and h % int(inv) == 0 and w % int(inv) == 0:
    k = int(inv)
    return image.reshape(h // k, k, w // k, k, *image.shape[2:]).mean(axis=(1, 3)).astype(image.dtype)
  raise ValueError(f'only integer up/down-scaling is built (got {scale_factor} for a {h}x{w} image)')


def parallel_map(fn, items, max_threads=None):
  with futures.ThreadPoolExecutor(max_threads) as ex:
    return list(ex.map(fn, items))


class DataSource:
  """Ids, metadata vocabularies and item loading (core.py:192-300, 565-619)."""

  def __init__(self, train_ids, val_ids, use_appearance_id=False, use_camera_id=False, use_warp_id=False,
               use_depth=False, use_relative_depth=False, use_time=False, random_seed=0, train_stride=1, val_stride=1,
               preload=True, **_):
    if use_depth or use_relative_depth:
      raise NotImplementedError('depth inputs are not on the built path (no shipped preset uses them)')
    self.use_time = bool(use_time)   # metadata['time'] for the TimeEncoder (core.py:217, 602-603)
    self._train_ids, self._val_ids = list(train_ids), list(val_ids)
    self.train_stride, self.val_stride = train_stride, val_stride
    self.use_appearance_id, self.use_camera_id, self.use_warp_id = use_appearance_id, use_camera_id, use_warp_id
    self.random_seed = random_seed
    self.rng = np.random.RandomState(random_seed)
    self.preload = preload
    self._vocab: Dict[str, tuple] = {}

  all_ids = property(lambda self: sorted(list(self.train_ids) + list(self.val_ids)))
  train_ids = property(lambda self: self._train_ids[::self.train_stride])
  val_ids = property(lambda self: self._val_ids[::self.val_stride])
  has_metadata = property(lambda self: self.use_appearance_id or self.use_warp_id or self.use_camera_id or self.use_time)

  def _ids(self, kind, enabled, getter):
    """Sorted set of the raw ids seen in the TRAINING items; model embeddings are indexed by position in it."""
    if not enabled:
      return tuple()
    if kind not in self._vocab:
      self._vocab[kind] = tuple(sorted(set(getter(i) for i in self.train_ids)))
    return self._vocab[kind]

  appearance_ids = property(lambda self: self._ids('appearance', self.use_appearance_id, self.get_appearance_id))
  camera_ids = property(lambda self: self._ids('camera', self.use_camera_id, self.get_camera_id))
  warp_ids = property(lambda self: self._ids('warp', self.use_warp_id, self.get_warp_id))
  time_ids = property(lambda self: self._ids('time', self.use_time, self.get_time_id))   # core.py:298-303

  def get_time(self, item_id) -> float:
    """Time stamp in [-1, 1] (core.py:272-274): time_id / max(time_ids) * 2 - 1."""
    return (self.get_time_id(item_id) / max(self.time_ids)) * 2.0 - 1.0

  def item_metadata(self, item_id) -> Dict[str, int]:
    """Embedding-table rows of an item (core.py:593-600)."""
    md = {}
    if self.use_appearance_id:
      md['appearance'] = self.appearance_ids.index(self.get_appearance_id(item_id))
    if self.use_camera_id:
      md['camera'] = self.camera_ids.index(self.get_camera_id(item_id))
    if self.use_warp_id:
      md['warp'] = self.warp_ids.index(self.get_warp_id(item_id))
    if self.use_time:   # a float, not a table row (core.py:602-603)
      md['time'] = float(self.get_time(item_id))
    return md

  def get_item(self, item_id, scale_factor=1.0) -> Dict[str, Any]:
    """{'rgb' [H,W,3] float32 host array, 'camera' Camera, 'metadata' {name: table row}} (core.py:565-619;
    'camera' replaces 'camera_params': the object is what the ray kernel takes)."""
    rgb = self.load_rgb(item_id)
    if scale_factor != 1.0:
      rgb = rescale_image(rgb, scale_factor)
    return {'rgb': rgb, 'camera': self.load_camera(item_id, scale_factor), 'metadata': self.item_metadata(item_id)}

  # ---- device side ----
  def item_rays(self, item_id, device='cuda', scale_factor=1.0) -> Dict[str, Any]:
    """One frame as [H, W, .] device tensors: origins, directions, pixels, rgb, metadata broadcast to [H, W, 1]
    (core.py:163-190): what eval.py iterates over with batch_size=0."""
    from . import evaluation
    item = self.get_item(item_id, scale_factor)
    rays = evaluation.rays_from_camera(item['camera'], item['metadata'], device)
    h, w = rays['origins'].shape[:2]
    if item['rgb'].shape[:2] != (h, w):
      raise ValueError(f'item {item_id}: image is {item["rgb"].shape[:2]} but the camera says {(h, w)}')
    rays['rgb'] = torch.from_numpy(item['rgb']).to(rays['origins'].device)
    return rays

  def create_ray_table(self, item_ids: Sequence[str], device='cuda', shuffle=True) -> 'RayTable':
    """Every ray of `item_ids`, flattened, (optionally) under one global permutation, resident on `device`
    (core.py:392-447 _create_preloaded_dataset with flatten=True)."""
    host_items = parallel_map(self.get_item, list(item_ids))       # PNG decode + JSON on host threads
    cols: Dict[str, List[torch.Tensor]] = {}
    from . import evaluation
    for item in host_items:
      rays = evaluation.rays_from_camera(item['camera'], item['metadata'], device)
      h, w = rays['origins'].shape[:2]
      if item['rgb'].shape[:2] != (h, w):
        raise ValueError(f'image is {item["rgb"].shape[:2]} but the camera says {(h, w)}')
      rays['rgb'] = torch.from_numpy(item['rgb']).to(rays['origins'].device)
      flat = {'origins': rays['origins'], 'directions': rays['directions'], 'pixels': rays['pixels'], 'rgb': rays['rgb']}
      flat.update({'metadata/' + k: v for k, v in rays.get('metadata', {}).items()})
      for k, v in flat.items():
        cols.setdefault(k, []).append(v.reshape(h * w, -1))
    table = {k: torch.cat(v, 0) for k, v in cols.items()}
    n = table['origins'].shape[0]
    if shuffle:
      g = torch.Generator(device='cpu').manual_seed(int(self.random_seed))
      perm = torch.randperm(n, generator=g).to(table['origins'].device)    # same on every rank
      table = {k: v.index_select(0, perm) for k, v in table.items()}
    return RayTable(table, n)

  def create_iterator(self, item_ids, batch_size: int, repeat: bool = True, flatten: bool = False, shuffle: bool = False,
                      prefetch_size: int = 0, shuffle_buffer_size: int = 1000000, devices=None, device='cuda'):
    """core.py:352-373.  batch_size > 0 with flatten: ray batches (this rank's shard) from the HBM-resident table;
    batch_size == 0: whole frames (evaluation).  prefetch / shuffle-buffer sizes are accepted for signature parity:
    nothing is streamed from the host, so there is nothing to prefetch."""
    del prefetch_size, shuffle_buffer_size, devices
    if batch_size > 0:
      if not flatten:
        raise NotImplementedError('batched whole frames are not used by train.py / eval.py')
      return self.create_ray_table(item_ids, device, shuffle).batches(batch_size, repeat)
    ids = list(item_ids)
    if shuffle:
      ids = [ids[i] for i in self.rng.permutation(len(ids))]

    def frames():
      while True:
        for i in ids:
          yield self.item_rays(i, device)
        if not repeat:
          return
    return frames()


class RayTable:
  """Flat ray table on the device; `batches` yields contiguous slices (views) of it."""

  def __init__(self, columns: Dict[str, torch.Tensor], num_rays: int):
    self.columns, self.num_rays = columns, num_rays

  def nbytes(self):
    return sum(v.numel() * v.element_size() for v in self.columns.values())

  def batch(self, start: int, size: int) -> Dict[str, Any]:
    out: Dict[str, Any] = {}
    for k, v in self.columns.items():
      if k.startswith('metadata/'):
        out.setdefault('metadata', {})[k[9:]] = v[start:start + size]
      else:
        out[k] = v[start:start + size]
    return out

  def batches(self, batch_size: int, repeat: bool = True) -> Iterator[Dict[str, Any]]:
    """Global batches of `batch_size` rays, of which this rank gets rows [rank*per, (rank+1)*per); the incomplete
    tail of an epoch is dropped when repeating (tf.data batches across the epoch boundary instead; the rays are
    i.i.d. after the permutation either way) and yielded short otherwise."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    if batch_size % world:
      raise ValueError(f'batch_size ({batch_size}) must be divisible by the number of devices ({world})')   # train.py:155
    per = batch_size // world
    if repeat and self.num_rays < batch_size:
      raise ValueError(f'{self.num_rays} rays cannot fill a batch of {batch_size}')
    while True:
      for start in range(0, self.num_rays - batch_size + 1, batch_size):
        yield self.batch(start + rank * per, per)
      tail = self.num_rays % batch_size
      if not repeat:
        if tail:
          t_per = -(-tail // world)
          lo = self.num_rays - tail + rank * t_per
          yield self.batch(lo, max(0, min(t_per, self.num_rays - lo)))
        return


class NerfiesDataSource(DataSource):
  """The capture format of README.md:82-218 (datasets/nerfies.py:81-193)."""

  def __init__(self, data_dir, image_scale: int, shuffle_pixels=False, camera_type='json',
               test_camera_trajectory='orbit-extreme', **kwargs):
    self.data_dir = str(data_dir)
    with open(os.path.join(self.data_dir, 'dataset.json'), 'r') as f:      # ids may skip frames COLMAP failed on
      dj = json.load(f)
    super().__init__(train_ids=[str(i) for i in dj['train_ids']], val_ids=[str(i) for i in dj['val_ids']], **kwargs)
    self.scene_center, self.scene_scale, self._near, self._far = load_scene_info(self.data_dir)
    self.test_camera_trajectory = test_camera_trajectory
    self.image_scale, self.shuffle_pixels = image_scale, shuffle_pixels
    self.rgb_dir = os.path.join(self.data_dir, 'rgb', f'{image_scale}x')
    if camera_type != 'json':
      raise ValueError(f'Unknown camera_type {camera_type}')
    self.camera_type, self.camera_ext = camera_type, '.json'
    self.camera_dir = os.path.join(self.data_dir, 'camera')
    self.metadata_dict = None
    mp = os.path.join(self.data_dir, 'metadata.json')
    if os.path.exists(mp):
      with open(mp, 'r') as f:
        self.metadata_dict = json.load(f)

  near = property(lambda self: self._near)
  far = property(lambda self: self._far)

  def get_rgb_path(self, item_id):
    return os.path.join(self.rgb_dir, f'{item_id}.png')

  def load_rgb(self, item_id):
    return load_image(self.get_rgb_path(item_id))

  def load_camera(self, item_id, scale_factor=1.0):
    path = item_id if str(item_id).endswith(self.camera_ext) and os.path.exists(str(item_id)) else \
        os.path.join(self.camera_dir, f'{item_id}{self.camera_ext}')
    return load_camera(path, scale_factor=scale_factor / self.image_scale, scene_center=self.scene_center,
                       scene_scale=self.scene_scale)

  def glob_cameras(self, path):
    return sorted(os.path.join(path, f) for f in os.listdir(path) if f.endswith(self.camera_ext))

  def load_test_cameras(self, count=None):
    d = os.path.join(self.data_dir, 'camera-paths', self.test_camera_trajectory)
    if not os.path.isdir(d):
      return []
    paths = self.glob_cameras(d)
    if count is not None:
      paths = paths[::max(1, len(paths) // count)]
    return parallel_map(self.load_camera, paths)

  def load_points(self, shuffle=False):
    """Background points in the normalised scene frame (datasets/nerfies.py:160-171)."""
    points = np.load(os.path.join(self.data_dir, 'points.npy'))
    points = ((points - self.scene_center) * self.scene_scale).astype(np.float32)
    if shuffle:
      points = points[self.rng.permutation(len(points))]
    return points

  def get_appearance_id(self, item_id):
    return self.metadata_dict[item_id]['appearance_id']

  def get_camera_id(self, item_id):
    return self.metadata_dict[item_id]['camera_id']

  def get_warp_id(self, item_id):
    return self.metadata_dict[item_id]['warp_id']

  def get_time_id(self, item_id):
    md = self.metadata_dict[item_id]
    return md['time_id'] if 'time_id' in md else md['warp_id']


def from_config(spec, **kwargs):
  """datasets/__init__.py:21-27: {'type': 'nerfies', 'data_dir': ...} -> data source."""
  spec = dict(spec)
  kind = spec.pop('type')
  if kind != 'nerfies':
    raise ValueError(f'Unknown datasource type {kind!r}')
  return NerfiesDataSource(**spec, **kwargs)


def write_synthetic_scene(data_dir, num_frames=4, size=(32, 24), image_scale=1, seed=0, num_points=64):
  """Writes a small capture in the reference's directory layout (README.md:82-218): a textured sphere seen from
  `num_frames` cameras on an arc (the last one goes to val_ids).  Used by tests and smoke runs; returns the ids."""
  from PIL import Image
  rng = np.random.default_rng(seed)
  w, h = size
  os.makedirs(os.path.join(data_dir, 'camera'), exist_ok=True)
  os.makedirs(os.path.join(data_dir, 'rgb', f'{image_scale}x'), exist_ok=True)
  ids = [f'{i:06d}' for i in range(num_frames)]
  base = cam.Camera(orientation=np.eye(3), position=np.zeros(3), focal_length=1.2 * w * image_scale,
                    principal_point=[w * image_scale / 2, h * image_scale / 2],
                    image_size=[w * image_scale, h * image_scale], radial_distortion=[0.01, 0.0, 0.0])
  metadata = {}
  for k, item in enumerate(ids):
    ang = (k - num_frames / 2) * 0.15
    pos = np.array([3.0 * np.sin(ang), 0.2 * k, -3.0 * np.cos(ang)]) + 5.0      # scene centre is (5, 5, 5)
    c = base.look_at(pos, np.array([5.0, 5.0, 5.0]), np.array([0.0, 1.0, 0.0]))
    with open(os.path.join(data_dir, 'camera', f'{item}.json'), 'w') as f:
      json.dump(c.to_json(), f)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 7 + k * 13) % 256, (yy * 9 + k * 5) % 256, (xx + yy + k * 29) % 256], -1).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(data_dir, 'rgb', f'{image_scale}x', f'{item}.png'))
    # the validation frame borrows the ids of the last training frame (validation-rig convention, README.md:165-180:
    # ids outside the training vocabulary have no embedding row)
    j = min(k, num_frames - 2)
    metadata[item] = {'appearance_id': j, 'warp_id': j, 'camera_id': j % 2}
  with open(os.path.join(data_dir, 'scene.json'), 'w') as f:
    json.dump({'center': [5.0, 5.0, 5.0], 'scale': 0.1, 'near': 0.05, 'far': 0.8}, f)
  with open(os.path.join(data_dir, 'dataset.json'), 'w') as f:
    json.dump({'count': num_frames, 'num_exemplars': num_frames - 1, 'ids': ids, 'train_ids': ids[:-1], 'val_ids': ids[-1:]}, f)
  with open(os.path.join(data_dir, 'metadata.json'), 'w') as f:
    json.dump(metadata, f)
  np.save(os.path.join(data_dir, 'points.npy'), 5.0 + rng.normal(size=(num_points, 3)))
  return ids
