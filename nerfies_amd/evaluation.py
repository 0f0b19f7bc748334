"""Host-side mirror of nerfies.evaluation.render_image (reference: nerfies/evaluation.py:28-101)."""
import math
from typing import Any, Callable, Dict, Optional

import torch
import torch.distributed as dist


def _tree_map(fn, tree):
  if isinstance(tree, dict):
    return {k: _tree_map(fn, v) for k, v in tree.items()}
  return fn(tree)


class GraphedChunkRenderer:
  """One fixed-size chunk of NerfModel.apply captured in a hipGraph (torch.cuda.CUDAGraph over the
  library's launches on the capture stream) and replayed per chunk: the eval forward is ~20 kernel
  launches of a few hundred microseconds at 8k rays, so launch gaps matter (BASELINE config E).

  model_fn-compatible: `renderer(key_0, key_1, params, rays_dict, warp_extra)`; rays are copied into
  static buffers, the graph is replayed and copies of the static outputs are returned.
  One graph per (chunk size, parameter buffer, warp_alpha, time_alpha, metadata keys) is kept -- the two step scalars are
  by-value kernel arguments baked into the captured launches -- so a frame whose last chunk is
  shorter replays two graphs instead of re-capturing twice per frame; render_image additionally edge-pads the tail to
  the full chunk when the renderer asks for it (`wants_fixed_chunks`), so that one graph serves the whole frame."""
  wants_fixed_chunks = True
  MAX_GRAPHS = 8
  # what NerfModel.apply reads of a ray tree (models.py:289-375).  Dataset items also carry 'rgb' / 'pixels' / 'depth'
  # (datasets.item_rays) while camera-path frames do not (rays_from_camera): only the consumed keys are captured and copied, so
  # one renderer serves both kinds of frame (eval.py renders val / train items and then test cameras through the same one)
  CONSUMED = ('origins', 'directions', 'viewdirs', 'metadata')

  def __init__(self, model, use_warp=True, bf16=False):
    self.model = model
    self.use_warp = use_warp
    self.bf16 = bf16   # NRF_FLAG_BF16 inference mode (bfloat16 MLP operands)
    self._slots = {}   # key -> (graph, static inputs, static outputs); insertion-ordered, oldest evicted
    self.captures = 0

  def _capture(self, fp, rays, warp_extra):
    model = self.model
    ins = _tree_map(lambda x: x.clone(), rays)
    call = lambda out=None: model.apply({'params': fp}, ins, warp_extra, use_warp=self.use_warp, out=out, bf16=self.bf16)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
      outs = call()               # warm-up: uploads the descriptor tables (not capturable), sizes the workspace
      call(outs)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      call(outs)
    self.captures += 1
    return graph, ins, outs

  def __call__(self, key_0, key_1, params, rays, warp_extra):
    # the static buffers are overwritten by the next replay: a plain model_fn caller gets copies
    return _tree_map(lambda x: x.clone(), self.call_static(key_0, key_1, params, rays, warp_extra))

  def call_static(self, key_0, key_1, params, rays, warp_extra):
    """As __call__, but returns the graph's own output buffers (valid until the next replay): render_image copies each chunk
    straight into the frame at its offset, so a chunk's outputs are moved once instead of cloned and then concatenated."""
    del key_0, key_1               # eval is deterministic (eval.py:239 forces use_stratified_sampling off)
    rays = {k: rays[k] for k in self.CONSUMED if rays.get(k) is not None}
    n = rays['origins'].shape[0]
    # every scalar of lib.StepScalars is part of the key: a replay would otherwise render with the captured value
    scalars = tuple(float((warp_extra or {}).get(k, 0.0)) for k in ('alpha', 'time_alpha'))
    key = (n, params.flat.data_ptr(), scalars, 'viewdirs' in rays, tuple(sorted((rays.get('metadata') or {}).keys())))
    slot = self._slots.get(key)
    if slot is None:
      # every chunk size has its own workspace (descriptor tables included), so graphs of different sizes coexist
      while len(self._slots) >= self.MAX_GRAPHS:
        del self._slots[next(iter(self._slots))]
      slot = self._slots[key] = self._capture(params, rays, warp_extra)
    else:
      def cp(dst, src):
        if isinstance(dst, dict):
          for k in dst:
            cp(dst[k], src[k])
        else:
          dst.copy_(src)
      _check_same_tree(slot[1], rays)   # same keys / shapes / dtypes as the captured chunk, or a clear NrfError
      cp(slot[1], rays)
    slot[0].replay()
    return slot[2]


def _check_same_tree(dst, src, path='rays'):
  """The replay path copies new rays into the captured static buffers: same keys, same shapes, same dtypes -- or a clear error
  (a missing sub-dict used to surface as a TypeError on None)."""
  from nerfies_amd import lib as L
  if isinstance(dst, dict):
    if not isinstance(src, dict) or set(dst) != set(src):
      have = sorted(src) if isinstance(src, dict) else type(src).__name__
      raise L.NrfError(f'GraphedChunkRenderer: {path} has keys {have}, the captured chunk had {sorted(dst)}; render with the same '
                       'ray tree or use a new renderer')
    for k in dst:
      _check_same_tree(dst[k], src[k], f'{path}/{k}')
  elif tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
    raise L.NrfError(f'GraphedChunkRenderer: {path} is {tuple(src.shape)} {src.dtype}, the captured chunk had {tuple(dst.shape)} {dst.dtype}')


def _all_gather_rows(gathered, packed):
  """dist.all_gather_into_tensor of a rank's packed rows.  RCCL (backend 'nccl') is a stream operation and takes the device
  buffers as they are.  gloo -- the transport of the tests that put several ranks on ONE GPU -- is a host collective: staged
  through host tensors explicitly (the device-to-host copy is the synchronisation point), so that it never reads a device buffer
  the stream has not finished writing."""
  if packed.is_cuda and dist.get_backend() != 'nccl':
    host = torch.empty(gathered.shape, dtype=gathered.dtype)
    dist.all_gather_into_tensor(host, packed.cpu())
    gathered.copy_(host)
  else:
    dist.all_gather_into_tensor(gathered, packed)


def _pack(ret, rows):
  """Every output key of a rendered chunk side by side in one (rows, sum of widths) float32 buffer: one collective carries all."""
  keys = list(ret.keys())
  cols = [ret[k].reshape(rows, -1).to(torch.float32) for k in keys]
  return keys, cols, [c.shape[1] for c in cols]


def render_image(state, rays_dict: Dict[str, Any], model_fn: Callable, device_count: int = 1, rng=0,
                 chunk: int = 8192, default_ret_key: Optional[str] = None, tile_parallel: str = 'band'):
  """Renders all pixels of an (H,W) ray image in chunks (evaluation.py:62-99).

  model_fn(key_0, key_1, params, chunk_rays_dict, warp_extra) -> {'coarse': {...}, 'fine': {...}}.
  With torch.distributed initialised the frame is split over the ranks (image tiles are disjoint: no reduction, the
  reference all_gathers and keeps replica 0, eval.py:339, evaluation.py:92):
    tile_parallel='band' (default): every rank renders a contiguous BAND of whole chunks -- full-size launches, the shape the
      kernels are tuned for -- into a band buffer, and ONE all_gather per frame assembles it (BASELINE configs[4]: "8-GPU
      image-tile parallel");
    tile_parallel='chunk': the reference's own order -- every chunk is cut `world` ways (evaluation.py:61-92), one packed
      all_gather per chunk (64 latency-bound collectives and 1/world-size launches for a 960x540 frame on 8 GPUs).
  Both give the single-rank frame bit for bit (rows are rendered independently of their position in a launch).  The last
  chunk is edge-padded (evaluation.py:71-78): to the full chunk for a graph-replaying renderer, to a multiple of the world
  size in 'chunk' mode."""
  if tile_parallel not in ('band', 'chunk'):
    raise ValueError("tile_parallel must be 'band' or 'chunk'")
  h, w = rays_dict['origins'].shape[:2]
  flat = _tree_map(lambda x: x.reshape(h * w, -1), rays_dict)
  num_rays = h * w
  dist_on = dist.is_available() and dist.is_initialized()   # a one-rank communicator still runs the gather (RCCL on the GPU)
  world = dist.get_world_size() if dist_on else 1
  rank = dist.get_rank() if world > 1 else 0
  del device_count
  call = getattr(model_fn, 'call_static', model_fn)   # a graph renderer hands out its static output buffers (no per-chunk clone)
  num_chunks = int(math.ceil(num_rays / chunk))
  # a graph-replaying renderer wants ONE chunk size per frame: the tail is edge-padded to the full chunk (rendered and
  # dropped) instead of being a second shape; otherwise it is padded to a multiple of the world size only
  fixed = bool(getattr(model_fn, 'wants_fixed_chunks', False)) and num_chunks > 1

  def padded_chunk(batch_idx, multiple):
    i0 = batch_idx * chunk
    rays = _tree_map(lambda x: x[i0:i0 + chunk], flat)
    n = rays['origins'].shape[0]
    full = -(-chunk // multiple) * multiple if fixed else n
    pad = max(full - n, 0) + (multiple - max(full, n) % multiple) % multiple
    if pad:
      rays = _tree_map(lambda x: torch.cat([x, x[-1:].expand(pad, *x.shape[1:])], 0), rays)
    return i0, n, n + pad, rays

  if dist_on and tile_parallel == 'band':
    # balanced bands: rank r renders base (+ 1 for the first `rem` ranks) whole chunks starting at r * base + min(r, rem) --
    # 9 chunks on 8 ranks are 2 + 1 x 7, not 2 x 4 + 1 + 0 x 3.  all_gather wants equal pieces, so every band buffer holds
    # cmax chunks and the frame is assembled from each rank's leading rows
    base, rem = divmod(num_chunks, world)
    cmax = base + (1 if rem else 0)
    first = lambda r: r * base + min(r, rem)
    count = lambda r: base + (1 if r < rem else 0)
    band_rows = cmax * chunk
    band, meta = None, None
    for c in range(first(rank), first(rank) + count(rank)):
      i0, n, _, rays = padded_chunk(c, 1)
      out = call(rng, rng + 1, state.optimizer.target, rays, state.warp_extra)
      ret = out[default_ret_key or ('fine' if 'fine' in out else 'coarse')]
      keys, cols, widths = _pack(ret, rays['origins'].shape[0])
      if band is None:
        band = torch.zeros(band_rows, sum(widths), dtype=torch.float32, device=cols[0].device)
        meta = (keys, widths, [tuple(ret[k].shape[1:]) for k in keys], [ret[k].dtype for k in keys])
      at, col = i0 - first(rank) * chunk, 0
      for cdata, wd in zip(cols, widths):
        band[at:at + n, col:col + wd].copy_(cdata[:n])
        col += wd
    # only when there are more ranks than chunks does a rank own none; it then learns the output layout from rank 0 (every rank
    # sees base == 0 locally, so all of them join this broadcast or none does).  Otherwise: ONE collective per frame.
    if world > 1 and base == 0:
      lay = [meta]
      dist.broadcast_object_list(lay, src=0)
      meta = lay[0]
    keys, widths, shapes, dtypes = meta
    if band is None:
      band = torch.zeros(band_rows, sum(widths), dtype=torch.float32, device=flat['origins'].device)
    gathered = torch.empty(world * band_rows, band.shape[1], dtype=band.dtype, device=band.device)
    _all_gather_rows(gathered, band)     # ONE collective per frame
    if rem:   # bands of unequal length: keep each rank's leading count(r) chunks
      gathered = torch.cat([gathered[r * band_rows:r * band_rows + count(r) * chunk] for r in range(world) if count(r)], 0)
    split = torch.split(gathered[:num_rays], widths, 1)
    return {k: split[i].reshape(h, w, *shapes[i]).to(dtypes[i]) for i, k in enumerate(keys)}

  frame = None          # output key -> (num_rays, ...) buffer, filled chunk by chunk at the chunk's offset
  for batch_idx in range(num_chunks):
    i0, n, total, chunk_rays = padded_chunk(batch_idx, world)
    per = total // world
    mine = _tree_map(lambda x: x[rank * per:(rank + 1) * per], chunk_rays) if world > 1 else chunk_rays
    out = call(rng, rng + 1, state.optimizer.target, mine, state.warp_extra)
    ret_key = default_ret_key or ('fine' if 'fine' in out else 'coarse')
    ret = out[ret_key]
    if dist_on:   # ONE all_gather per chunk: every output key packed side by side into a (per, sum of widths) buffer
      keys, cols, widths = _pack(ret, per)
      packed = torch.cat(cols, 1).contiguous()
      gathered = torch.empty(world * per, packed.shape[1], dtype=packed.dtype, device=packed.device)
      _all_gather_rows(gathered, packed)
      split = torch.split(gathered, widths, 1)
      ret = {k: split[i].reshape(world * per, *ret[k].shape[1:]).to(ret[k].dtype) for i, k in enumerate(keys)}
    if frame is None:
      frame = {k: torch.empty(num_rays, *v.shape[1:], dtype=v.dtype, device=v.device) for k, v in ret.items()}
    for k, v in ret.items():
      frame[k][i0:i0 + n].copy_(v[:n])     # drops the padding; the only copy a chunk's outputs see
  return {k: v.reshape(h, w, *v.shape[1:]) for k, v in frame.items()}


def rays_from_camera(camera, metadata: Optional[Dict[str, int]] = None, device='cuda'):
  """One dataset item for a camera, built on the GPU: what datasets/core.py:163-190 (_camera_to_rays_fn +
  _tf_broadcast_metadata_fn) assembles on the host -- origins / directions / pixels [H, W, .] and every metadata id
  ('warp', 'appearance', 'camera') broadcast to [H, W, 1] int32."""
  rays = camera.to_rays(device)
  h, w = rays['origins'].shape[:2]
  if metadata:   # ids are int32 table rows; 'time' is the float stamp of the TimeEncoder (core.py:508-509, 602-603)
    dev = rays['origins'].device
    rays['metadata'] = {k: (torch.full((h, w, 1), float(v), dtype=torch.float32, device=dev) if k == 'time' else
                            torch.full((h, w, 1), int(v), dtype=torch.int32, device=dev)) for k, v in metadata.items()}
  return rays


def compute_psnr(mse):
  """utils.compute_psnr (utils.py:283-293): -10 log10(mse)."""
  return -10.0 * torch.log10(torch.as_tensor(mse))


_MSSSIM_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def compute_multiscale_ssim(image1: torch.Tensor, image2: torch.Tensor, max_val: float = 1.0, filter_size: int = 11,
                            filter_sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> torch.Tensor:
  """MS-SSIM of two [H, W, C] images (eval.py:60-62 calls tf.image.ssim_multiscale(image1, image2, max_val=1.0)).

  TensorFlow is a third-party dependency that is absent here, so its published algorithm (Wang, Simoncelli & Bovik 2003,
  with TF's defaults) is restated and is UNPINNED against TF itself: per channel, 5 scales; at each scale an
  11x11 sigma-1.5 Gaussian window (VALID) gives the local means / variances / covariance, the contrast-structure term
  cs = (2 s12 + c2) / (s1 + s2 + c2) and the full ssim = cs * (2 m1 m2 + c1) / (m1^2 + m2^2 + c1) are averaged over
  the window positions; images are halved by 2x2 average pooling between scales (odd sizes padded symmetrically);
  result = prod_i relu(cs_i)^w_i over the first 4 scales times relu(ssim_5)^w_5, averaged over channels.  Needs
  H, W >= 11 * 2^4 = 176."""
  x = image1.to(torch.float32).permute(2, 0, 1)[None]
  y = image2.to(x.device, torch.float32).permute(2, 0, 1)[None]
  c = x.shape[1]
  if min(x.shape[-2:]) < filter_size * 2 ** (len(_MSSSIM_WEIGHTS) - 1):
    raise ValueError(f'MS-SSIM needs images of at least {filter_size * 2 ** (len(_MSSSIM_WEIGHTS) - 1)} px per side')
  r = torch.arange(filter_size, dtype=torch.float32, device=x.device) - (filter_size - 1) / 2
  g = torch.exp(-0.5 * (r / filter_sigma) ** 2)
  g = g / g.sum()
  win = (g[:, None] * g[None, :])[None, None].repeat(c, 1, 1, 1)
  blur = lambda t: torch.nn.functional.conv2d(t, win, groups=c)
  c1, c2 = (k1 * max_val) ** 2, (k2 * max_val) ** 2
  terms = []
  for i, w in enumerate(_MSSSIM_WEIGHTS):
    if i:
      ph, pw = x.shape[-2] % 2, x.shape[-1] % 2
      if ph or pw:
        x = torch.nn.functional.pad(x, (0, pw, 0, ph), mode='replicate')     # symmetric padding of one pixel
        y = torch.nn.functional.pad(y, (0, pw, 0, ph), mode='replicate')
      x = torch.nn.functional.avg_pool2d(x, 2)
      y = torch.nn.functional.avg_pool2d(y, 2)
    m1, m2 = blur(x), blur(y)
    num0, den0 = m1 * m2 * 2.0, m1 * m1 + m2 * m2
    cs = (blur(x * y) * 2.0 - num0 + c2) / (blur(x * x + y * y) - den0 + c2)
    if i + 1 < len(_MSSSIM_WEIGHTS):
      terms.append(torch.relu(cs.mean(dim=(-2, -1))) ** w)
    else:
      terms.append(torch.relu((cs * (num0 + c1) / (den0 + c1)).mean(dim=(-2, -1))) ** w)
  return torch.stack(terms, 0).prod(0).mean()


def image_metrics(rgb: torch.Tensor, target: torch.Tensor) -> Dict[str, torch.Tensor]:
  """mse / psnr (/ multiscale ssim when the frame is large enough for 5 scales) of a rendered frame against its
  target (eval.py:121-128)."""
  target = target.to(rgb.device)
  mse = ((rgb - target) ** 2).mean()
  out = {'mse': mse, 'psnr': compute_psnr(mse)}
  if min(rgb.shape[:2]) >= 176:
    out['ssim'] = compute_multiscale_ssim(target, rgb)
  return out
