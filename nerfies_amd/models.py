"""Host-side mirror of nerfies.models (reference: nerfies/models.py:31-489).

`NerfModel.apply` keeps the reference signature (models.py:289-299); the work is done by the HIP
library through the C-ABI (include/nerfies_amd.h).  Differences forced by the missing JAX runtime:
  * `rngs={'coarse': k0, 'fine': k1}`: a key is either an int seed (on-device Philox) or a float
    tensor of explicit uniforms, (B,N_c) for 'coarse' and (B,N_f) for 'fine' (parity runs);
  * arrays are torch tensors on the GPU; params are a `FlatParams` (fast path) or a flax-style
    nested dict (copied into a flat buffer per call).
"""
import ctypes as C
import os
import dataclasses
from typing import Any, Dict, Mapping, Optional, Sequence, Tuple

import torch

from nerfies_amd import lib as L
from nerfies_amd import params as P


def _ptr(t: Optional[torch.Tensor]):
  return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, device):
  return torch.as_tensor(t, device=device).to(torch.float32).contiguous()


def _ids(t, device):
  if t is None:
    return None
  t = torch.as_tensor(t, device=device)
  if t.dim() == 2 and t.shape[-1] == 1:   # glo.py:50-51 squeezes the trailing 1
    t = t[..., 0]
  return t.to(torch.int32).contiguous()


def _scalars(warp_extra, dynamic: Optional[torch.Tensor] = None) -> 'L.StepScalars':
  """nrf_step_scalars of warp_extra; `dynamic`: the device buffer of a training.DynamicScalars (graph-replayable step)."""
  we = warp_extra or {}
  return L.StepScalars(float(we.get('alpha', 0.0)), float(we.get('time_alpha', 0.0)), None if dynamic is None else dynamic.data_ptr())


def rng_seed_of(coarse_key: int, fine_key: int) -> int:
  """The Philox seed NerfModel._rand_struct derives from integer 'coarse' / 'fine' keys (what a device-resident
  nrf_dynamic_scalars.rng_seed must hold to reproduce the eager call)."""
  seed = 0
  for k in (coarse_key, fine_key):
    seed = (seed * 0x9E3779B97F4A7C15 + int(k)) & 0xFFFFFFFFFFFFFFFF
  return seed


def _act_name(a):
  if isinstance(a, str):
    return a
  name = getattr(a, '__name__', str(a))
  for k in ('softplus', 'relu'):
    if k in name:
      return k
  raise ValueError(f'unsupported activation {a!r}')


@dataclasses.dataclass
class NerfModel:
  """Attributes mirror nerfies.models.NerfModel (models.py:75-119)."""
  num_coarse_samples: int
  num_fine_samples: int
  use_viewdirs: bool
  near: float
  far: float
  noise_std: Optional[float]
  nerf_trunk_depth: int
  nerf_trunk_width: int
  nerf_rgb_branch_depth: int
  nerf_rgb_branch_width: int
  nerf_skips: Tuple[int, ...]
  alpha_channels: int
  rgb_channels: int
  use_stratified_sampling: bool
  num_nerf_point_freqs: int
  num_nerf_viewdir_freqs: int
  appearance_ids: Sequence[int]
  camera_ids: Sequence[int]
  warp_ids: Sequence[int]
  num_appearance_features: int
  num_camera_features: int
  num_warp_features: int
  num_warp_freqs: int
  activation: Any = 'relu'
  sigma_activation: Any = 'relu'
  use_white_background: bool = False
  use_linear_disparity: bool = False
  use_sample_at_infinity: bool = True
  warp_field_type: str = 'se3'
  warp_metadata_encoder_type: str = 'glo'
  use_appearance_metadata: bool = False
  use_camera_metadata: bool = False
  use_warp: bool = False
  use_warp_jacobian: bool = False
  use_weights: bool = False
  use_trunk_condition: bool = False
  use_alpha_condition: bool = False
  use_rgb_condition: bool = False
  warp_kwargs: Mapping[str, Any] = dataclasses.field(default_factory=dict)
  metadata_encoded: bool = False

  def __post_init__(self):
    self._handle = None
    self._layout = None
    self._ws = {}
    self._lib = None

  # models.py:121-131
  @property
  def num_appearance_embeddings(self):
    return max(self.appearance_ids) + 1

  @property
  def num_warp_embeddings(self):
    return max(self.warp_ids) + 1

  @property
  def num_camera_embeddings(self):
    return max(self.camera_ids) + 1

  # ---- C-ABI plumbing -------------------------------------------------------------------
  def desc(self) -> L.ModelDesc:
    if _act_name(self.activation) != 'relu':
      raise L.NrfError('only relu trunk activation is built')
    if self.alpha_channels != 1 or self.rgb_channels != 3:
      raise L.NrfError('alpha_channels/rgb_channels must be 1/3')
    skips = tuple(self.nerf_skips)
    if len(skips) > 1:
      raise L.NrfError('at most one skip layer')
    d = L.ModelDesc()
    d.num_coarse_samples = self.num_coarse_samples
    d.num_fine_samples = self.num_fine_samples
    d.use_viewdirs = int(self.use_viewdirs)
    d.near_plane = float(self.near)
    d.far_plane = float(self.far)
    d.nerf_trunk_depth = self.nerf_trunk_depth
    d.nerf_trunk_width = self.nerf_trunk_width
    d.nerf_rgb_branch_depth = self.nerf_rgb_branch_depth
    d.nerf_rgb_branch_width = self.nerf_rgb_branch_width
    d.nerf_skip_layer = skips[0] if skips else -1
    d.use_stratified_sampling = int(self.use_stratified_sampling)
    d.num_nerf_point_freqs = self.num_nerf_point_freqs
    d.num_nerf_viewdir_freqs = self.num_nerf_viewdir_freqs
    d.sigma_activation = L.ACT[_act_name(self.sigma_activation)]
    d.use_white_background = int(self.use_white_background)
    d.use_linear_disparity = int(self.use_linear_disparity)
    d.use_sample_at_infinity = int(self.use_sample_at_infinity)
    d.use_appearance_metadata = int(self.use_appearance_metadata)
    d.num_appearance_embeddings = self.num_appearance_embeddings if self.use_appearance_metadata else 0
    d.num_appearance_features = self.num_appearance_features
    d.use_camera_metadata = int(self.use_camera_metadata)
    d.num_camera_embeddings = self.num_camera_embeddings if self.use_camera_metadata else 0
    d.num_camera_features = self.num_camera_features
    d.use_alpha_condition = int(self.use_alpha_condition)
    d.use_rgb_condition = int(self.use_rgb_condition)
    d.use_trunk_condition = int(self.use_trunk_condition)
    if self.use_warp and self.warp_field_type not in L.WARP_FIELD:
      raise L.NrfError(f"warp_field_type must be one of {sorted(L.WARP_FIELD)} (warping.py:36-44)")
    wdepth, wwidth = self._warp_trunk_shape()
    if self.use_warp and self.warp_metadata_encoder_type not in L.META_ENCODER:
      raise L.NrfError(f"warp_metadata_encoder_type must be one of {sorted(L.META_ENCODER)} ('blend' exists only in the "
                       'TranslationField, warping.py:142-146, and no preset selects it)')
    d.use_warp = int(self.use_warp)
    d.num_warp_freqs = self.num_warp_freqs
    d.num_warp_embeddings = self.num_warp_embeddings if self.use_warp else 0
    d.num_warp_features = self.num_warp_features
    d.warp_field_type = L.WARP_FIELD[self.warp_field_type] if self.use_warp else 0
    d.noise_std = float(self.noise_std or 0.0)
    d.warp_metadata_encoder_type = L.META_ENCODER[self.warp_metadata_encoder_type] if self.use_warp else 0
    d.num_time_encoder_freqs = 1   # metadata_encoder_num_freqs (warping.py:234): the field's default (_warp_trunk_shape refuses others)
    d.warp_trunk_depth, d.warp_trunk_width = (wdepth, wwidth) if self.use_warp else (0, 0)
    return d

  # ModelConfig.warp_kwargs (configs.py:105) are passed through create_warp_field to the field's constructor
  # (models.py:165-184, warping.py:29-59).  Built: the trunk's depth (<= 6) and width (<= 128) -- SE3Field trunk_depth / trunk_width
  # (warping.py:225-226), TranslationField depth / hidden_channels (warping.py:90-91).  Every other attribute must keep the field's
  # default: the kernels have the branches as bare 3-channel output layers (rotation / pivot / translation depth 0), no pivot or
  # translation branch, relu, skips (4,), frequencies 2^0 .. 2^(F-1) with the identity map.
  _WARP_KW_DEFAULTS = {
      'se3': dict(skips=(4,), rotation_depth=0, pivot_depth=0, translation_depth=0, use_pivot=False, use_translation=False,
                  min_freq_log2=0, max_freq_log2=None, use_identity_map=True, metadata_encoder_num_freqs=1),
      'translation': dict(skips=(4,), min_freq_log2=0, max_freq_log2=None, use_identity_map=True, metadata_encoder_num_freqs=1),
  }
  _WARP_KW_TRUNK = {'se3': ('trunk_depth', 'trunk_width'), 'translation': ('depth', 'hidden_channels')}

  def _warp_trunk_shape(self):
    kw = dict(self.warp_kwargs or {})
    if not self.use_warp or not kw:
      return 0, 0
    ft = self.warp_field_type
    dk, wk = self._WARP_KW_TRUNK.get(ft, (None, None))
    depth, width = int(kw.pop(dk, 6)), int(kw.pop(wk, 128))
    if not (1 <= depth <= 6 and 1 <= width <= 128):
      raise L.NrfError(f'warp_kwargs: {dk} must be in [1, 6] and {wk} in [1, 128] (the warp kernels are 6 x 128; got {depth} x {width})')
    # widths of branches that do not exist at depth 0 are inert (modules.MLP builds no hidden layer), as is the activation name relu
    for inert in ('rotation_width', 'pivot_width', 'translation_width'):
      kw.pop(inert, None)
    if 'activation' in kw and _act_name(kw['activation']) == 'relu':
      kw.pop('activation')
    if 'metadata_encoder_type' in kw and kw['metadata_encoder_type'] == self.warp_metadata_encoder_type:
      kw.pop('metadata_encoder_type')
    for k, dflt in self._WARP_KW_DEFAULTS.get(ft, {}).items():
      if k in kw and (tuple(kw[k]) if k == 'skips' else kw[k]) == dflt:
        kw.pop(k)
    if kw:
      raise L.NrfError(f'warp_kwargs {kw} are not supported: built are {dk} <= 6 and {wk} <= 128; every other attribute of the '
                       f'{ft} field must keep its default (warping.py:216-243)')
    return depth, width


  @property
  def lib(self):
    if self._lib is None:
      self._lib = L.load_library()
    return self._lib

  @property
  def handle(self):
    if self._handle is None:
      h = C.c_void_p()
      d = self.desc()
      L.check(self.lib.nrf_create(C.byref(d), C.byref(h)), self.lib)
      self._handle = h
      rows = os.environ.get('NRF_CHAIN_TILE_ROWS')   # experiments / the A-B tests: 32 or 64 for every model of the process
      if rows:
        self.set_chain_tile_rows(int(rows))
      if os.environ.get('NRF_BF16_WGRAD_MERGE'):   # experiments / the A-B record: merged bf16 wgrad groups (nerfies_amd.h)
        L.check(self.lib.nrf_set_option(h, L.NRF_OPT_BF16_WGRAD_MERGE, int(os.environ['NRF_BF16_WGRAD_MERGE'])), self.lib)
    return self._handle

  def set_chain_tile_rows(self, rows: int):
    """NRF_OPT_CHAIN_TILE_ROWS: rows per workgroup tile of the float32 NeRF chain kernels -- 64 (two workgroups per CU), 32
    (four per CU) or 0 = automatic (the default).  A tuning knob without a reference counterpart; forward results are bit-identical
    under both tilings, gradients agree to the order of float atomics."""
    L.check(self.lib.nrf_set_option(self.handle, L.NRF_OPT_CHAIN_TILE_ROWS, int(rows)), self.lib)
    self._drop_workspaces()

  def set_bf16_wgrad_merge(self, on: bool):
    """NRF_OPT_BF16_WGRAD_MERGE: merged weight-gradient groups of the bf16 training mode (default on).  The option changes the
    training workspace's SIZE and layout, so the cached workspaces (and the stash a pending `backward` would read) are dropped:
    a live model picks the option up at its next step instead of running the new plan in a buffer sized for the old one."""
    L.check(self.lib.nrf_set_option(self.handle, L.NRF_OPT_BF16_WGRAD_MERGE, int(bool(on))), self.lib)
    self._drop_workspaces()

  def _drop_workspaces(self):
    self._ws = {}
    self._train_ws = None

  @property
  def layout(self) -> P.ParamLayout:
    if self._layout is None:
      n = C.c_int32(0)
      L.check(self.lib.nrf_param_layout(self.handle, None, C.byref(n)), self.lib)
      infos = (L.TensorInfo * n.value)()
      L.check(self.lib.nrf_param_layout(self.handle, infos, C.byref(n)), self.lib)
      total = C.c_int64(0)
      L.check(self.lib.nrf_param_count(self.handle, C.byref(total)), self.lib)
      self._layout = P.layout_from_infos(infos, total.value)
    return self._layout

  @staticmethod
  def bf16_flags(bf16) -> int:
    """bf16=True: NRF_FLAG_BF16 (NeRF MLPs AND the SE3 trunk on bfloat16 operands); bf16='mlp': the NeRF MLPs only
    (NRF_FLAG_WARP_F32: the round-3 behaviour); bf16='x3' (inference only): NRF_FLAG_BF16X3, the NeRF MLPs and the SE3 trunk in split-bf16
    (float32-emulating) arithmetic ('x3mlp': the trunk stays float32); False: float32."""
    if not bf16:
      return 0
    if bf16 == 'x3':
      return L.NRF_FLAG_BF16X3
    if bf16 == 'x3mlp':   # split-bf16 NeRF MLPs, float32 SE3 trunk (bit-identical warped points)
      return L.NRF_FLAG_BF16X3 | L.NRF_FLAG_WARP_F32
    if bf16 == 'mlp':
      return L.NRF_FLAG_BF16 | L.NRF_FLAG_WARP_F32
    return L.NRF_FLAG_BF16

  def workspace(self, num_rays: int, train: bool, device, num_background_points: int = 0, elastic: bool = False,
                jacobian: bool = False, bf16=False) -> torch.Tensor:
    # the TRAINING layout depends on it (bf16 stashes instead of the fp32 ones); so does an 'x3' inference plan (its weight streams)
    bf16 = 'x3' if bf16 in ('x3', 'x3mlp') else (bf16 if train else False)
    key = (int(num_rays), bool(train), str(device), int(num_background_points), bool(elastic), bool(jacobian), bf16)
    ws = self._ws.get(key)
    if ws is None:
      nbytes = C.c_size_t(0)
      flags = (L.NRF_FLAG_TRAIN if train else 0) | (L.NRF_FLAG_WARP_JACOBIAN if jacobian else 0) | self.bf16_flags(bf16)
      L.check(self.lib.nrf_workspace_bytes_ex(self.handle, num_rays, flags,
                                              int(num_background_points), int(bool(elastic)), C.byref(nbytes)), self.lib)
      ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=device)
      self._ws[key] = ws
    return ws

  def flat_params(self, variables, device) -> P.FlatParams:
    params = variables['params'] if isinstance(variables, dict) and 'params' in variables else variables
    if isinstance(params, P.FlatParams):
      return params
    if isinstance(params, dict) and 'model' in params and 'nerf_mlps_coarse' not in params:
      params = params['model']
    return P.FlatParams(P.flat_from_tree(params, self.layout, device), self.layout)

  def _rays_struct(self, rays_dict, device, metadata_encoded=False):
    origins = _f32(rays_dict['origins'], device)
    directions = _f32(rays_dict['directions'], device)
    viewdirs = _f32(rays_dict['viewdirs'], device) if 'viewdirs' in rays_dict else None
    md = rays_dict.get('metadata', {}) or {}
    keep = [origins, directions, viewdirs]
    r = L.Rays()
    r.num_rays = origins.shape[0]
    r.origins, r.directions, r.viewdirs = _ptr(origins), _ptr(directions), _ptr(viewdirs)
    time_enc = self.use_warp and self.warp_metadata_encoder_type == 'time'
    for field, cfield, key, width in (('warp_ids', 'warp_codes', 'warp', self.num_warp_features),
                                      ('appearance_ids', 'appearance_codes', 'appearance', self.num_appearance_features),
                                      ('camera_ids', 'camera_codes', 'camera', self.num_camera_features)):
      src = md.get('time' if (key == 'warp' and time_enc) else key)   # models.py:252-254
      if src is None:
        continue
      if metadata_encoded:   # the codes themselves, (B, features) (models.py:198-199, 210-211, 251)
        t = _f32(src, device).reshape(r.num_rays, -1)
        if t.shape[1] != width:
          raise L.NrfError(f"metadata_encoded: metadata[{key!r}] must have {width} features, got {t.shape[1]}")
        setattr(r, cfield, _ptr(t))
      elif key == 'warp' and time_enc:   # float timestamps for the TimeEncoder (datasets/core.py:272-274)
        t = _f32(src, device).reshape(-1)
        r.time = _ptr(t)
      else:
        t = _ids(src, device)
        setattr(r, field, _ptr(t))
      keep.append(t)
    return r, keep

  def _rand_struct(self, rngs, num_rays, device):
    rnd = L.Rand()
    keep = []
    rngs = rngs or {}
    for field, key, n in (('noise_coarse', 'noise_coarse', self.num_coarse_samples),
                          ('noise_fine', 'noise_fine', self.num_coarse_samples + self.num_fine_samples)):
      k = rngs.get(key)   # explicit standard normals for noise_regularize (parity runs)
      if k is not None:
        t = _f32(k, device)
        if tuple(t.shape) != (num_rays, n):
          raise L.NrfError(f"rngs[{key!r}] normals must have shape {(num_rays, n)}")
        keep.append(t)
        setattr(rnd, field, _ptr(t))
    for field, key, n in (('t_rand', 'coarse', self.num_coarse_samples), ('u', 'fine', self.num_fine_samples)):
      k = rngs.get(key)
      if isinstance(k, torch.Tensor) and k.is_floating_point() and k.dim() == 2:
        if tuple(k.shape) != (num_rays, n):
          raise L.NrfError(f"rngs[{key!r}] uniforms must have shape {(num_rays, n)}")
        t = _f32(k, device)
        keep.append(t)
        setattr(rnd, field, _ptr(t))
      elif k is not None:
        seed = int(k.item()) if isinstance(k, torch.Tensor) else int(k)
        rnd.seed = (rnd.seed * 0x9E3779B97F4A7C15 + seed) & 0xFFFFFFFFFFFFFFFF
    return rnd, keep

  # ---- NerfModel.__call__ (models.py:289-375) ---------------------------------------------
  def apply(self, variables, rays_dict: Dict[str, Any], warp_extra: Dict[str, Any] = None, metadata_encoded=False,
            use_warp=True, return_points=False, return_weights=False, return_warp_jacobian=False,
            deterministic=False, rngs=None, *, train=False, return_z_vals=False, out=None, bf16=False):
    """Returns {'coarse': {...}, 'fine': {...}} like the reference.  `train=True` keeps the
    activation stash so `backward` can follow (used by training.train_step / autograd).  `out`: a dict
    returned by an earlier call with the same shapes/flags, to be overwritten in place (fixed output
    addresses: what a captured hipGraph replay needs).  `bf16=True` (inference only, no reference counterpart): the NeRF
    MLPs take bfloat16 operands (NRF_FLAG_BF16), everything else stays fp32."""
    del deterministic   # accepted and unused, as in the reference (models.py:298)
    warp_on = bool(self.use_warp and use_warp)
    # models.py:345-346, 367-368: the coarse level carries the Jacobian when the model was built with use_warp_jacobian
    # or the call asks for it, the fine level only when the call asks for it
    jac_levels = set()
    if warp_on and not train:
      if return_warp_jacobian or self.use_warp_jacobian:
        jac_levels.add('coarse')
      if return_warp_jacobian:
        jac_levels.add('fine')
    if train and metadata_encoded:
      raise L.NrfError('metadata_encoded=True is an inference input (the reference never trains with it)')
    device = torch.as_tensor(rays_dict['origins']).device
    if device.type != 'cuda':
      raise L.NrfError('rays must live on the GPU: the hot path has no CPU fallback')
    fp = self.flat_params(variables, device)
    rays, keep = self._rays_struct(rays_dict, device, metadata_encoded)
    B = rays.num_rays
    rnd, keep2 = self._rand_struct(rngs, B, device)
    return_weights = self.use_weights or return_weights
    S = (self.num_coarse_samples, self.num_coarse_samples + self.num_fine_samples)
    reuse = out
    out = L.Outputs()
    ret = {}
    levels = [('coarse', out.coarse, S[0])] + ([('fine', out.fine, S[1])] if self.num_fine_samples > 0 else [])
    for name, lo, s in levels:
      if reuse is not None:
        d = reuse[name]
      else:
        d = {'rgb': torch.empty(B, 3, device=device), 'depth': torch.empty(B, device=device),
             'med_depth': torch.empty(B, device=device), 'acc': torch.empty(B, device=device)}
        if return_weights:
          d['weights'] = torch.empty(B, s, device=device)
        if return_z_vals:   # extra (not in the reference dict): the sample depths of this level
          d['z_vals'] = torch.empty(B, s, device=device)
        if return_points:   # models.py:247-248 (always), 266-267 (only behind the warp field)
          d['points'] = torch.empty(B, s, 3, device=device)
          if warp_on:
            d['warped_points'] = torch.empty(B, s, 3, device=device)
        if name in jac_levels:   # models.py:264-265
          d['warp_jacobian'] = torch.empty(B, s, 3, 3, device=device)
      for k, t in d.items():
        setattr(lo, k, _ptr(t))
      ret[name] = d
    scal = _scalars(warp_extra)
    ws = self.workspace(B, train, device, jacobian=bool(jac_levels), bf16=bf16)
    if train:
      self._train_ws = (B, ws)   # the stash `backward` differentiates (fp32 or bf16 layout), and the batch size it is for
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    flags = (L.NRF_FLAG_TRAIN if train else 0) | (L.NRF_FLAG_NO_WARP if self.use_warp and not warp_on else 0) | \
        self.bf16_flags(bf16) | (L.NRF_FLAG_WARP_JACOBIAN if jac_levels else 0)
    L.check(self.lib.nrf_forward(self.handle, _ptr(fp.flat), C.byref(rays), C.byref(scal), C.byref(rnd), C.byref(out),
                                 flags, _ptr(ws), ws.numel() * 4, stream), self.lib)
    del keep, keep2
    return ret

  def backward(self, variables, rays_dict, d_rgb_coarse, d_rgb_fine, grad_out: torch.Tensor = None):
    """VJP of the last `apply(..., train=True)` on the same rays: returns the flat parameter gradient."""
    device = torch.as_tensor(rays_dict['origins']).device
    fp = self.flat_params(variables, device)
    rays, keep = self._rays_struct(rays_dict, device)
    grad = grad_out if grad_out is not None else torch.empty_like(fp.flat)
    dc = None if d_rgb_coarse is None else _f32(d_rgb_coarse, device)
    df = None if d_rgb_fine is None else _f32(d_rgb_fine, device)
    stash = getattr(self, '_train_ws', None)
    if stash is None:
      raise L.NrfError('backward() needs a preceding apply(..., train=True)')
    if stash[0] != rays.num_rays:
      raise L.NrfError(f'backward(): the stashed forward was run on {stash[0]} rays, not {rays.num_rays}')
    ws = stash[1]   # the library also refuses a stash whose workspace plan was replaced by another call (NRF_E_STATE)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    L.check(self.lib.nrf_backward(self.handle, _ptr(fp.flat), C.byref(rays), _ptr(dc), _ptr(df), _ptr(grad), _ptr(ws),
                                  ws.numel() * 4, stream), self.lib)
    del keep
    return grad

  def loss_and_grad(self, fp: P.FlatParams, batch, warp_extra=None, rngs=None, grad_out=None, stats_out=None,
                    background=None, elastic=None, warp_reg=None, bf16=False, dynamic=None):
    """forward + MSE_coarse + MSE_fine [+ background regulariser] + backward in one library call
    (training.py:168-265).  `background` = dict(points (N,3) already noised, warp_ids (N,), weight, alpha=-2,
    scale=1e-3) adds weight * mean(general_loss(|warp(x) - x|^2)) (training.py:117-135, 248-259).
    Instead of 'warp_ids' the dict may carry `id_choices` (the model's warp ids) and `noise_std`: the library then draws the id
    per point and adds the noise itself (training.py:121-126), `points` being the raw points.  `dynamic`: the device buffer
    of a training.DynamicScalars -- warp_alpha / time_alpha / the rng key / the elastic weight are read from it on the device
    (a captured launch sequence stays valid when they change).
    `elastic` = dict(weight, reduce_method='weight', eps=1e-6, alpha=-2, scale=0.03) adds the elastic regulariser
    on the coarse samples (training.py:71-114, 177-197); `loss_type` in lib.ELASTIC_TYPE.  `warp_reg` = dict(weight,
    alpha=-2, scale=1e-3): training.py:199-212 on both levels.  `bf16`: bfloat16 MLP operands (NRF_FLAG_BF16).
    stats (lib.NRF_NUM_STATS floats) = [mse_c, mse_f, psnr_c, psnr_f, total, background_loss, loss/elastic,
    residual/elastic, warp_reg_c, warp_reg_f, warp_reg residual c, f, jacobian det, div, curl, 0]."""
    device = fp.flat.device
    rays, keep = self._rays_struct(batch, device)
    rnd, keep2 = self._rand_struct(rngs, rays.num_rays, device)
    target = _f32(batch['rgb'], device)[..., :3].contiguous()
    grad = grad_out if grad_out is not None else torch.empty_like(fp.flat)
    stats = stats_out if stats_out is not None else torch.empty(L.NRF_NUM_STATS, device=device)
    scal = _scalars(warp_extra, dynamic)
    bg, nbg, keep3 = None, 0, []
    if background is not None:
      pts = _f32(background['points'], device).reshape(-1, 3)
      nbg = pts.shape[0]
      if background.get('warp_ids') is not None:   # the caller drew ids and noise (parity runs)
        ids = _ids(background['warp_ids'], device).reshape(-1)
        keep3 = [pts, ids]
        bg = L.Background(nbg, _ptr(pts), _ptr(ids), float(background.get('weight', 1.0)), float(background.get('alpha', -2.0)),
                          float(background.get('scale', 0.001)), None, 0, 0.0)
      else:                                        # the library draws them (training.py:121-126)
        choices = _ids(background['id_choices'], device).reshape(-1)
        keep3 = [pts, choices]
        bg = L.Background(nbg, _ptr(pts), None, float(background.get('weight', 1.0)), float(background.get('alpha', -2.0)),
                          float(background.get('scale', 0.001)), _ptr(choices), choices.numel(), float(background.get('noise_std', 0.001)))
    el = None
    if elastic is not None:
      method = elastic.get('reduce_method', 'weight')
      if method not in L.ELASTIC_REDUCE:
        raise L.NrfError(f'unknown elastic_reduce_method {method!r}')
      ltype = elastic.get('loss_type', 'log_svals')
      if ltype not in L.ELASTIC_TYPE:
        raise L.NrfError(f"elastic_loss_type {ltype!r} is not built (one of {sorted(L.ELASTIC_TYPE)}; 'nr' produces NaNs in "
                         'the reference itself, training.py:58)')
      el = L.Elastic(float(elastic.get('weight', 0.0)), L.ELASTIC_REDUCE[method], float(elastic.get('eps', 1e-6)),
                     float(elastic.get('alpha', -2.0)), float(elastic.get('scale', 0.03)), L.ELASTIC_TYPE[ltype])
    wr = None
    if warp_reg is not None:
      wr = L.WarpReg(float(warp_reg.get('weight', 0.0)), float(warp_reg.get('alpha', -2.0)), float(warp_reg.get('scale', 0.001)))
    ws = self.workspace(rays.num_rays, True, device, nbg, el is not None, bf16=bf16)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    L.check(self.lib.nrf_train_step_loss_grad_ex(self.handle, _ptr(fp.flat), C.byref(rays), _ptr(target), C.byref(scal),
                                                 C.byref(rnd), C.byref(bg) if bg is not None else None,
                                                 C.byref(el) if el is not None else None,
                                                 C.byref(wr) if wr is not None else None, self.bf16_flags(bf16),
                                                 _ptr(grad), _ptr(stats), _ptr(ws), ws.numel() * 4, stream), self.lib)
    del keep, keep2, keep3
    return grad, stats

  def warp_points(self, variables, points, warp_ids, warp_extra):
    """model.create_warp_field(model, num_batch_dims=1).apply(points, ids, warp_extra, False, False)
    ['warped_points'] (models.py:165-184, warping.py:355-389): (N,3) points, one warp id per point."""
    device = torch.as_tensor(points).device
    fp = self.flat_params(variables, device)
    pts = _f32(points, device).reshape(-1, 3)
    ids = _ids(warp_ids, device).reshape(-1)
    n = pts.shape[0]
    nbytes = C.c_size_t(0)
    L.check(self.lib.nrf_warp_points_workspace_bytes(self.handle, n, C.byref(nbytes)), self.lib)
    ws = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=device)
    out = torch.empty(n, 3, device=device)
    scal = _scalars(warp_extra)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    L.check(self.lib.nrf_warp_points(self.handle, _ptr(fp.flat), _ptr(pts), _ptr(ids), n, C.byref(scal), _ptr(out), _ptr(ws),
                                     ws.numel() * 4, stream), self.lib)
    return out.reshape(torch.as_tensor(points).shape)

  def profile_enable(self, on=True):
    L.check(self.lib.nrf_profile_enable(self.handle, int(bool(on))), self.lib)

  def profile_read(self):
    """[{name, ms, launches, flops_per_launch}] accumulated since the last read (device-side HIP events)."""
    n = C.c_int32(0)
    L.check(self.lib.nrf_profile_read(self.handle, None, C.byref(n)), self.lib)
    arr = (L.ProfileEntry * max(n.value, 1))()
    L.check(self.lib.nrf_profile_read(self.handle, arr, C.byref(n)), self.lib)
    return [dict(name=arr[i].name.decode(), ms=arr[i].ms, launches=arr[i].launches,
                 flops_per_launch=arr[i].flops_per_launch) for i in range(n.value)]

  def __del__(self):
    try:
      if self._handle is not None and self._lib is not None:
        self._lib.nrf_destroy(self._handle)
    except Exception:   # interpreter shutdown
      pass


def construct_nerf(key, config, batch_size: int, appearance_ids: Sequence[int], camera_ids: Sequence[int],
                   warp_ids: Sequence[int], near: float, far: float, use_warp_jacobian: bool = False,
                   use_weights: bool = False, device='cuda'):
  """models.construct_nerf (models.py:378-489): builds the model and reference-initialised params.

  `key` is an int seed.  `config` is any object with the ModelConfig attributes (configs.py:35-105).
  `batch_size` is accepted for signature parity (shape inference is not needed here)."""
  del batch_size
  g = lambda name, default=None: getattr(config, name, default)
  model = NerfModel(
      num_coarse_samples=g('num_coarse_samples', 64), num_fine_samples=g('num_fine_samples', 128),
      use_viewdirs=g('use_viewdirs', True), near=near, far=far, noise_std=g('noise_std'),
      nerf_trunk_depth=g('nerf_trunk_depth', 8), nerf_trunk_width=g('nerf_trunk_width', 256),
      nerf_rgb_branch_depth=g('nerf_rgb_branch_depth', 1), nerf_rgb_branch_width=g('nerf_rgb_branch_width', 128),
      nerf_skips=tuple(g('nerf_skips', (4,))), alpha_channels=g('alpha_channels', 1), rgb_channels=g('rgb_channels', 3),
      use_stratified_sampling=g('use_stratified_sampling', True), num_nerf_point_freqs=g('num_nerf_point_freqs', 10),
      num_nerf_viewdir_freqs=g('num_nerf_viewdir_freqs', 4), appearance_ids=appearance_ids, camera_ids=camera_ids,
      warp_ids=warp_ids, num_appearance_features=g('appearance_metadata_dims', 8),
      num_camera_features=g('camera_metadata_dims', 2), num_warp_features=g('num_warp_features', 8),
      num_warp_freqs=g('num_warp_freqs', 8), activation=g('activation', 'relu'),
      sigma_activation=g('sigma_activation', 'relu'), use_white_background=g('use_white_background', False),
      use_linear_disparity=g('use_linear_disparity', False), use_sample_at_infinity=g('use_sample_at_infinity', True),
      warp_field_type=g('warp_field_type', 'translation'),
      warp_metadata_encoder_type=g('warp_metadata_encoder_type', 'glo'),
      use_appearance_metadata=g('use_appearance_metadata', False), use_camera_metadata=g('use_camera_metadata', False),
      use_warp=g('use_warp', False), use_warp_jacobian=use_warp_jacobian, use_weights=use_weights,
      use_trunk_condition=g('use_trunk_condition', False),
      use_alpha_condition=g('use_alpha_condition', False), use_rgb_condition=g('use_rgb_condition', False),
      warp_kwargs=dict(g('warp_kwargs', {}) or {}))
  flat = P.init_flat(model.layout, int(key), device)
  return model, P.FlatParams(flat, model.layout)
