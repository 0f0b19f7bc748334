#!/usr/bin/env python
"""Golden vectors from the REAL reference sources (google/nerfies at /root/reference), executed on NumPy
float64 through the import-name stand-ins of oracle/_shim (jax.numpy -> numpy, a minimal eager flax.linen,
jax.random -> caller-supplied arrays, jax.jacfwd -> central differences; see oracle/_shim/README.md).

Runs only where /root/reference exists (the build container); writes tests/golden/ref_*.npz, which
tests/test_reference_vectors.py replays against oracle/nerfies_oracle.py on any machine.  No reference
source is copied: the modules are imported from /root/reference as they lie.

  python tests/golden/make_reference_vectors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path[:0] = [os.path.join(ROOT, 'oracle', '_shim'), REF, ROOT]

import torch  # noqa: E402

from jax import random as jrandom  # noqa: E402  (the shim)
from nerfies import camera as ref_camera  # noqa: E402
from nerfies import model_utils as ref_mu  # noqa: E402
from nerfies import models as ref_models  # noqa: E402
from nerfies import modules as ref_modules  # noqa: E402
from nerfies import rigid_body as ref_rigid  # noqa: E402
from nerfies import schedules as ref_sched  # noqa: E402
from nerfies import training as ref_training  # noqa: E402
from nerfies import utils as ref_utils  # noqa: E402
from nerfies import warping as ref_warping  # noqa: E402
from flax import linen as nn  # noqa: E402  (the shim)

from oracle import nerfies_oracle as O  # noqa: E402  (only for parameter trees and synthetic batches)


# jnp arrays are immutable: `weights += eps` (model_utils.py:156) REBINDS the local name under JAX, whereas NumPy
# would add eps into the caller's array (the coarse weights the model returns).  The only in-place statement on the
# path is given JAX semantics by handing that function a private copy; the reference source itself is untouched.
_pdf = ref_mu.piecewise_constant_pdf
ref_mu.piecewise_constant_pdf = lambda key, bins, weights, *a, **k: _pdf(key, bins, np.array(weights, copy=True), *a, **k)


def tree_np(t):
  return {k: tree_np(v) for k, v in t.items()} if isinstance(t, dict) else np.asarray(t.detach().numpy() if torch.is_tensor(t) else t)


def save(name, **arrays):
  path = os.path.join(HERE, f'ref_{name}.npz')
  np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
  print(f'ref_{name}.npz: {len(arrays)} arrays, {os.path.getsize(path)} bytes')


def rigid_body():
  rng = np.random.default_rng(1)
  S, th, T = [], [], []
  for _ in range(16):
    w = rng.normal(size=3); w /= np.linalg.norm(w)
    v = rng.normal(size=3)
    theta = rng.uniform(1e-3, 2.5)
    S.append(np.concatenate([w, v])); th.append(theta)
    T.append(ref_rigid.exp_se3(S[-1], theta))
  w = rng.normal(size=3)
  save('rigid_body', screw=np.stack(S), theta=np.array(th), exp_se3=np.stack(T), skew_in=w, skew=ref_rigid.skew(w),
       exp_so3=ref_rigid.exp_so3(S[0][:3], th[0]))


def model_utils():
  rng = np.random.default_rng(2)
  B, Nc, Nf = 5, 12, 9
  o = rng.uniform(-0.5, 0.5, (B, 3)); d = rng.normal(size=(B, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
  t_rand = rng.uniform(0, 1, (B, Nc)); u = rng.uniform(0, 1, (B, Nf))
  out = dict(origins=o, directions=d, t_rand=t_rand, u=u, near=0.1, far=1.7)
  for strat in (0, 1):
    for lind in (0, 1):
      z, pts = ref_mu.sample_along_rays(jrandom.Key(uniform=t_rand), o, d, Nc, 0.1, 1.7, bool(strat), bool(lind))
      out[f'sample_z_s{strat}_l{lind}'] = z; out[f'sample_pts_s{strat}_l{lind}'] = pts
  z = out['sample_z_s1_l0']
  rgb = rng.uniform(0, 1, (B, Nc, 3)); sigma = rng.uniform(0, 30, (B, Nc))
  out.update(vr_rgb=rgb, vr_sigma=sigma, vr_z=z)
  for white in (0, 1):
    for inf in (0, 1):
      r = ref_mu.volumetric_rendering(rgb, sigma, z, d, use_white_background=bool(white), sample_at_infinity=bool(inf),
                                      return_weights=True)
      for k, v in r.items():
        out[f'vr_w{white}_i{inf}_{k}'] = v
  w = ref_mu.volumetric_rendering(rgb, sigma, z, d, use_white_background=False, return_weights=True)['weights']
  out['depth_index'] = ref_mu.compute_depth_index(w); out['depth_map'] = ref_mu.compute_depth_map(w, z)
  out['opaqueness_mask'] = ref_mu.compute_opaqueness_mask(w)
  z_mid = .5 * (z[..., 1:] + z[..., :-1])
  for strat in (0, 1):
    zs = ref_mu.piecewise_constant_pdf(jrandom.Key(uniform=u), z_mid, w[..., 1:-1], Nf, bool(strat))
    zf, pf = ref_mu.sample_pdf(jrandom.Key(uniform=u), z_mid, w[..., 1:-1], o, d, z, Nf, bool(strat))
    out[f'pdf_z_s{strat}'] = zs; out[f'sample_pdf_z_s{strat}'] = zf; out[f'sample_pdf_pts_s{strat}'] = pf
  out['pdf_weights'] = w
  save('model_utils', **out)


def encoders_and_mlps():
  rng = np.random.default_rng(3)
  x = rng.uniform(-1, 1, (6, 3))
  out = dict(x=x)
  for F in (0, 4, 8):
    enc = ref_modules.SinusoidalEncoder(num_freqs=F)
    out[f'posenc_F{F}'] = np.stack([enc.apply({'params': {}}, xi) for xi in x])
  for alpha in (0.0, 2.5, 8.0):
    enc = ref_modules.AnnealedSinusoidalEncoder(num_freqs=8)
    out[f'annealed_a{alpha}'] = np.stack([enc.apply({'params': {}}, xi, alpha) for xi in x])
  out['window_a3.25'] = ref_modules.AnnealedSinusoidalEncoder.cosine_easing_window(0, 7, 8, 3.25)
  # NerfMLP with viewdir + camera condition (R = 27 + 2), skip at 4
  spec = O.ModelSpec(use_camera_metadata=True)
  p = tree_np(O.init_params(spec, seed=4, trained_like=True))['nerf_mlps_coarse']
  B, S = 3, 5
  pe = rng.normal(size=(B, S, spec.point_feat)); cond = rng.normal(size=(B, spec.rgb_cond_width))
  mlp = ref_modules.NerfMLP(trunk_depth=8, trunk_width=256, rgb_branch_depth=1, rgb_branch_width=128, skips=(4,))
  r = mlp.apply({'params': p}, pe, None, None, cond)
  out.update(mlp_in=pe, mlp_cond=cond, mlp_rgb=r['rgb'], mlp_alpha=r['alpha'])
  save('modules', **out)


def se3_field():
  rng = np.random.default_rng(5)
  spec = O.ModelSpec(use_warp=True, num_warp_freqs=6, num_warp_features=8, num_warp_embeddings=4)
  wp = tree_np(O.init_params(spec, seed=6, trained_like=True))['warp_field']
  field = ref_warping.SE3Field(num_freqs=6, num_embeddings=4, num_embedding_features=8)
  pts = rng.uniform(-0.5, 0.5, (7, 3)); ids = rng.integers(0, 4, (7, 1))
  outs = [field.apply({'params': wp}, pts[i], ids[i], {'alpha': 4.5, 'time_alpha': 0.0}, True, False) for i in range(7)]
  out = dict(points=pts, ids=ids, alpha=4.5, warped=np.stack([o['warped_points'] for o in outs]),
             jacobian_fd=np.stack([o['jacobian'] for o in outs]))
  save('se3_field', **out)


def translation_field():
  """warping.TranslationField (the ModelConfig dataclass default warp_field_type) on the oracle's parameters."""
  rng = np.random.default_rng(15)
  spec = O.ModelSpec(use_warp=True, warp_field_type='translation', num_warp_freqs=5, num_warp_features=8, num_warp_embeddings=4)
  wp = tree_np(O.init_params(spec, seed=16, trained_like=True))['warp_field']
  field = ref_warping.TranslationField(num_freqs=5, num_embeddings=4, num_embedding_features=8)
  pts = rng.uniform(-0.5, 0.5, (7, 3)); ids = rng.integers(0, 4, (7, 1))
  outs = [field.apply({'params': wp}, pts[i], ids[i], {'alpha': 3.25, 'time_alpha': 0.0}, True, False) for i in range(7)]
  save('translation_field', points=pts, ids=ids, alpha=3.25, warped=np.stack([o['warped_points'] for o in outs]),
       jacobian_fd=np.stack([o['jacobian'] for o in outs]))


NERF_CASES = {
    'nowarp': (dict(num_coarse_samples=10, num_fine_samples=7, num_nerf_point_freqs=6, use_stratified_sampling=True), 0.0),
    'camera': (dict(num_coarse_samples=8, num_fine_samples=8, num_nerf_point_freqs=4, use_stratified_sampling=False,
                    use_camera_metadata=True), 0.0),
    'warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                  num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True), 3.25),
}


def build_ref_model(spec, **extra):
  return ref_models.NerfModel(
      num_coarse_samples=spec.num_coarse_samples, num_fine_samples=spec.num_fine_samples, use_viewdirs=spec.use_viewdirs,
      near=spec.near, far=spec.far, noise_std=spec.noise_std, nerf_trunk_depth=spec.nerf_trunk_depth, nerf_trunk_width=256,
      nerf_rgb_branch_depth=1, nerf_rgb_branch_width=128, nerf_skips=tuple(spec.nerf_skips), alpha_channels=1, rgb_channels=3,
      use_stratified_sampling=spec.use_stratified_sampling, num_nerf_point_freqs=spec.num_nerf_point_freqs,
      num_nerf_viewdir_freqs=spec.num_nerf_viewdir_freqs, appearance_ids=tuple(range(spec.num_appearance_embeddings)),
      camera_ids=tuple(range(spec.num_camera_embeddings)), warp_ids=tuple(range(spec.num_warp_embeddings)),
      num_appearance_features=spec.num_appearance_features, num_camera_features=spec.num_camera_features,
      num_warp_features=spec.num_warp_features, num_warp_freqs=spec.num_warp_freqs, sigma_activation=nn.softplus,
      use_camera_metadata=spec.use_camera_metadata, use_warp=spec.use_warp, warp_field_type=extra.pop('warp_field_type', 'se3'),
      use_appearance_metadata=spec.use_appearance_metadata, use_alpha_condition=spec.use_alpha_condition, **extra)


def nerf_model():
  for name, (kw, alpha) in NERF_CASES.items():
    spec = O.ModelSpec(**kw)
    seed = sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    batch = O.synthetic_batch(3, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    t_rand = rng.uniform(0, 1, (3, spec.num_coarse_samples)); u = rng.uniform(0, 1, (3, spec.num_fine_samples))
    model = build_ref_model(spec)
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(),
            'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    ret = model.apply({'params': tree_np(params)}, rays, {'alpha': alpha, 'time_alpha': 0.0}, return_points=spec.use_warp,
                      return_weights=True, return_warp_jacobian=spec.use_warp,
                      rngs={'coarse': jrandom.Key(uniform=t_rand), 'fine': jrandom.Key(uniform=u)})
    out = dict(t_rand=t_rand, u=u, alpha=alpha, seed=seed)
    for lv, d in ret.items():
      for k, v in d.items():
        out[f'{lv}/{k}'] = v
    if spec.use_warp:   # the elastic term of training.py on the coarse Jacobians (finite-difference Jacobians: ~1e-7)
      el, res = zip(*[ref_training.compute_elastic_loss(j) for j in ret['coarse']['warp_jacobian'].reshape(-1, 3, 3)])
      out['coarse/elastic_loss'] = np.array(el).reshape(3, -1); out['coarse/elastic_residual'] = np.array(res).reshape(3, -1)
    save('nerf_' + name, **out)


def background_loss():
  """training.compute_background_loss (training.py:117-135) run by the reference on the oracle's warp parameters, with the
  ids / noise it draws supplied through the key."""
  import types
  spec = O.ModelSpec(use_warp=True, num_warp_freqs=6, num_warp_features=8, num_warp_embeddings=4)
  params = O.init_params(spec, seed=21, trained_like=True)
  model = build_ref_model(spec)
  rng = np.random.default_rng(22)
  n = 9
  pts = rng.uniform(-0.4, 0.4, (n, 3)); ids = rng.integers(0, 4, (n, 1)); noise = rng.normal(size=(n, 3))
  state = types.SimpleNamespace(warp_extra={'alpha': 4.5, 'time_alpha': 0.0})
  key = jrandom.Key(normal=noise, choice=ids)
  loss = ref_training.compute_background_loss(model, state, tree_np(params), key, pts, 0.001, alpha=-2, scale=0.001)
  save('background_loss', points=pts, ids=ids, noise=noise, alpha=4.5, noise_std=0.001, loss=np.asarray(loss))


def render_image():
  """evaluation.render_image (evaluation.py:28-101): the chunk / pad / shard / un-pad bookkeeping of the reference, driven
  with a per-ray stand-in for the pmapped model on a ragged 5 x 7 frame over 3 devices."""
  import types
  from nerfies import evaluation as ref_evaluation
  # 36 rays in chunks of 9 over 3 devices.  (Chunks that are not a multiple of the device count cannot be driven through
  # the reference: it slices the padded chunk back to its unpadded length -- `per_host_rays = num_chunk_rays //
  # process_count`, evaluation.py:76-80 -- and the shard reshape then fails.  nerfies_amd pads and un-pads correctly;
  # tests/test_distributed_gloo.py covers the ragged case against the single-process render.)
  h, w, devices, chunk = 4, 9, 3, 9
  rng = np.random.default_rng(31)
  rays = {'origins': rng.normal(size=(h, w, 3)), 'directions': rng.normal(size=(h, w, 3)),
          'metadata': {'warp': rng.integers(0, 4, (h, w, 1))}}

  def pmapped(key_0, key_1, params, r, warp_extra):     # r leaves: (devices, n, c); output as after lax.all_gather
    rgb = np.tanh(r['origins'] + 0.5 * r['directions']) + 0.1 * r['metadata']['warp']
    out = {'rgb': rgb, 'depth': (r['origins'] * r['directions']).sum(-1), 'acc': np.abs(r['origins'][..., 0])}
    return {'fine': {k: np.broadcast_to(v[None], (devices,) + v.shape) for k, v in out.items()}}
  state = types.SimpleNamespace(optimizer=types.SimpleNamespace(target={'model': None}), warp_extra={})
  ret = ref_evaluation.render_image(state, rays, pmapped, devices, jrandom.Key(), chunk=chunk)
  save('render_image', origins=rays['origins'], directions=rays['directions'], warp=rays['metadata']['warp'],
       devices=devices, chunk=chunk, **{'out/' + k: v for k, v in ret.items()})


def dataset_items():
  """nerfies.datasets.nerfies.NerfiesDataSource (the real class) reading a capture written by
  nerfies_amd.datasets.write_synthetic_scene: ids, metadata vocabularies, per-item camera / rgb / metadata, points, and
  datasets.core.camera_to_rays of one loaded camera.  (cv2.imdecode is stood in for by PIL; the tf.data iterators are
  outside the shim.)"""
  import tempfile
  from nerfies.datasets import core as ref_core
  from nerfies.datasets import nerfies as ref_ds
  from nerfies_amd import datasets as mine
  d = tempfile.mkdtemp()
  ids = mine.write_synthetic_scene(d, num_frames=5, size=(16, 12), image_scale=2, seed=3)
  src = ref_ds.NerfiesDataSource(d, image_scale=2, use_appearance_id=True, use_camera_id=True, use_warp_id=True, random_seed=5)
  out = dict(train_ids=np.array(src.train_ids), val_ids=np.array(src.val_ids), appearance_ids=np.array(src.appearance_ids),
             camera_ids=np.array(src.camera_ids), warp_ids=np.array(src.warp_ids), near=src.near, far=src.far,
             points=src.load_points())
  for i in ids:
    item = src.get_item(i)
    for k, v in item['camera_params'].items():
      out[f'{i}/camera/{k}'] = np.asarray(v)
    out[f'{i}/rgb'] = item['rgb']
    out[f'{i}/metadata'] = np.array([item['metadata'][k] for k in ('appearance', 'camera', 'warp')])
  rays = ref_core.camera_to_rays(src.load_camera(ids[2]))
  out.update({'rays/' + k: v for k, v in rays.items()})
  save('dataset_items', **out)


def losses_and_schedules():
  sq = np.array([0.0, 1e-8, 1e-4, 0.01, 0.5, 3.0])
  out = dict(sq=sq, gl_m2_c03=ref_utils.general_loss_with_squared_residual(sq, alpha=-2.0, scale=0.03),
             gl_m2_c001=ref_utils.general_loss_with_squared_residual(sq, alpha=-2.0, scale=0.001),
             gl_1_c1=ref_utils.general_loss_with_squared_residual(sq, alpha=1.0, scale=1.0),
             psnr=ref_utils.compute_psnr(np.array([0.5, 0.01, 1e-4])))
  rng = np.random.default_rng(9)
  J = np.eye(3) + 0.2 * rng.normal(size=(5, 3, 3))
  el = [ref_training.compute_elastic_loss(j) for j in J]
  out.update(el_J=J, el_loss=np.array([e[0] for e in el]), el_residual=np.array([e[1] for e in el]))
  steps = np.array([0, 1, 10, 499, 500, 501, 2500, 10000, 80000, 250000])
  defs = {
      'constant': ('constant', 0.3),
      'linear': ('linear', 0.0, 8.0, 80000),
      'exponential': ('exponential', 1e-3, 1e-4, 250000),
      'cosine_easing': ('cosine_easing', 0.01, 1e-8, 5000),
      'piecewise': ('piecewise', [(500, ('constant', 0.01)), (2000, ('cosine_easing', 0.01, 1e-5, 2000)), (1, ('constant', 1e-5))]),
      'delayed': ('delayed', ('exponential', 1e-3, 1e-4, 250000), 2500, 0.01),
      'step': ('step', 1e-3, 1000, 0.5, 3),
      'dict_linear': {'type': 'linear', 'initial_value': 1.0, 'final_value': 0.25, 'num_steps': 1000},
  }
  for k, dfn in defs.items():
    # from_config's Mapping test uses collections.Mapping (removed in Python 3.10): dispatch here instead
    sch = ref_sched.from_dict(dfn) if isinstance(dfn, dict) else ref_sched.from_tuple(dfn)
    out['sched_' + k] = np.array([float(sch(int(s))) for s in steps])
  out['sched_steps'] = steps
  save('losses_schedules', **out)


def cameras():
  rng = np.random.default_rng(11)
  out = {}
  for tag, dist in (('pinhole', None), ('distorted', ([0.05, -0.02, 0.004], [0.001, -0.002]))):
    q = rng.normal(size=(3, 3)); R, _ = np.linalg.qr(q)
    cam = ref_camera.Camera(orientation=R, position=rng.normal(size=3), focal_length=412.5, principal_point=[160.2, 119.7],
                            image_size=[320, 240], skew=0.3, pixel_aspect_ratio=1.02,
                            radial_distortion=None if dist is None else dist[0],
                            tangential_distortion=None if dist is None else dist[1], dtype=np.float64)
    px = rng.uniform(0, [320, 240], size=(40, 2))
    rays = cam.pixels_to_rays(px)
    pts = cam.position + rays * rng.uniform(0.5, 3.0, (40, 1))
    out.update({f'{tag}/orientation': R, f'{tag}/position': cam.position, f'{tag}/pixels': px, f'{tag}/rays': rays,
                f'{tag}/points': pts, f'{tag}/project': cam.project(pts),
                f'{tag}/radial': cam.radial_distortion, f'{tag}/tangential': cam.tangential_distortion})
    small = ref_camera.Camera(orientation=R, position=cam.position, focal_length=20.0, principal_point=[3.5, 2.5], image_size=[7, 5],
                              radial_distortion=None if dist is None else dist[0],
                              tangential_distortion=None if dist is None else dist[1], dtype=np.float64)
    out[f'{tag}/centers_7x5'] = small.get_pixel_centers()
    out[f'{tag}/centers_rays_7x5'] = small.pixels_to_rays(small.get_pixel_centers())
  out['intrinsics'] = np.array([412.5, 160.2, 119.7, 0.3, 1.02])
  save('camera', **out)


# ---------------------------------------------------------------------------------------------
# round 2: the rest of NerfModel.apply's contract and of train_step's loss assembly
# ---------------------------------------------------------------------------------------------
NERF_CASES_R2 = {
    # use_alpha_condition: the appearance code conditions the alpha head AND (models.py:206) the rgb branch
    'alpha_cond': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True,
                        use_appearance_metadata=True, use_alpha_condition=True, use_camera_metadata=True), 0.0),
    # (noise_std > 0 cannot be driven through the reference's NerfModel: models.py:274 hands noise_regularize the
    #  {'rgb', 'alpha'} DICT the MLP returns, and model_utils.py:280 slices it like the array it was in jaxnerf ->
    #  TypeError under JAX as well.  noise_regularize itself is pinned on an array in elastic_types_and_noise().)
    # every code path at once, evaluated with metadata_encoded=True on the gathered codes
    'encoded': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_warp=True,
                     num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True, use_appearance_metadata=True,
                     use_alpha_condition=True), 3.25),
    # warp_metadata_encoder_type = 'time': modules.TimeEncoder on metadata['time'], half-open annealing window
    'time': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                  num_warp_freqs=5, num_warp_features=8, warp_metadata_encoder_type='time'), 3.25),
}
TIME_ALPHA = 0.6


def nerf_model_r2():
  for name, (kw, alpha) in NERF_CASES_R2.items():
    spec = O.ModelSpec(**kw)
    seed = sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    pn = tree_np(params)
    batch = O.synthetic_batch(3, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    Nc, Nf = spec.num_coarse_samples, spec.num_fine_samples
    t_rand = rng.uniform(0, 1, (3, Nc)); u = rng.uniform(0, 1, (3, Nf))
    nz_c = rng.normal(size=(3, Nc)); nz_f = rng.normal(size=(3, Nc + Nf))
    model = build_ref_model(spec, warp_metadata_encoder_type=spec.warp_metadata_encoder_type)
    md = {k: v.numpy() for k, v in batch['metadata'].items()}
    encoded = name == 'encoded'
    if encoded:   # the codes the encoders would produce (glo.py:50-53), handed over pre-encoded
      md = {'warp': pn['warp_field']['metadata_encoder']['embed']['embedding'][md['warp'][:, 0]],
            'appearance': pn['appearance_encoder']['embed']['embedding'][md['appearance'][:, 0]],
            'camera': pn['camera_encoder']['embed']['embedding'][md['camera'][:, 0]]}
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(), 'metadata': md}
    ret = model.apply({'params': pn}, rays, {'alpha': alpha, 'time_alpha': TIME_ALPHA}, metadata_encoded=encoded,
                      return_points=spec.use_warp, return_weights=True, return_warp_jacobian=spec.use_warp,
                      rngs={'coarse': jrandom.Key(uniform=t_rand, normal=nz_c[..., None]),
                            'fine': jrandom.Key(uniform=u, normal=nz_f[..., None])})
    out = dict(t_rand=t_rand, u=u, noise_coarse=nz_c, noise_fine=nz_f, alpha=alpha, time_alpha=TIME_ALPHA, seed=seed)
    if encoded:
      out.update({'codes/' + k: v for k, v in md.items()})
    for lv, d in ret.items():
      for k, v in d.items():
        out[f'{lv}/{k}'] = v
    save('nerf_' + name, **out)


TRAIN_CASES = {
    # name: (elastic_loss_type, elastic_reduce_method, use_warp_reg_loss)
    'log_svals_weight': ('log_svals', 'weight', True),
    'svals_median': ('svals', 'median', False),
    'jtj_weight': ('jtj', 'weight', False),
    'div_weight': ('div', 'weight', True),
    'det_median': ('det', 'median', False),
    'log_det_weight': ('log_det', 'weight', False),
}


def train_step_stats():
  """training.train_step itself (training.py:138-271), forward half: the reference's own loss assembly -- rgb, every
  elastic_loss_type under both reduce methods, warp_reg, background, the Jacobian metrics, totals, psnr -- on the oracle's
  parameters.  jax.value_and_grad is stood in for by a plain evaluation (the gradient half is torch.autograd on the
  oracle), random.split hands train_step the four prepared keys, the optimizer is a pass-through."""
  import types
  import jax
  spec = O.ModelSpec(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                     num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True)
  params = O.init_params(spec, seed=41, trained_like=True)
  batch = O.synthetic_batch(3, seed=42)
  rng = np.random.default_rng(43)
  t_rand = rng.uniform(0, 1, (3, 8)); u = rng.uniform(0, 1, (3, 6))
  nbg = 7
  bg_pts = rng.uniform(-0.4, 0.4, (nbg, 3)); bg_ids = rng.integers(0, 4, (nbg, 1)); bg_noise = rng.normal(size=(nbg, 3))
  model = build_ref_model(spec, use_warp_jacobian=True)
  rb = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(), 'rgb': batch['rgb'].numpy(),
        'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}, 'background_points': bg_pts}
  sp = ref_training.ScalarParams(learning_rate=1e-3, elastic_loss_weight=0.01, warp_reg_loss_weight=0.5, warp_reg_loss_alpha=-2.0,
                                 warp_reg_loss_scale=0.001, background_loss_weight=1.5, background_noise_std=0.001)

  class Opt:
    target = {'model': tree_np(params)}

    def apply_gradient(self, grad, learning_rate):
      return self
  state = types.SimpleNamespace(optimizer=Opt(), warp_extra={'alpha': 3.25, 'time_alpha': 0.0}, replace=lambda **kw: None)
  bundle = jrandom.Key()
  bundle.parts = [jrandom.Key(), jrandom.Key(uniform=u), jrandom.Key(uniform=t_rand), jrandom.Key(normal=bg_noise, choice=bg_ids)]
  split0, vag0 = jrandom.split, jax.value_and_grad
  jrandom.split = lambda key, num=2: key.parts if hasattr(key, 'parts') and num == 4 else split0(key, num)
  jax.value_and_grad = lambda fn, has_aux=False: (lambda p: (fn(p), None))
  out = dict(t_rand=t_rand, u=u, alpha=3.25, bg_points=bg_pts, bg_ids=bg_ids, bg_noise=bg_noise, elastic_loss_weight=0.01,
             warp_reg_loss_weight=0.5, background_loss_weight=1.5)
  try:
    for name, (ltype, method, wreg) in TRAIN_CASES.items():
      _, stats, _ = ref_training.train_step(model, bundle, state, rb, sp, use_elastic_loss=True, elastic_reduce_method=method,
                                            elastic_loss_type=ltype, use_background_loss=True, use_warp_reg_loss=wreg)
      for lv in ('coarse', 'fine'):
        for k, v in stats[lv].items():
          out[f'{name}/{lv}/{k}'] = np.asarray(v)
      out[f'{name}/background_loss'] = np.asarray(stats['background_loss'])
  finally:
    jrandom.split, jax.value_and_grad = split0, vag0
  save('train_step_stats', **out)


def elastic_types_and_noise():
  rng = np.random.default_rng(51)
  J = np.eye(3) + 0.25 * rng.normal(size=(6, 3, 3))
  out = dict(J=J, div=np.array([ref_utils.jacobian_to_div(j) for j in J]), curl=np.stack([ref_utils.jacobian_to_curl(j) for j in J]))
  for t in ('log_svals', 'svals', 'jtj', 'div', 'det', 'log_det'):
    el = [ref_training.compute_elastic_loss(j, loss_type=t) for j in J]
    out[f'{t}/loss'] = np.array([e[0] for e in el]); out[f'{t}/residual'] = np.array([e[1] for e in el])
  raw = rng.normal(size=(2, 5, 4)); nz = rng.normal(size=(2, 5, 1))
  out.update(raw=raw, normals=nz, noised_strat=ref_mu.noise_regularize(jrandom.Key(normal=nz), raw, 0.4, True),
             noised_det=ref_mu.noise_regularize(jrandom.Key(normal=nz), raw, 0.4, False),
             noised_none=ref_mu.noise_regularize(jrandom.Key(normal=nz), raw, None, True))
  save('elastic_types_noise', **out)


# ---------------------------------------------------------------------------------------------
# round 3: every branch of utils.general_loss_with_squared_residual (utils.py:304-329)
# ---------------------------------------------------------------------------------------------
# ---------------------------------------------------------------------------------------------
# round 4: NerfMLP without ANY condition (use_viewdirs = False, no camera / appearance code): modules.py:149-164 then builds no
# bottleneck layer and feeds the trunk output to the rgb and alpha branches directly
# ---------------------------------------------------------------------------------------------
NERF_CASES_R4 = {
    'nocond': (dict(num_coarse_samples=9, num_fine_samples=7, num_nerf_point_freqs=5, use_stratified_sampling=True, use_viewdirs=False), 0.0),
    'nocond_warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_viewdirs=False,
                         use_warp=True, num_warp_freqs=5, num_warp_features=8), 2.75),
    # trunks shallower than the kernels' 8 layers (modules.MLP, modules.py:41-62): skip at layer 4 still inside / never reached
    'depth6': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_trunk_depth=6,
                    use_camera_metadata=True), 0.0),
    'depth3': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_trunk_depth=3), 0.0),
}


def nerf_model_r4():
  for name, (kw, alpha) in NERF_CASES_R4.items():
    spec = O.ModelSpec(**kw)
    seed = sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    assert ('bottleneck' in params['nerf_mlps_coarse']) == (not name.startswith('nocond'))
    assert len(params['nerf_mlps_coarse']['MLP_0']) == spec.nerf_trunk_depth
    batch = O.synthetic_batch(3, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    t_rand = rng.uniform(0, 1, (3, spec.num_coarse_samples)); u = rng.uniform(0, 1, (3, spec.num_fine_samples))
    model = build_ref_model(spec)
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(),
            'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    ret = model.apply({'params': tree_np(params)}, rays, {'alpha': alpha, 'time_alpha': 0.0}, return_points=spec.use_warp,
                      return_weights=True, return_warp_jacobian=spec.use_warp,
                      rngs={'coarse': jrandom.Key(uniform=t_rand), 'fine': jrandom.Key(uniform=u)})
    out = dict(t_rand=t_rand, u=u, alpha=alpha, seed=seed)
    for lv, d in ret.items():
      for k, v in d.items():
        out[f'{lv}/{k}'] = v
    save('nerf_' + name, **out)

# ---------------------------------------------------------------------------------------------
# round 6: the Gin surface beside the presets -- nerf_skips at another layer (configs.py:63, modules.py:47-48) and the warp
# field's trunk shape from ModelConfig.warp_kwargs (configs.py:105 -> models.py:165-184 -> warping.py:225-226 / 90-91)
# ---------------------------------------------------------------------------------------------
NERF_CASES_R6 = {
    # skip at 5 in an 8-layer trunk: the skip GEMM moves to layer 5 (float32 chains)
    'skip5': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_skips=(5,),
                   use_camera_metadata=True), 0.0),
    # skip at 2 in a 6-layer trunk: laid out around the kernels' own layer 4 with identity layers in between (every mode)
    'skip2_depth6': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, nerf_skips=(2,),
                          nerf_trunk_depth=6), 0.0),
    # skip at 1 with the warp field: d posenc through the moved skip layer (the Jacobian and the warped points are outputs)
    'skip1_warp': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, nerf_skips=(1,),
                        use_warp=True, num_warp_freqs=5, num_warp_features=8), 2.5),
    # SE3Field(trunk_depth=5, trunk_width=96): the skip at 4 is the last layer
    'warp_trunk5x96': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True, use_warp=True,
                            num_warp_freqs=5, num_warp_features=8, warp_trunk_depth=5, warp_trunk_width=96, use_camera_metadata=True), 3.25),
    # SE3Field(trunk_depth=3, trunk_width=64): the skip is never reached
    'warp_trunk3x64': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=False, use_warp=True,
                            num_warp_freqs=4, num_warp_features=8, warp_trunk_depth=3, warp_trunk_width=64), 1.5),
    # TranslationField(depth=4, hidden_channels=80)
    'translation_trunk4x80': (dict(num_coarse_samples=8, num_fine_samples=6, num_nerf_point_freqs=4, use_stratified_sampling=True,
                                   use_warp=True, warp_field_type='translation', num_warp_freqs=5, num_warp_features=8,
                                   warp_trunk_depth=4, warp_trunk_width=80), 2.25),
}


def warp_kwargs_of(spec):
  d, w = spec.warp_trunk_depth, spec.warp_trunk_width
  if not spec.use_warp or (d, w) == (6, 128):
    return {}
  return {'depth': d, 'hidden_channels': w} if spec.warp_field_type == 'translation' else {'trunk_depth': d, 'trunk_width': w}


def nerf_model_r6():
  for name, (kw, alpha) in NERF_CASES_R6.items():
    spec = O.ModelSpec(**kw)
    seed = sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    batch = O.synthetic_batch(3, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    t_rand = rng.uniform(0, 1, (3, spec.num_coarse_samples)); u = rng.uniform(0, 1, (3, spec.num_fine_samples))
    # warp_kwargs reach the field exactly as NerfModel passes them on (models.py:165-184)
    model = build_ref_model(spec, warp_field_type=spec.warp_field_type, warp_kwargs=warp_kwargs_of(spec))
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(),
            'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    ret = model.apply({'params': tree_np(params)}, rays, {'alpha': alpha, 'time_alpha': 0.0}, return_points=spec.use_warp,
                      return_weights=True, return_warp_jacobian=spec.use_warp,
                      rngs={'coarse': jrandom.Key(uniform=t_rand), 'fine': jrandom.Key(uniform=u)})
    out = dict(t_rand=t_rand, u=u, alpha=alpha, seed=seed)
    for lv, d in ret.items():
      for k, v in d.items():
        out[f'{lv}/{k}'] = v
    save('nerf_' + name, **out)



def general_loss_branches():
  sq = np.array([0.0, 1e-8, 1e-4, 0.01, 0.5, 3.0])
  out = dict(sq=sq)
  for name, alpha in (('neginf', -np.inf), ('m2', -2.0), ('zero', 0.0), ('one', 1.0), ('two', 2.0), ('posinf', np.inf)):
    for cname, scale in (('c03', 0.03), ('c1', 1.0)):
      with np.errstate(over='ignore', invalid='ignore'):
        out[f'{name}_{cname}'] = ref_utils.general_loss_with_squared_residual(sq, alpha=np.float64(alpha), scale=scale)
  save('general_loss_branches', **out)


# ---- round 5: NerfModel.apply at the BASELINE shapes (every earlier ref_nerf_* case is 3 rays x <= 10+7 samples) ----
BASELINE_CASES = {
    # configs[1] gpu_quarterhd as measured: 64 + 128 samples, F_p = 8, stratified, warp off (bench.py's headline workload)
    'cfgA': (dict(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True), 0.0, 64),
    # configs[2] gpu_vrig_paper: 128 + 128, F_p = 8, SE3 warp F_w = 6, G = 8, camera code
    'cfgC': (dict(num_coarse_samples=128, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True, use_warp=True,
                  num_warp_freqs=6, num_warp_features=8, use_camera_metadata=True), 6.0, 16),
    # configs[3] gpu_fullhd: 256 + 256, F_p = 10, SE3 warp F_w = 8, appearance ids
    'cfgD': (dict(num_coarse_samples=256, num_fine_samples=256, num_nerf_point_freqs=10, use_stratified_sampling=True, use_warp=True,
                  num_warp_freqs=8, num_warp_features=8, use_appearance_metadata=True), 8.0, 8),
}


def nerf_model_baseline_shapes():
  """models.NerfModel.apply (models.py:289-375) by the unmodified reference at the sample counts / posenc widths BASELINE.json
  names.  No Jacobian output (the shim's jacfwd is 6 warp evaluations per sample); float32 storage keeps the fixtures small --
  the consumers compare float32 kernels with 1e-4 tolerances."""
  for name, (kw, alpha, B) in BASELINE_CASES.items():
    spec = O.ModelSpec(**kw)
    seed = 500 + sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    batch = O.synthetic_batch(B, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    t_rand = rng.uniform(0, 1, (B, spec.num_coarse_samples)).astype(np.float32).astype(np.float64)
    u = rng.uniform(0, 1, (B, spec.num_fine_samples)).astype(np.float32).astype(np.float64)
    model = build_ref_model(spec)
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(),
            'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    ret = model.apply({'params': tree_np(params)}, rays, {'alpha': alpha, 'time_alpha': 0.0}, return_points=spec.use_warp,
                      return_weights=True, rngs={'coarse': jrandom.Key(uniform=t_rand), 'fine': jrandom.Key(uniform=u)})
    out = dict(t_rand=t_rand.astype(np.float32), u=u.astype(np.float32), alpha=alpha, seed=seed, num_rays=B)
    for lv, d in ret.items():
      for k, v in d.items():
        if k in ('rgb', 'depth', 'med_depth', 'acc', 'weights', 'warped_points'):
          out[f'{lv}/{k}'] = np.asarray(v, dtype=np.float32)
    save('nerf_' + name, **out)


# ---- round 6: the same three configurations at the FULL batch sizes of BASELINE.json (1024 / 768 / 512 rays): the rendered outputs
#      only (rgb, depth, med_depth, acc: 48 KB for 1024 rays); the uniforms are re-drawn from the seed by the consumer ----
FULL_BATCH = {'cfgA': 1024, 'cfgC': 768, 'cfgD': 512}


def full_batch_uniforms(seed, B, spec):
  rng = np.random.default_rng(seed + 2)
  t_rand = rng.uniform(0, 1, (B, spec.num_coarse_samples)).astype(np.float32).astype(np.float64)
  u = rng.uniform(0, 1, (B, spec.num_fine_samples)).astype(np.float32).astype(np.float64)
  return t_rand, u


def nerf_model_full_batches():
  import time
  for name, (kw, alpha, _) in BASELINE_CASES.items():
    B = FULL_BATCH[name]
    spec = O.ModelSpec(**kw)
    seed = 900 + sum(ord(c) for c in name)
    params = O.init_params(spec, seed=seed, trained_like=True)
    batch = O.synthetic_batch(B, seed=seed + 1)
    t_rand, u = full_batch_uniforms(seed, B, spec)
    model = build_ref_model(spec)
    rays = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(),
            'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    t0 = time.time()
    ret = model.apply({'params': tree_np(params)}, rays, {'alpha': alpha, 'time_alpha': 0.0}, return_points=False, return_weights=False,
                      rngs={'coarse': jrandom.Key(uniform=t_rand), 'fine': jrandom.Key(uniform=u)})
    print(f'{name}: {B} rays through the reference in {time.time() - t0:.1f} s')
    out = dict(alpha=alpha, seed=seed, num_rays=B)
    for lv, d in ret.items():
      for k, v in d.items():
        if k in ('rgb', 'depth', 'med_depth', 'acc'):
          out[f'{lv}/{k}'] = np.asarray(v, dtype=np.float32)
    save('nerf_' + name + '_full', **out)


# ---- round 5: a REFERENCE-side gradient.  jax.value_and_grad cannot run under the NumPy shim; what can is the
#      reference's own _loss_fn (training.py:229-262, the closure train_step differentiates), evaluated in float64 at
#      params +- eps * v: a central-difference directional derivative.  lax.stop_gradient (model_utils.py:187 on the fine
#      z samples, training.py:181 / :200 on the weights) must then keep its meaning -- value passes, derivative does not --
#      so the base evaluation RECORDS what flows through every stop_gradient call and the perturbed evaluations REPLAY those
#      values in call order. ----
LOSS_DIR_CASES = {
    # the headline workload's model, warp off (only the two MSE terms)
    'nowarp': dict(spec=dict(num_coarse_samples=64, num_fine_samples=128, num_nerf_point_freqs=8, use_stratified_sampling=True),
                   B=12, alpha=0.0, bg=0, eps=1e-6),
    # SE3 warp + camera code + background regulariser (training.py:117-135); no elastic term here: its Jacobian is a central
    # difference under the shim and a difference of differences is noise
    'warp_bg': dict(spec=dict(num_coarse_samples=24, num_fine_samples=24, num_nerf_point_freqs=6, use_stratified_sampling=True, use_warp=True,
                              num_warp_freqs=5, num_warp_features=8, use_camera_metadata=True), B=8, alpha=3.25, bg=9, eps=1e-7),
}
# eps: the warped points enter a posenc of 2^(F_p-1) x their value, so the loss curves ~30x faster along a direction that moves the
# warp field: the central difference's truncation error is 1e-3 of the slope at eps = 1e-6 and 1e-6 of it at 1e-7 (measured at
# 1e-5 / 1e-6 / 1e-7: the values converge quadratically), while float64 cancellation at 1e-7 is still ~1e-9
LOSS_DIR_NDIR = 8


def loss_directions(params_np, seed, ndir=LOSS_DIR_NDIR):
  """The seeded parameter directions (shared with the tests): per direction one standard-normal array per leaf, leaves in
  sorted path order, each scaled by the leaf's rms (so eps * v is a RELATIVE perturbation of every leaf)."""
  leaves = []

  def walk(t, path):
    for k in sorted(t):
      (walk(t[k], path + (k,)) if isinstance(t[k], dict) else leaves.append((path + (k,), np.asarray(t[k]))))
  walk(params_np, ())
  rng = np.random.default_rng(seed)
  dirs = []
  for _ in range(ndir):
    d = {}
    for path, a in leaves:
      d[path] = rng.standard_normal(a.shape) * (np.sqrt(np.mean(a * a)) + 1e-3)
    dirs.append(d)
  return dirs


def _tree_axpy(t, d, c, path=()):
  return {k: (_tree_axpy(v, d, c, path + (k,)) if isinstance(v, dict) else np.asarray(v) + c * d[path + (k,)]) for k, v in t.items()}


def loss_directional():
  import types
  import jax
  from jax import lax as jlax
  for name, case in LOSS_DIR_CASES.items():
    spec = O.ModelSpec(**case['spec'])
    B, nbg, alpha = case['B'], case['bg'], case['alpha']
    seed = 700 + sum(ord(c) for c in name)
    params = tree_np(O.init_params(spec, seed=seed, trained_like=True))
    batch = O.synthetic_batch(B, seed=seed + 1)
    rng = np.random.default_rng(seed + 2)
    t_rand = rng.uniform(0, 1, (B, spec.num_coarse_samples)).astype(np.float32).astype(np.float64)
    u = rng.uniform(0, 1, (B, spec.num_fine_samples)).astype(np.float32).astype(np.float64)
    rb = {'origins': batch['origins'].numpy(), 'directions': batch['directions'].numpy(), 'rgb': batch['rgb'].numpy(),
          'metadata': {k: v.numpy() for k, v in batch['metadata'].items()}}
    eps = case['eps']
    out = dict(t_rand=t_rand.astype(np.float32), u=u.astype(np.float32), alpha=alpha, seed=seed, num_rays=B, eps=eps)
    keys = [jrandom.Key(), jrandom.Key(uniform=u), jrandom.Key(uniform=t_rand), jrandom.Key()]
    if nbg:
      bg_pts = rng.uniform(-0.4, 0.4, (nbg, 3)).astype(np.float32).astype(np.float64)
      bg_ids = rng.integers(0, 4, (nbg, 1)); bg_noise = rng.normal(size=(nbg, 3)).astype(np.float32).astype(np.float64)
      rb['background_points'] = bg_pts
      keys[3] = jrandom.Key(normal=bg_noise, choice=bg_ids)
      out.update(bg_points=bg_pts.astype(np.float32), bg_ids=bg_ids, bg_noise=bg_noise.astype(np.float32), background_loss_weight=1.5)
    sp = ref_training.ScalarParams(learning_rate=1e-3, background_loss_weight=1.5, background_noise_std=0.001)
    model = build_ref_model(spec)

    class Opt:
      target = {'model': params}

      def apply_gradient(self, grad, learning_rate):
        return self
    state = types.SimpleNamespace(optimizer=Opt(), warp_extra={'alpha': alpha, 'time_alpha': 0.0}, replace=lambda **kw: None)
    bundle = jrandom.Key()
    bundle.parts = keys
    captured = []
    split0, vag0, sg0 = jrandom.split, jax.value_and_grad, jlax.stop_gradient
    jrandom.split = lambda key, num=2: key.parts if hasattr(key, 'parts') and num == 4 else split0(key, num)

    def vag(fn, has_aux=False):
      captured.append(fn)          # the reference's own _loss_fn closure (training.py:229-262)
      return lambda p: (fn(p), None)
    jax.value_and_grad = vag
    tape = {'mode': 'off', 'vals': [], 'at': 0}

    def stop_gradient(x):
      if tape['mode'] == 'record':
        tape['vals'].append(np.array(x, copy=True))
        return x
      if tape['mode'] == 'replay':
        v = tape['vals'][tape['at']]
        tape['at'] += 1
        assert np.shape(v) == np.shape(x)
        return v
      return x
    jlax.stop_gradient = stop_gradient
    try:
      ref_training.train_step(model, bundle, state, rb, sp, use_elastic_loss=False, use_background_loss=bool(nbg))
      loss_fn = captured[0]
      tape['mode'] = 'record'
      base = loss_fn(state.optimizer.target)
      base = float(base[0] if isinstance(base, tuple) else base)
      nstop = len(tape['vals'])
      tape['mode'] = 'replay'
      dd = []
      for d in loss_directions(params, seed + 3):
        vals = []
        for sgn in (+1.0, -1.0):
          tape['at'] = 0
          r = loss_fn({'model': _tree_axpy(params, d, sgn * eps)})
          assert tape['at'] == nstop
          vals.append(float(r[0] if isinstance(r, tuple) else r))
        dd.append((vals[0] - vals[1]) / (2 * eps))
    finally:
      jrandom.split, jax.value_and_grad, jlax.stop_gradient = split0, vag0, sg0
    out.update(loss=base, directional=np.array(dd), dir_seed=seed + 3, num_stop_gradient_calls=nstop)
    print(name, 'loss', base, 'stop_gradient calls', nstop, 'directional', np.array(dd))
    save('loss_directional_' + name, **out)


if __name__ == '__main__':
  if len(sys.argv) > 1:   # only the named generators (fixtures of earlier rounds stay byte-identical either way)
    for name in sys.argv[1:]:
      globals()[name]()
    sys.exit(0)
  general_loss_branches()
  rigid_body()
  model_utils()
  encoders_and_mlps()
  se3_field()
  translation_field()
  nerf_model()
  background_loss()
  render_image()
  dataset_items()
  losses_and_schedules()
  cameras()
  nerf_model_r2()
  train_step_stats()
  elastic_types_and_noise()
  nerf_model_r4()
  nerf_model_baseline_shapes()
  loss_directional()
  nerf_model_r6()
  nerf_model_full_batches()
