"""Shared test plumbing: oracle ModelSpec -> nerfies_amd model on the GPU with identical parameters."""
import types

import torch

from oracle import nerfies_oracle as O

DEV = 'cuda:0'


def config_from_spec(spec):
  """An object with the configs.ModelConfig attribute names construct_nerf reads (configs.py:35-105)."""
  return types.SimpleNamespace(
      num_coarse_samples=spec.num_coarse_samples, num_fine_samples=spec.num_fine_samples,
      use_viewdirs=spec.use_viewdirs, nerf_trunk_depth=spec.nerf_trunk_depth, nerf_trunk_width=spec.nerf_trunk_width,
      nerf_rgb_branch_depth=spec.nerf_rgb_branch_depth, nerf_rgb_branch_width=spec.nerf_rgb_branch_width,
      nerf_skips=tuple(spec.nerf_skips), use_stratified_sampling=spec.use_stratified_sampling,
      num_nerf_point_freqs=spec.num_nerf_point_freqs, num_nerf_viewdir_freqs=spec.num_nerf_viewdir_freqs,
      sigma_activation=spec.sigma_activation, use_white_background=spec.use_white_background,
      use_linear_disparity=spec.use_linear_disparity, use_sample_at_infinity=spec.use_sample_at_infinity,
      use_appearance_metadata=spec.use_appearance_metadata, use_camera_metadata=spec.use_camera_metadata,
      appearance_metadata_dims=spec.num_appearance_features, camera_metadata_dims=spec.num_camera_features,
      use_warp=spec.use_warp, num_warp_freqs=spec.num_warp_freqs, num_warp_features=spec.num_warp_features,
      warp_field_type=spec.warp_field_type, use_alpha_condition=spec.use_alpha_condition, use_rgb_condition=spec.use_rgb_condition)


def gpu_model(spec, oparams, batch_size=0):
  """(model, FlatParams on DEV) holding exactly the oracle parameter tree `oparams`."""
  from nerfies_amd import models, params as P
  model, fp = models.construct_nerf(
      0, config_from_spec(spec), batch_size, list(range(spec.num_appearance_embeddings)),
      list(range(spec.num_camera_embeddings)), list(range(spec.num_warp_embeddings)), spec.near, spec.far)
  P.flat_from_tree(O.tree_map(lambda t: t.float(), oparams), model.layout, DEV, out=fp.flat)
  return model, fp


def gpu_batch(batch):
  out = {k: v.to(DEV).float() for k, v in batch.items() if torch.is_tensor(v)}
  out['metadata'] = {k: v.to(DEV) for k, v in batch.get('metadata', {}).items()}
  return out


def flat_grad_from_tree(grads, layout):
  from nerfies_amd import params as P
  return P.flat_from_tree(O.tree_map(lambda t: t.float(), grads), layout, 'cpu')
