"""ctypes binding of include/nerfies_amd.h.  Fails loudly when the HIP extension is absent:
there is no CPU or PyTorch fallback for the hot path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, '_lib', 'libnerfies_amd.so')

NRF_FLAG_TRAIN = 1
NRF_FLAG_NO_WARP = 2
NRF_FLAG_BF16 = 4
NRF_FLAG_WARP_JACOBIAN = 8
NRF_FLAG_WARP_F32 = 16
NRF_FLAG_BF16X3 = 32
NRF_NUM_STATS = 16
ACT = {'relu': 0, 'softplus': 1}
WARP_FIELD = {'se3': 0, 'translation': 1}
META_ENCODER = {'glo': 0, 'time': 1}
ELASTIC_REDUCE = {'weight': 0, 'median': 1}
ELASTIC_TYPE = {'log_svals': 0, 'svals': 1, 'jtj': 2, 'div': 3, 'det': 4, 'log_det': 5}


class NrfError(RuntimeError):
  pass


class ModelDesc(C.Structure):
  _fields_ = [
      ('num_coarse_samples', C.c_int32), ('num_fine_samples', C.c_int32), ('use_viewdirs', C.c_int32),
      ('near_plane', C.c_float), ('far_plane', C.c_float),
      ('nerf_trunk_depth', C.c_int32), ('nerf_trunk_width', C.c_int32),
      ('nerf_rgb_branch_depth', C.c_int32), ('nerf_rgb_branch_width', C.c_int32), ('nerf_skip_layer', C.c_int32),
      ('use_stratified_sampling', C.c_int32), ('num_nerf_point_freqs', C.c_int32),
      ('num_nerf_viewdir_freqs', C.c_int32), ('sigma_activation', C.c_int32),
      ('use_white_background', C.c_int32), ('use_linear_disparity', C.c_int32), ('use_sample_at_infinity', C.c_int32),
      ('use_appearance_metadata', C.c_int32), ('num_appearance_embeddings', C.c_int32),
      ('num_appearance_features', C.c_int32), ('use_camera_metadata', C.c_int32),
      ('num_camera_embeddings', C.c_int32), ('num_camera_features', C.c_int32),
      ('use_alpha_condition', C.c_int32), ('use_rgb_condition', C.c_int32), ('use_trunk_condition', C.c_int32),
      ('use_warp', C.c_int32), ('num_warp_freqs', C.c_int32), ('num_warp_embeddings', C.c_int32),
      ('num_warp_features', C.c_int32), ('warp_field_type', C.c_int32),
      ('noise_std', C.c_float), ('warp_metadata_encoder_type', C.c_int32), ('num_time_encoder_freqs', C.c_int32),
      ('warp_trunk_depth', C.c_int32), ('warp_trunk_width', C.c_int32),
  ]


class TensorInfo(C.Structure):
  _fields_ = [('name', C.c_char * 96), ('offset', C.c_int64), ('rows', C.c_int32), ('cols', C.c_int32)]


class Rays(C.Structure):
  _fields_ = [('num_rays', C.c_int32), ('origins', C.c_void_p), ('directions', C.c_void_p), ('viewdirs', C.c_void_p),
              ('warp_ids', C.c_void_p), ('appearance_ids', C.c_void_p), ('camera_ids', C.c_void_p),
              ('warp_codes', C.c_void_p), ('appearance_codes', C.c_void_p), ('camera_codes', C.c_void_p),
              ('time', C.c_void_p)]


class DynamicScalars(C.Structure):
  """nrf_dynamic_scalars: the per-step scalars in DEVICE memory (64 bytes) of a graph-replayed train step."""
  _fields_ = [('warp_alpha', C.c_float), ('time_alpha', C.c_float), ('elastic_loss_weight', C.c_float), ('learning_rate', C.c_float),
              ('adam_c1', C.c_float), ('adam_c2', C.c_float), ('grad_scale', C.c_float), ('reserved0', C.c_float),
              ('rng_seed', C.c_uint64), ('rng_offset', C.c_uint64), ('reserved1', C.c_uint64 * 2)]


class StepScalars(C.Structure):
  _fields_ = [('warp_alpha', C.c_float), ('time_alpha', C.c_float), ('dynamic', C.c_void_p)]


class Rand(C.Structure):
  _fields_ = [('t_rand', C.c_void_p), ('u', C.c_void_p), ('seed', C.c_uint64), ('offset', C.c_uint64),
              ('noise_coarse', C.c_void_p), ('noise_fine', C.c_void_p)]


class LevelOut(C.Structure):
  _fields_ = [('rgb', C.c_void_p), ('depth', C.c_void_p), ('med_depth', C.c_void_p), ('acc', C.c_void_p),
              ('weights', C.c_void_p), ('z_vals', C.c_void_p), ('points', C.c_void_p), ('warped_points', C.c_void_p),
              ('warp_jacobian', C.c_void_p)]


class Outputs(C.Structure):
  _fields_ = [('coarse', LevelOut), ('fine', LevelOut)]


class Background(C.Structure):
  _fields_ = [('num_points', C.c_int32), ('points', C.c_void_p), ('warp_ids', C.c_void_p), ('loss_weight', C.c_float),
              ('loss_alpha', C.c_float), ('loss_scale', C.c_float), ('id_choices', C.c_void_p), ('num_choices', C.c_int32),
              ('noise_std', C.c_float)]


class Elastic(C.Structure):
  _fields_ = [('loss_weight', C.c_float), ('reduce_method', C.c_int32), ('eps', C.c_float), ('loss_alpha', C.c_float),
              ('loss_scale', C.c_float), ('loss_type', C.c_int32)]


class WarpReg(C.Structure):
  _fields_ = [('loss_weight', C.c_float), ('loss_alpha', C.c_float), ('loss_scale', C.c_float)]


class CameraDesc(C.Structure):
  _fields_ = [('orientation', C.c_float * 9), ('position', C.c_float * 3), ('focal_length', C.c_float),
              ('principal_point', C.c_float * 2), ('skew', C.c_float), ('pixel_aspect_ratio', C.c_float),
              ('radial_distortion', C.c_float * 3), ('tangential_distortion', C.c_float * 2),
              ('image_size', C.c_int32 * 2)]


class ProfileEntry(C.Structure):
  _fields_ = [('name', C.c_char * 32), ('ms', C.c_double), ('launches', C.c_int32), ('pad_', C.c_int32),
              ('flops_per_launch', C.c_double)]


NRF_OPT_CHAIN_TILE_ROWS = 1
NRF_OPT_BF16_WGRAD_MERGE = 2

EXPORTS = [
    'nrf_version', 'nrf_last_error', 'nrf_create', 'nrf_destroy', 'nrf_param_count', 'nrf_param_layout',
    'nrf_workspace_bytes', 'nrf_forward', 'nrf_backward', 'nrf_train_step_loss_grad', 'nrf_adam_step',
    'nrf_sample_along_rays', 'nrf_volumetric_rendering', 'nrf_sample_pdf', 'nrf_profile_enable', 'nrf_profile_read',
    'nrf_debug_wgrad_segments', 'nrf_debug_ws_offset', 'nrf_train_step_loss_grad_ex', 'nrf_workspace_bytes_ex',
    'nrf_warp_points_workspace_bytes', 'nrf_warp_points',
    'nrf_camera_pixels_to_rays', 'nrf_camera_pixels_to_points', 'nrf_camera_project',
    'nrf_dynamic_scalars_write', 'nrf_adam_step_dynamic', 'nrf_set_option',
]

_lib = None


def load_library(path=None):
  """Loads libnerfies_amd.so (built by nerfies_amd.build / __graft_entry__.build())."""
  global _lib
  if _lib is not None and path is None:
    return _lib
  # torch first: its wheel carries its own HIP runtime; loading ours before it would put two runtimes in one
  # process (device pointers of one are "no ROCm-capable device" errors in the other)
  import torch  # noqa: F401
  path = path or os.environ.get('NRF_LIB_PATH') or LIB_PATH   # NRF_LIB_PATH: experiment builds of the same ABI
  if not os.path.exists(path):
    raise NrfError(
        f'HIP extension not built: {path} is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
        '(needs hipcc). There is no CPU fallback for the hot path.')
  lib = C.CDLL(path)
  vp, i32, i64, u32, u64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
  lib.nrf_version.restype = C.c_int
  lib.nrf_last_error.restype = C.c_char_p
  sigs = {
      'nrf_create': [C.POINTER(ModelDesc), C.POINTER(vp)],
      'nrf_destroy': [vp],
      'nrf_param_count': [vp, C.POINTER(i64)],
      'nrf_param_layout': [vp, C.POINTER(TensorInfo), C.POINTER(i32)],
      'nrf_workspace_bytes': [vp, i32, u32, C.POINTER(C.c_size_t)],
      'nrf_forward': [vp, vp, C.POINTER(Rays), C.POINTER(StepScalars), C.POINTER(Rand), C.POINTER(Outputs), u32, vp,
                      C.c_size_t, vp],
      'nrf_backward': [vp, vp, C.POINTER(Rays), vp, vp, vp, vp, C.c_size_t, vp],
      'nrf_train_step_loss_grad': [vp, vp, C.POINTER(Rays), vp, C.POINTER(StepScalars), C.POINTER(Rand), vp, vp, vp,
                                   C.c_size_t, vp],
      'nrf_adam_step': [vp, vp, vp, vp, i64, f64, f64, f64, f64, i64, f64, vp],
      'nrf_adam_step_dynamic': [vp, vp, vp, vp, i64, f64, f64, f64, vp, vp],
      'nrf_dynamic_scalars_write': [vp, C.POINTER(DynamicScalars), vp],
      'nrf_sample_along_rays': [vp, vp, i32, i32, f32, f32, i32, i32, vp, u64, u64, vp, vp],
      'nrf_volumetric_rendering': [vp, vp, vp, i32, i32, i32, i32, C.POINTER(LevelOut), vp],
      'nrf_sample_pdf': [vp, vp, i32, i32, i32, i32, vp, u64, u64, vp, vp],
      'nrf_profile_enable': [vp, i32],
      'nrf_profile_read': [vp, C.POINTER(ProfileEntry), C.POINTER(i32)],
      'nrf_debug_wgrad_segments': [vp, vp, C.POINTER(C.c_double), C.POINTER(i32)],
      'nrf_debug_ws_offset': [vp, C.c_char_p, i32, C.POINTER(i64)],
      'nrf_set_option': [vp, i32, i64],
      'nrf_train_step_loss_grad_ex': [vp, vp, C.POINTER(Rays), vp, C.POINTER(StepScalars), C.POINTER(Rand),
                                      C.POINTER(Background), C.POINTER(Elastic), C.POINTER(WarpReg), u32, vp, vp, vp,
                                      C.c_size_t, vp],
      'nrf_workspace_bytes_ex': [vp, i32, u32, i32, i32, C.POINTER(C.c_size_t)],
      'nrf_warp_points_workspace_bytes': [vp, i32, C.POINTER(C.c_size_t)],
      'nrf_warp_points': [vp, vp, vp, vp, i32, C.POINTER(StepScalars), vp, vp, C.c_size_t, vp],
      'nrf_camera_pixels_to_rays': [C.POINTER(CameraDesc), vp, i64, vp, vp, vp, vp],
      'nrf_camera_pixels_to_points': [C.POINTER(CameraDesc), vp, vp, i64, vp, vp],
      'nrf_camera_project': [C.POINTER(CameraDesc), vp, i64, vp, vp],
  }
  for name, argtypes in sigs.items():
    fn = getattr(lib, name)
    fn.argtypes = argtypes
    fn.restype = C.c_int
  _lib = lib
  return lib


def check(rc, lib=None):
  if rc != 0:
    lib = lib or load_library()
    raise NrfError(f'nerfies_amd error {rc}: {lib.nrf_last_error().decode()}')
