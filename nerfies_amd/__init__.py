"""nerfies_amd: MI355X-native (gfx950) hot path of google/nerfies.

The compute path is hand-written HIP behind a C-ABI (include/nerfies_amd.h,
nerfies_amd/csrc); this package is the thin Python host that mirrors the
reference's NerfModel.apply / train_step / render_image interfaces on top of it.
PyTorch is used for device memory, streams and torch.distributed only.

Modules (same names as the reference's): models, training, evaluation, camera, datasets, schedules, configs (+ gin_lite),
checkpoints, utils, visualization; the drivers are train.py / eval.py at the repository root.
"""
from nerfies_amd.lib import NrfError, load_library  # noqa: F401

__version__ = '0.1.1'
