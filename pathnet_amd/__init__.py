"""pathnet_amd -- MI355X-native implementation of PathNet's path-aggregation hot path.

  pathnet_amd.sampler   MERW random-walk path sampler (GPU walker + the gen_merw CLI)
  pathnet_amd.pathfile  the reference's text path-file format
  pathnet_amd.modules   PathNet / PathNet_homo / PAGG nn.Modules on hand-written HIP kernels
  pathnet_amd.dist      node sharding across the GPUs of one box (RCCL)
  pathnet_amd.optim     CrossEntropyLoss / Adam of the reference's training step as single launches
  pathnet_amd.trainer   train_fixed_indices with everything resident on the GPU

Everything computes through csrc/libpathnet_hip.so (C ABI: include/pathnet_hip.h).
"""
from . import _lib  # noqa: F401
from .modules import PAGG, PathNet, PathNet_homo  # noqa: F401
from .optim import Adam, CrossEntropyLoss, StepState, backward, cross_entropy  # noqa: F401
from .sampler import DRAW_GLIBC_REPLAY, DRAW_PHILOX, MerwSampler, UniformSampler  # noqa: F401

__version__ = "0.1.0"
