"""xingtian_amd -- MI355X-native PPO/IMPALA learner update behind XingTian's plugin API.

Host code is Python (as the reference is); all arithmetic runs in hand-written HIP
kernels (``csrc/``) reached through the C ABI in ``include/xt_mi355x.h`` via ctypes.
PyTorch-ROCm tensors are used only as device storage / stream handles.
"""
from xingtian_amd.register import Registers, import_config  # noqa: F401

__version__ = "0.1.0"
