"""Model module: ``model_builder`` + the HIP-backed model classes (xt/model/__init__.py:43-47)."""
from xingtian_amd.register import Registers
from xingtian_amd.model.model import XTModel  # noqa: F401


def model_builder(model_info):
    """Create the interface func for creating model (xt/model/__init__.py:43-47)."""
    model_name = model_info["model_name"]
    return Registers.model[model_name](model_info)


def _register_defaults():
    # the reference auto-imports xt/model/*/*.py (register.py:95-139); we import explicitly
    from xingtian_amd.model.ppo import ppo_cnn, ppo_mlp  # noqa: F401
    from xingtian_amd.model.impala import impala_cnn_opt  # noqa: F401


_register_defaults()
