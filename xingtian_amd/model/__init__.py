"""HIP-backed model plugins.  ``model_builder(model_info)`` is the framework's factory entry point
(xt/model/__init__.py:43-47): the class is chosen by ``model_info["model_name"]``."""
from xingtian_amd.register import Registers
from xingtian_amd.model.model import XTModel  # noqa: F401


def model_builder(model_info):
    return Registers.model.build(model_info["model_name"], model_info)


# the reference discovers plugins by importing every xt/model/*/*.py (register.py:95-139); here the list is explicit
from xingtian_amd.model.ppo import ppo_cnn, ppo_mlp  # noqa: E402,F401
from xingtian_amd.model.impala import impala_cnn, impala_cnn_opt  # noqa: E402,F401
