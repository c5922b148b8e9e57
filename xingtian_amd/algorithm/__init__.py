"""Algorithm module: base class + ``alg_builder`` (xt/algorithm/__init__.py:19-28)."""
from xingtian_amd.algorithm.algorithm import Algorithm, AGENT_PREFIX, MODEL_PREFIX  # noqa: F401
from xingtian_amd.register import Registers


def alg_builder(alg_name, model_info, alg_config, **kwargs):
    """The API to build a algorithm instance (xt/algorithm/__init__.py:19-28)."""
    return Registers.algorithm[alg_name](model_info, alg_config, **kwargs)


def _register_defaults():
    from xingtian_amd.algorithm.ppo import ppo  # noqa: F401
    from xingtian_amd.algorithm.impala import impala_opt  # noqa: F401


_register_defaults()
