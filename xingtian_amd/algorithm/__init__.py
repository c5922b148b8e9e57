"""Learner-side algorithm plugins.  ``alg_builder(alg_name, model_info, alg_config)`` is the framework's factory
entry point (xt/algorithm/__init__.py:19-28): the class is chosen by ``alg_para.alg_name``."""
from xingtian_amd.algorithm.algorithm import Algorithm, AGENT_PREFIX, MODEL_PREFIX  # noqa: F401
from xingtian_amd.register import Registers


def alg_builder(alg_name, model_info, alg_config, **kwargs):
    return Registers.algorithm.build(alg_name, model_info, alg_config, **kwargs)


from xingtian_amd.algorithm.ppo import ppo  # noqa: E402,F401
from xingtian_amd.algorithm.impala import impala, impala_opt  # noqa: E402,F401
