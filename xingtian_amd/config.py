"""From a XingTian YAML to the learner's ``Algorithm`` -- the few lines of host glue between the reference's config
surface and ``alg_builder``: what ``patch_alg_within_config`` (xt/framework/learner.py:491-533),
``patch_model_config_by_env_info`` (:481-489) and ``setup_learner`` (:536-544) do to ``alg_para`` before the
learner calls ``alg_builder(**alg_para)`` (xt/framework/trainer.py:25, agent_group.py:153-166).

The reference builds the environment once to read ``env_info`` (``api_type``, ``action_type``); environments are out
of scope here, so the caller passes ``env_info`` (Atari / CartPole: ``{"api_type": "standalone",
"action_type": "Categorical"}``, the default).  ``tests/golden/learner_config.json`` holds what the reference's own
functions produce for the bundled example YAMLs (executed by ``oracle/gen_golden_cfg.py``); this module is tested
against it.
"""
import copy

import yaml

# the literal the reference injects (learner.py:498-508: "for quickly run 2s_vs_1sc map")
ENV_ATTR = {"state_shape": 27, "obs_shape": 17, "n_actions": 7, "n_agents": 2, "episode_limit": 300,
            "api_type": "standalone", "agent_ids": [0]}
DEFAULT_ENV_INFO = {"api_type": "standalone", "action_type": "Categorical"}


def load_yaml(path):
    with open(path) as f:
        return yaml.safe_load(f)


def patch_alg_within_config(config, env_info=None, node_type="node_config"):
    """``config``: the parsed YAML.  Returns a deep copy whose ``alg_para`` carries ``alg_config`` with the injected
    ``instance_num`` (= env_num x number of nodes), ``agent_num``, ``env_attr``, ``api_type`` and the patched
    ``model_info`` (``model_config.action_type`` + the env_attr keys), exactly as the reference does."""
    config = copy.deepcopy(config)
    env_info = dict(DEFAULT_ENV_INFO if env_info is None else env_info)
    alg_para = dict(config["alg_para"])
    agent_para = config["agent_para"]
    node_config = config.get(node_type) or [("127.0.0.1", "username", "passwd")]      # the YAML default: one local node
    if "alg_config" not in alg_para or alg_para["alg_config"] is None:
        alg_para["alg_config"] = dict()
    alg_para["alg_config"].update({"instance_num": config["env_num"] * len(node_config),
                                   "agent_num": agent_para.get("agent_num", 1),
                                   "env_attr": dict(ENV_ATTR), "api_type": env_info.get("api_type")})
    model_info = config["model_para"]
    if "model_config" not in model_info["actor"] or model_info["actor"]["model_config"] is None:
        model_info["actor"]["model_config"] = dict()
    model_info["actor"]["model_config"].update({"action_type": env_info.get("action_type")})
    model_info["actor"]["model_config"].update(ENV_ATTR)
    alg_para["model_info"] = model_info
    config["alg_para"] = alg_para
    return config


def learner_alg_para(config, env_info=None):
    """``alg_para`` as ``setup_learner`` hands it to the learner: the model is marked ``type: learner``."""
    alg_para = copy.deepcopy(patch_alg_within_config(config, env_info)["alg_para"])
    alg_para["model_info"]["actor"].update({"type": "learner"})
    return alg_para


def build_learner_algorithm(config_or_path, env_info=None):
    """YAML (path or parsed dict) -> the learner's ``Algorithm`` on the GPU (``alg_builder(**alg_para)``)."""
    from xingtian_amd.algorithm import alg_builder
    config = load_yaml(config_or_path) if isinstance(config_or_path, str) else config_or_path
    return alg_builder(**learner_alg_para(config, env_info))


def build_explorer_model(config_or_path, env_info=None):
    """The same configuration as an explorer / evaluator process builds it (no ``type``): the CPU replica when no GPU
    is visible."""
    from xingtian_amd.model import model_builder
    config = load_yaml(config_or_path) if isinstance(config_or_path, str) else config_or_path
    return model_builder(patch_alg_within_config(config, env_info)["alg_para"]["model_info"]["actor"])
