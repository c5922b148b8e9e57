"""Oracle (test infrastructure): what the reference's OWN config patching produces for its example YAMLs.

``patch_model_config_by_env_info`` and ``patch_alg_within_config`` (xt/framework/learner.py:481-533) are exec'd from
the file's source text (the module itself imports zmq, absl, ... which are not installed) with ``env_builder`` stubbed
to return the ``env_info`` an Atari / CartPole gym environment reports; ``setup_learner``'s ``type: learner`` mark
(:544) is applied as there.  Input (parsed YAML) and output are committed as tests/golden/learner_config.json, so that
nothing at test time reads /root/reference.

Usage:  python oracle/gen_golden_cfg.py
"""
import copy
import json
import os
import re

import yaml

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "learner_config.json")
# every PPO / IMPALA example of the reference's examples/ directory (the YAML surface north_star asks to keep)
YAMLS = ["examples/cartpole_ppo.yaml", "examples/breakout_ppo.yaml", "examples/breakout_impala.yaml",
         "examples/pong_impala_speedup.yaml", "examples/pendulum_ppo.yaml", "examples/ant_ppo.yaml", "examples/dog_ppo.yaml",
         "examples/beamrider_ppo.yaml", "examples/pong_ppo.yaml", "examples/qbert_ppo.yaml", "examples/spaceinvader_ppo.yaml",
         "examples/beamrider_impala.yaml", "examples/qbert_impala.yaml", "examples/spaceinvader_impala.yaml",
         "examples/cartpole_impala.yaml"]


def main():
    src = open(os.path.join(REF, "xt/framework/learner.py")).read()
    a = src.index("def patch_model_config_by_env_info(")
    b = src.index("def setup_learner(")
    code = src[a:b]
    out = {}
    for rel in YAMLS:
        path = os.path.join(REF, rel)
        if not os.path.exists(path):
            continue
        config = yaml.safe_load(open(path))
        config.setdefault("node_config", [["127.0.0.1", "username", "passwd"]])
        env_info = {"api_type": "standalone",
                    "action_type": "DiagGaussian" if "pendulum" in rel else "Categorical"}

        class _Env(object):
            def get_env_info(self):
                return dict(env_info)

            def close(self):
                pass

        ns = {"env_builder": lambda **kw: _Env()}
        exec(compile(code, "learner.py[481:533]", "exec"), ns)
        patched = ns["patch_alg_within_config"](copy.deepcopy(config))
        alg_para = copy.deepcopy(patched["alg_para"])
        alg_para["model_info"]["actor"].update({"type": "learner"})          # setup_learner, learner.py:544
        out[rel] = {"config": config, "env_info": env_info, "alg_para": alg_para}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT, list(out))


if __name__ == "__main__":
    main()
