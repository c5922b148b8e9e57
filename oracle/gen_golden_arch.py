"""Oracle (test infrastructure): the reference's architecture tables, produced by EXECUTING the reference.

``xt/model/atari_model.py::get_atari_filter`` (no imports) and ``xt/model/model_utils.py``'s
``get_default_filters`` / ``get_mlp_default_settings`` / ``get_cnn_default_settings`` are loaded from
/root/reference with importlib; model_utils.py's two imports (``xt.model.tf_compat``: Keras symbols, and
``xt.model.tf_utils``) are stubbed with attribute sinks because the table functions do not touch them.
Writes tests/golden/arch_tables.json; nothing at test time reads /root/reference.

Usage:  python oracle/gen_golden_arch.py
"""
import importlib.util
import json
import os
import sys
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "arch_tables.json")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    for n in ("xt", "xt.model"):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m
    tfc = types.ModuleType("xt.model.tf_compat")
    class _Any(object):                       # attribute sink: model_utils.py builds an ACTIVATION_MAP from tf.* at import
        def __getattr__(self, name):
            return _Any()
    for sym in ("K", "Conv2D", "Input", "Lambda", "Flatten", "Model", "Dense", "Concatenate", "tf"):
        setattr(tfc, sym, _Any())
    sys.modules["xt.model.tf_compat"] = tfc
    tfu = types.ModuleType("xt.model.tf_utils")
    tfu.gelu = tfu.norm_initializer = object()
    sys.modules["xt.model.tf_utils"] = tfu
    mu = _load("_ref_model_utils", "xt/model/model_utils.py")
    am = _load("_ref_atari_model", "xt/model/atari_model.py")
    norm = lambda f: [[int(c), int(k[0] if isinstance(k, (tuple, list)) else k),
                       int(s[0] if isinstance(s, (tuple, list)) else s)] for c, k, s in f]
    out = {
        "ppo_cnn_filters": {"%dx%d" % (h, h): norm(mu.get_default_filters((h, h, 4))) for h in (84, 42, 15)},
        "impala_filters": {"%dx%d" % (h, h): norm(am.get_atari_filter((h, h, 4))) for h in (84, 42)},
        "mlp_defaults": {"hidden_sizes": mu.get_mlp_default_settings("hidden_sizes"),
                         "activation": mu.get_mlp_default_settings("activation")},
        "cnn_defaults": {"hidden_sizes": mu.get_cnn_default_settings("hidden_sizes"),
                         "activation": mu.get_cnn_default_settings("activation")},
    }
    # any other square observation: the reference INFERS an architecture (model_utils.py:150-176)
    out["ppo_cnn_inferred"] = {"%dx%d" % (h, h): norm(mu.get_default_filters((h, h, 4)))
                               for h in (3, 7, 10, 30, 64, 65, 100, 128, 200)}
    try:
        mu.get_default_filters((84, 84))
        out["ppo_cnn_rank2_raises"] = False
    except ValueError:
        out["ppo_cnn_rank2_raises"] = True
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", OUT)


if __name__ == "__main__":
    main()


def dump_defaults():
    """tests/golden/defaults.json: the module constants of the reference's four default_config files, read by
    EXECUTING them (xt/model/{ppo,impala}/default_config.py, xt/algorithm/{ppo,impala}/default_config.py)."""
    import json
    out = {}
    for key, rel in [("model/ppo", "xt/model/ppo/default_config.py"), ("model/impala", "xt/model/impala/default_config.py"),
                     ("algorithm/ppo", "xt/algorithm/ppo/default_config.py"),
                     ("algorithm/impala", "xt/algorithm/impala/default_config.py")]:
        ns = {}
        with open(os.path.join("/root/reference", rel)) as f:
            exec(f.read(), ns)
        out[key] = {k: v for k, v in ns.items() if k.isupper()}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "defaults.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    dump_defaults()
