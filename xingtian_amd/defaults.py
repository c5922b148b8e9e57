"""Default hyper-parameters of the plugin classes, in one place.

The reference keeps them as module constants next to each class and lets the YAML override them through
``import_config(globals(), config)`` (zeus/common/util/common.py:32-44); the per-package ``default_config`` modules
of this package publish the entries below as module globals so that the same override mechanism applies.
Values: xt/model/ppo/default_config.py, xt/model/impala/default_config.py, xt/algorithm/ppo/default_config.py,
xt/algorithm/impala/default_config.py (pinned by tests/golden/defaults.json, produced by executing those files).
"""

DEFAULTS = {
    "model/ppo": dict(LR=3e-4, BATCH_SIZE=200, NUM_SGD_ITER=4, LOSS_CLIPPING=0.2, ENTROPY_LOSS=1e-3,
                      CRITIC_LOSS_COEF=1.0, VF_CLIP=5.0, MAX_GRAD_NORM=5.0, SUMMARY=False,
                      CNN_SHARE_LAYERS=True, MLP_SHARE_LAYERS=False),
    "model/impala": dict(LR=3e-4, ENTROPY_LOSS=0.01, HIDDEN_SIZE=128, NUM_LAYERS=1, GAMMA=0.99),
    "algorithm/ppo": dict(GAMMA=0.99, LAM=0.95, BATCH_SIZE=512),
    "algorithm/impala": dict(GAMMA=0.99, BATCH_SIZE=512),
}


def publish(namespace, key):
    """Copy DEFAULTS[key] into a module namespace (called by the default_config modules)."""
    namespace.update(DEFAULTS[key])
    namespace["__all__"] = sorted(DEFAULTS[key])
