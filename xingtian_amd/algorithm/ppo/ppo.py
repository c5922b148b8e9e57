"""PPO ``Algorithm`` (xt/algorithm/ppo/ppo.py:30-95): accumulate trajectories, train.

Accepts the reference's wire payload unchanged (``cur_state, action, logp, adv,
old_value, target_value``).  If a trajectory arrives WITHOUT ``adv`` but with the raw
``value [T+1,1] / reward / done`` fields, GAE (xt/agent/ppo/ppo.py:77-106) is computed on
the learner GPU by ``xingtian_amd.ops.gae`` -- the north-star move of the return
computation from the actors to the learner.  No advantage normalisation (ppo.py:73 is a
comment in the reference).
"""
import numpy as np

from xingtian_amd.algorithm.algorithm import Algorithm, RolloutFields
from xingtian_amd.algorithm.ppo.default_config import GAMMA, LAM  # noqa: F401
from xingtian_amd.register import Registers, import_config


@Registers.algorithm
class PPO(Algorithm):
    """Synchronous PPO learner: collect ``prepare_data_times`` trajectories, one ``Model.train`` per update."""

    FIELDS = ("cur_state", "action", "logp", "adv", "old_value", "target_value")

    def __init__(self, model_info, alg_config, **kwargs):
        import_config(globals(), alg_config)
        super().__init__(alg_name=kwargs.get("name") or "ppo", model_info=model_info["actor"], alg_config=alg_config)
        self.async_flag = False
        self._rollout = RolloutFields(*self.FIELDS)
        self._streamed = 0
        # GAMMA / LAM belong to the algorithm's configuration (xt/algorithm/ppo/default_config.py): the model's
        # learner-side GAE (trajectories arriving without advantages) uses them
        if hasattr(self.actor, "gamma"):
            self.actor.gamma, self.actor.lam = float(GAMMA), float(LAM)
        if model_info.get("finetune_weight"):
            self.actor.load_model(model_info["finetune_weight"], by_name=True)

    # the reference keeps one python list per field under these names (xt/algorithm/ppo/ppo.py:59-65)
    obs = property(lambda self: self._rollout.parts["cur_state"])
    behavior_action = property(lambda self: self._rollout.parts["action"])
    old_logp = property(lambda self: self._rollout.parts["logp"])
    adv = property(lambda self: self._rollout.parts["adv"])
    old_v = property(lambda self: self._rollout.parts["old_value"])
    target_v = property(lambda self: self._rollout.parts["target_value"])

    def _forget_rollout(self):
        self._rollout.reset()
        self._streamed = 0
        if self.dp is not None:
            self.dp.new_rollout()

    def prepare_data(self, train_data, **kwargs):
        if self.dp is not None and not self.dp.takes(train_data):
            return                  # DP_FEED round_robin: this trajectory belongs to another learner rank
        streaming = getattr(self.actor, "stream_ingest", False) and hasattr(self.actor, "ingest_trajectory")
        if "adv" not in train_data and not streaming:
            # raw value/reward/done without the streaming ingest (continuous actions, odd vector widths): GAE on the
            # learner GPU per message; the streaming path below batches it into ONE launch per rollout instead
            from xingtian_amd import ops
            adv, old_v, tgt = ops.gae(np.asarray(train_data["value"], np.float32).reshape(1, -1),
                                      np.asarray(train_data["reward"], np.float64).reshape(1, -1),
                                      np.asarray(train_data["done"], bool).reshape(1, -1), GAMMA, LAM)
            train_data = dict(train_data, adv=adv.reshape(-1, 1), old_value=old_v.reshape(-1, 1),
                              target_value=tgt.reshape(-1, 1))
        if streaming:
            ctr = kwargs.get("ctr_info") or {}
            pinned = bool(ctr.get("_pinned_views"))                          # views into a pinned transport ring
            self.actor.ingest_trajectory(train_data, pinned=pinned, slot_guard=ctr.get("_slot_guard"))   # H2D starts now
            self._streamed += 1
            # the staging buffers hold the copy: keep no reference to the arriving arrays (they may be zero-copy views
            # into a transport slot that is recycled as soon as this call returns, xingtian_amd/transport.py)
            self._rollout.add(**{k: None for k in self.FIELDS})
            return
        self._rollout.add(**{k: train_data[k] for k in self.FIELDS})

    def train(self, **kwargs):
        """No advantage normalisation (a comment in the reference, xt/algorithm/ppo/ppo.py:73).  The learner thread
        calls ``train(episode_num=...)`` (xt/framework/learner.py:348); the reference ignores its kwargs, here only
        ``perms`` (injected epoch shuffles, for tests) is understood by the model."""
        perms = kwargs.get("perms")
        streamed_all = self._streamed > 0 and self._streamed == len(self._rollout)
        if streamed_all:                 # the rollout already sits in HBM: no concat, no upload
            loss = self.actor.train_ingested(perms=perms)
        else:
            if self._streamed:
                raise RuntimeError("PPO.train: the rollout was only partly streamed to the device")
            obs, *labels = self._rollout.stacked()
            loss = self.actor.train([obs], labels, perms=perms)
        self._forget_rollout()
        return loss

    def predict(self, state):
        """One state (or a list of per-agent states) -> a batch for ``Model.predict``."""
        if isinstance(state, (list, tuple)):
            batch = np.vstack([s.reshape((1,) + s.shape) for s in state])
        else:
            batch = state.reshape((1,) + state.shape)
        return self.actor.predict(batch)
