"""PPO ``Algorithm`` (xt/algorithm/ppo/ppo.py:30-95): accumulate trajectories, train.

Accepts the reference's wire payload unchanged (``cur_state, action, logp, adv,
old_value, target_value``).  If a trajectory arrives WITHOUT ``adv`` but with the raw
``value [T+1,1] / reward / done`` fields, GAE (xt/agent/ppo/ppo.py:77-106) is computed on
the learner GPU by ``xingtian_amd.ops.gae`` -- the north-star move of the return
computation from the actors to the learner.  No advantage normalisation (ppo.py:73 is a
comment in the reference).
"""
import numpy as np

from xingtian_amd.algorithm.algorithm import Algorithm
from xingtian_amd.algorithm.ppo.default_config import GAMMA, LAM  # noqa: F401
from xingtian_amd.register import Registers, import_config


@Registers.algorithm
class PPO(Algorithm):
    """Build PPO algorithm."""

    def __init__(self, model_info, alg_config, **kwargs):
        import_config(globals(), alg_config)
        super().__init__(alg_name=kwargs.get("name") or "ppo", model_info=model_info["actor"],
                         alg_config=alg_config)
        self._init_train_list()
        self.async_flag = False
        if model_info.get("finetune_weight"):
            self.actor.load_model(model_info["finetune_weight"], by_name=True)

    def _init_train_list(self):
        self.obs = list()
        self.behavior_action = list()
        self.old_logp = list()
        self.adv = list()
        self.old_v = list()
        self.target_v = list()
        self._streamed = 0

    def train(self, **kwargs):
        """Train PPO Agent."""
        if self._streamed == len(self.obs) and self._streamed > 0:
            # every trajectory of this rollout was streamed to HBM as it arrived: no concat, no upload
            loss = self.actor.train_ingested(**kwargs)
            self._init_train_list()
            return loss
        if self._streamed:
            self.actor._ingest.reset()
        obs = np.concatenate(self.obs)
        behavior_action = np.concatenate(self.behavior_action)
        old_logp = np.concatenate(self.old_logp)
        adv = np.concatenate(self.adv)
        old_v = np.concatenate(self.old_v)
        target_v = np.concatenate(self.target_v)
        loss = self.actor.train([obs], [behavior_action, old_logp, adv, old_v, target_v], **kwargs)
        self._init_train_list()
        return loss

    def prepare_data(self, train_data, **kwargs):
        if "adv" not in train_data:
            from xingtian_amd import ops
            adv, old_v, tgt = ops.gae(np.asarray(train_data["value"], np.float32).reshape(1, -1),
                                      np.asarray(train_data["reward"], np.float64).reshape(1, -1),
                                      np.asarray(train_data["done"], bool).reshape(1, -1), GAMMA, LAM)
            train_data = dict(train_data, adv=adv.reshape(-1, 1), old_value=old_v.reshape(-1, 1),
                              target_value=tgt.reshape(-1, 1))
        if getattr(self.actor, "stream_ingest", False) and hasattr(self.actor, "ingest_trajectory"):
            self.actor.ingest_trajectory(train_data)
            self._streamed += 1
        self.obs.append(train_data["cur_state"])
        self.behavior_action.append(train_data["action"])
        self.old_logp.append(train_data["logp"])
        self.adv.append(train_data["adv"])
        self.old_v.append(train_data["old_value"])
        self.target_v.append(train_data["target_value"])

    def predict(self, state):
        """Overwrite the predict function, owing to the special input."""
        if not isinstance(state, (list, tuple)):
            state = state.reshape((1,) + state.shape)
        else:
            state = list(map(lambda x: x.reshape((1,) + x.shape), state))
            state = np.vstack(state)
        return self.actor.predict(state)
