"""Base class of the learner-side algorithms.

Public surface = what the host framework's learner / explorer processes call on an algorithm object (reference:
xt/algorithm/algorithm.py:34-237): ``prepare_data`` / ``prepare_data_times`` / ``train_ready`` / ``train`` /
``checkpoint_ready`` / ``predict`` / ``save`` / ``restore`` / ``get_weights`` / ``set_weights`` /
``if_save`` / ``weights_map`` / ``dist_model_policy`` / ``async_flag``.  The model behind ``self.actor`` is a
HIP learner model from ``xingtian_amd.model``.
"""
import os

import numpy as np

from xingtian_amd.algorithm.alg_utils import DefaultAlgDistPolicy

AGENT_PREFIX = "agent"
MODEL_PREFIX = "actor"
ZFILL_LENGTH = 5


class RolloutFields(object):
    """Named per-field accumulators for the trajectories of one update: ``add`` one message, ``stacked`` once."""

    def __init__(self, *names):
        self.names = names
        self.parts = {n: [] for n in names}

    def add(self, **arrays):
        """Arrays that do not own their memory are COPIED: a zero-copy transport hands ``prepare_data`` views into a
        slot that the producer overwrites as soon as the call returns (``transport.ShmRing.recv_into``); the reference's
        channel delivers private objects (zeus/common/ipc/share_by_plasma.py:74-95)."""
        for n in self.names:
            v = arrays[n]
            if isinstance(v, np.ndarray) and not v.flags.owndata:
                v = np.array(v, copy=True)
            self.parts[n].append(v)

    def stacked(self):
        return [np.concatenate(self.parts[n]) for n in self.names]

    def reset(self):
        for lst in self.parts.values():
            lst.clear()

    def __len__(self):
        return len(self.parts[self.names[0]])


class Algorithm(object):
    """One trainable model (``actor``) + the bookkeeping the learner loop needs."""

    buff = None
    actor = None

    def __init__(self, alg_name, model_info, alg_config=None, **kwargs):
        from xingtian_amd.model import model_builder
        cfg = alg_config
        self.alg_name, self.alg_config, self.model_info = alg_name, cfg, model_info
        self.actor = model_builder(model_info)
        self.state_dim, self.action_dim = model_info.get("state_dim"), model_info.get("action_dim")
        self.train_count = 0
        self.async_flag = True
        self._train_ready = True
        self._weights_map = self.update_weights_map()
        # how many rollout messages make one update, and who gets the new weights
        self._prepare_times_per_train = cfg.get("prepare_times_per_train", cfg["instance_num"] * cfg["agent_num"])
        self.dist_model_policy = DefaultAlgDistPolicy(cfg["instance_num"], prepare_times=self._prepare_times_per_train)
        self.learning_starts = cfg.get("learning_starts", 0)
        self._train_per_checkpoint = cfg.get("train_per_checkpoint", 1)
        if hasattr(self.actor, "eager_snapshot"):
            self.actor.eager_snapshot = (self._train_per_checkpoint == 1)
        self.if_save_model, self.save_interval = cfg.get("save_model", False), cfg.get("save_interval", 500)

    # ---- data in
    def prepare_data(self, train_data, **kwargs):
        raise NotImplementedError

    @property
    def prepare_data_times(self):
        return self._prepare_times_per_train

    def train_ready(self, elapsed_episode, **kwargs):
        """Replay-buffer algorithms wait for ``learning_starts`` samples; on-policy ones are always ready."""
        waiting = bool(getattr(self, "buff")) and self.learning_starts > 0 and self.buff.size() < self.learning_starts
        self._train_ready = not waiting
        return self._train_ready

    # ---- update
    def train(self, **kwargs):
        raise NotImplementedError

    def predict(self, state):
        return np.argmax(self.actor.predict(state.reshape((1,) + state.shape)))

    # ---- weights out / checkpoints
    @property
    def dp(self):
        """the data-parallel context of this learner rank (xingtian_amd/parallel.py::LearnerDP), or None"""
        return getattr(self.actor, "_dp", None)

    @property
    def is_publisher(self):
        """does THIS learner rank hand weights to explorers and write checkpoints?  (always True without data parallelism)"""
        return self.dp is None or self.dp.is_publisher

    def checkpoint_ready(self, train_count, **kwargs):
        self._train_ready = False
        return self.is_publisher and train_count % self.train_per_checkpoint == 0

    @property
    def train_per_checkpoint(self):
        return self._train_per_checkpoint

    @train_per_checkpoint.setter
    def train_per_checkpoint(self, interval):
        self._train_per_checkpoint = interval
        if hasattr(self.actor, "eager_snapshot"):
            self.actor.eager_snapshot = (interval == 1)

    def if_save(self, train_count):
        if not self.is_publisher:
            return False
        if self.if_save_model and train_count % self.save_interval == 0:
            return True
        return None if self.if_save_model else False

    def save(self, model_path, model_index):
        """-> [path]: ``actor_<index zero-filled to 5>.npz`` under ``model_path``."""
        stem = "{}_{}".format(MODEL_PREFIX, str(model_index).zfill(ZFILL_LENGTH))
        return [self.actor.save_model(os.path.join(model_path, stem))]

    def restore(self, model_name=None, model_weights=None):
        """In-memory weights win over a file name (setting them is cheaper than reading the disk)."""
        if model_weights is None:
            self.actor.load_model(model_name)
        else:
            self.actor.set_weights(model_weights)

    def get_weights(self):
        return self.actor.get_weights()

    def set_weights(self, weights):
        return self.actor.set_weights(weights)

    def publish_weights(self, ring, ctr_info=None, lag=0):
        """``get_weights`` + hand-over to the explorers in one step (``transport.WeightsRing``).  ``lag = 1`` hands out
        the previous update's weights without waiting for the current one (asynchronous algorithms only: a deviation from
        the reference's learner loop, which publishes what the train has just produced)."""
        return self.actor.publish_weights(ring, ctr_info, lag=lag)

    @staticmethod
    def update_weights_map(agent_in_group="agent_0", agent_in_env="agent_0"):
        """{agent_id: {"prefix": ..., "name": ...}} for multi-model setups; every agent shares one model by default."""
        return {}

    @property
    def weights_map(self):
        return self._weights_map

    @weights_map.setter
    def weights_map(self, map_info):
        self._weights_map = map_info

    def shutdown(self):
        pass
