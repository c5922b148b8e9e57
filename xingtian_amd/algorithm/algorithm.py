"""Algorithm base class with the reference's surface (xt/algorithm/algorithm.py:34-237)."""
import os

import numpy as np

from xingtian_amd.algorithm.alg_utils import DefaultAlgDistPolicy

AGENT_PREFIX = "agent"
MODEL_PREFIX = "actor"
ZFILL_LENGTH = 5


class Algorithm(object):
    """Build base class for Algorithm."""

    buff = None
    actor = None

    def __init__(self, alg_name, model_info, alg_config=None, **kwargs):
        from xingtian_amd.model import model_builder
        self.actor = model_builder(model_info)
        self.state_dim = model_info.get("state_dim")
        self.action_dim = model_info.get("action_dim")
        self.train_count = 0
        self.alg_name = alg_name
        self.alg_config = alg_config
        self.model_info = model_info
        self.async_flag = True
        self._weights_map = self.update_weights_map()
        self._train_ready = True
        self._prepare_times_per_train = alg_config.get(
            "prepare_times_per_train", alg_config["instance_num"] * alg_config["agent_num"])
        self.dist_model_policy = DefaultAlgDistPolicy(alg_config["instance_num"],
                                                      prepare_times=self._prepare_times_per_train)
        self.learning_starts = alg_config.get("learning_starts", 0)
        self._train_per_checkpoint = alg_config.get("train_per_checkpoint", 1)
        self.if_save_model = alg_config.get("save_model", False)
        self.save_interval = alg_config.get("save_interval", 500)

    def if_save(self, train_count):
        if not self.if_save_model:
            return False
        if train_count % self.save_interval == 0:
            return True

    @staticmethod
    def update_weights_map(agent_in_group="agent_0", agent_in_env="agent_0"):
        return {}

    def prepare_data(self, train_data, **kwargs):
        raise NotImplementedError

    @property
    def prepare_data_times(self):
        return self._prepare_times_per_train

    def predict(self, state):
        inputs = state.reshape((1, ) + state.shape)
        out = self.actor.predict(inputs)
        return np.argmax(out)

    def train_ready(self, elapsed_episode, **kwargs):
        self._train_ready = True
        if getattr(self, "buff") and self.learning_starts > 0:
            if self.buff.size() < self.learning_starts:
                self._train_ready = False
        return self._train_ready

    def train(self, **kwargs):
        raise NotImplementedError

    def checkpoint_ready(self, train_count, **kwargs):
        self._train_ready = False
        if train_count % self.train_per_checkpoint == 0:
            return True
        return False

    @property
    def train_per_checkpoint(self):
        return self._train_per_checkpoint

    @train_per_checkpoint.setter
    def train_per_checkpoint(self, interval):
        self._train_per_checkpoint = interval

    def save(self, model_path, model_index):
        model_name = self.actor.save_model(
            os.path.join(model_path, "actor_{}".format(str(model_index).zfill(ZFILL_LENGTH))))
        return [model_name]

    def restore(self, model_name=None, model_weights=None):
        if model_weights is not None:
            self.actor.set_weights(model_weights)
        else:
            self.actor.load_model(model_name)

    def get_weights(self):
        return self.actor.get_weights()

    def set_weights(self, weights):
        return self.actor.set_weights(weights)

    @property
    def weights_map(self):
        return self._weights_map

    @weights_map.setter
    def weights_map(self, map_info):
        self._weights_map = map_info

    def shutdown(self):
        pass
