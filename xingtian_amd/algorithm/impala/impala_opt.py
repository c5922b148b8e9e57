"""``IMPALAOpt`` (xt/algorithm/impala/impala_opt.py:37-147): concat messages, sequential
BATCH_SIZE chunks (no shuffle), mean of chunk losses."""
import os

import numpy as np

from xingtian_amd.algorithm.algorithm import Algorithm
from xingtian_amd.algorithm.alg_utils import FIFODistPolicy
from xingtian_amd.algorithm.impala.default_config import BATCH_SIZE
from xingtian_amd.register import Registers, import_config


@Registers.algorithm
class IMPALAOpt(Algorithm):
    """Build IMPALA algorithm."""

    def __init__(self, model_info, alg_config, **kwargs):
        import_config(globals(), alg_config)
        actor_info = dict(model_info["actor"])
        actor_info.setdefault("max_batch", BATCH_SIZE)
        super().__init__(alg_name="impala", model_info=actor_info, alg_config=alg_config)
        self.states = list()
        self.behavior_logits = list()
        self.actions = list()
        self.dones = list()
        self.rewards = list()
        self.async_flag = False
        self.dist_model_policy = FIFODistPolicy(alg_config["instance_num"],
                                                prepare_times=self._prepare_times_per_train)

    def train(self, **kwargs):
        """Train impala agent."""
        states = np.concatenate(self.states)
        behavior_logits = np.concatenate(self.behavior_logits)
        actions = np.concatenate(self.actions)
        dones = np.concatenate(self.dones)
        rewards = np.concatenate(self.rewards)
        nbatch = len(states)
        count = (nbatch + BATCH_SIZE - 1) // BATCH_SIZE
        loss_list = []
        for start in range(count):
            start_index = start * BATCH_SIZE
            env_index = start_index + BATCH_SIZE
            actor_loss = self.actor.train(
                states[start_index:env_index],
                [behavior_logits[start_index:env_index], actions[start_index:env_index],
                 dones[start_index:env_index], rewards[start_index:env_index]])
            loss_list.append(actor_loss)
        self.states.clear()
        self.behavior_logits.clear()
        self.actions.clear()
        self.dones.clear()
        self.rewards.clear()
        return np.mean(loss_list)

    def save(self, model_path, model_index):
        actor_name = "actor" + str(model_index).zfill(5)
        actor_name = self.actor.save_model(os.path.join(model_path, actor_name))
        return [actor_name.split("/")[-1]]

    def prepare_data(self, train_data, **kwargs):
        state, logit, action, done, reward = self._data_proc(train_data)
        self.states.append(state)
        self.behavior_logits.append(logit)
        self.actions.append(action)
        self.dones.append(done)
        self.rewards.append(reward)

    def predict(self, state):
        return self.actor.predict(state)

    @staticmethod
    def _data_proc(episode_data):
        states = episode_data["cur_state"]
        behavior_logits = episode_data["logit"]
        actions = episode_data["action"]
        dones = np.asarray(episode_data["done"], dtype=bool)
        rewards = np.asarray(episode_data["reward"])
        return states, behavior_logits, actions, dones, rewards
