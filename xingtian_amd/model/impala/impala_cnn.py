"""``ImpalaCnn`` / ``ImpalaMlp``: the models of the non-opt ``IMPALA`` algorithm on HIP kernels.

Reference: Keras models with a SOFTMAX policy output and a value output, trained by ``model.fit`` (one epoch,
minibatches of 128, shuffled) on the custom ``impala_loss`` + 0.5 * mse with ``tf.keras`` Adam
(xt/model/impala/impala_cnn.py:33-108: ``clipnorm=40.``, ``decay=5.12e-9``; xt/model/impala/impala_mlp.py:30-93: plain
Adam).  ``predict([state, adv])`` returns ``[probabilities, value [N,1]]``; ``train([state, adv], [one-hot, target])``
returns the epoch's sample-weighted mean loss (what ``History.history['loss'][0]`` holds there).
"""
import numpy as np
import torch

from xingtian_amd.model import netspec
from xingtian_amd.model.impala.default_config import ENTROPY_LOSS, HIDDEN_SIZE, LR, NUM_LAYERS  # noqa: F401
from xingtian_amd.model.model import XTModel, as_numpy, build_net
from xingtian_amd.register import Registers, import_config

FIT_BATCH = 128     # model.fit(batch_size=128), impala_cnn.py:76-80 / impala_mlp.py:68-72


class _KerasImpalaModel(XTModel):
    CLIPNORM, DECAY = 0.0, 0.0

    def __init__(self, model_info):
        model_config = model_info.get("model_config") or {}
        import_config(globals(), model_config)
        self.state_dim, self.action_dim = model_info["state_dim"], model_info["action_dim"]
        self.seed = model_config.get("SEED")
        # forward batches of the algorithm's pre-training pass over all stored states; a fit minibatch is 128 rows
        self.max_batch = max(FIT_BATCH, int(model_config.get("MAX_BATCH", model_info.get("max_batch", 1024))))
        self.iterations = 0                       # optimizer.iterations of tf.keras Adam
        super().__init__(model_info)

    def _spec(self):
        raise NotImplementedError

    def create_model(self, model_info):
        self.net = build_net(model_info, self._spec(), self.max_batch, self.seed)
        self.actor_var = self.net
        if self.net.inference_only:
            return True
        self._acc = torch.zeros((2,), dtype=torch.float32, device=self.net.device)
        return True

    def extra_optimizer_state(self):
        return {"keras_adam_iterations": np.int64(self.iterations)}

    def restore_extra_optimizer_state(self, arrays):
        if "keras_adam_iterations" in arrays:
            self.iterations = int(arrays["keras_adam_iterations"])

    def predict(self, state):
        """-> [softmax probabilities [N,A], value [N,1]] (numpy float32); ``state`` = [observations, dummy adv]."""
        logits, value = self.net.forward(np.asarray(state[0]))
        logits = torch.as_tensor(as_numpy(logits)) if self.net.inference_only else logits
        return [as_numpy(torch.softmax(logits, dim=-1)), as_numpy(value).reshape(-1, 1)]

    def train(self, state, label):
        obs, adv = state
        onehot, target = label
        n = len(obs)
        order = np.arange(n)
        np.random.shuffle(order)                  # model.fit(shuffle=True) draws from numpy's global generator
        return self.fit_in_order(obs, adv, onehot, target, order)

    def fit_in_order(self, obs, adv, onehot, target, order):
        """One epoch over the minibatches ``order[0:128], order[128:256], ...`` (the permutation injected)."""
        self._require_learner()
        dev = self.net.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(dev)
        d_obs = self.net.to_device_obs(obs)
        d_adv, d_hot, d_tgt = up(np.asarray(adv).reshape(-1)), up(onehot), up(np.asarray(target).reshape(-1))
        d_order = torch.from_numpy(np.ascontiguousarray(order, dtype=np.int32)).to(dev)
        self._acc.zero_()
        for lo in range(0, len(order), FIT_BATCH):
            self.net.keras_impala_step(d_obs, d_order[lo:lo + FIT_BATCH], d_adv, d_hot, d_tgt, ENTROPY_LOSS, self._acc)
            self.net.adam_keras(LR, self.iterations, clipnorm=self.CLIPNORM, decay=self.DECAY)
            self.iterations += 1
        acc = self._acc.cpu().numpy()
        return float(acc[0] / acc[1])


@Registers.model
class ImpalaCnn(_KerasImpalaModel):
    CLIPNORM, DECAY = 40.0, 0.00000000512

    def _spec(self):
        return netspec.impala_cnn(tuple(self.state_dim), self.action_dim)


@Registers.model
class ImpalaMlp(_KerasImpalaModel):
    def _spec(self):
        return netspec.impala_mlp(tuple(self.state_dim), self.action_dim, HIDDEN_SIZE, NUM_LAYERS)
