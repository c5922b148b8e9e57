"""``PpoCnn``: PPO on the Atari conv stack (reference class of the same name, xt/model/ppo/ppo_cnn.py:29-50)."""
from xingtian_amd.model import netspec
from xingtian_amd.model.ppo.default_config import CNN_SHARE_LAYERS
from xingtian_amd.model.ppo.ppo import PPO
from xingtian_amd.register import Registers


@Registers.model
class PpoCnn(PPO):
    TRUNK_DEFAULTS = ((512,), "relu")          # hidden_sizes, activation (get_cnn_default_settings)

    def __init__(self, model_info):
        self._read_trunk_options(model_info.get("model_config"), CNN_SHARE_LAYERS, *self.TRUNK_DEFAULTS)
        super().__init__(model_info)

    def build_spec(self):
        return netspec.ppo_cnn(tuple(self.state_dim), self.action_dim, tuple(self.hidden_sizes), self.activation,
                               self.vf_share_layers, self.input_dtype, self.action_type)
