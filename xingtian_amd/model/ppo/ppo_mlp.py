"""``PpoMlp``: PPO on vector observations (reference class of the same name, xt/model/ppo/ppo_mlp.py:29-49)."""
from xingtian_amd.model import netspec
from xingtian_amd.model.ppo.default_config import MLP_SHARE_LAYERS
from xingtian_amd.model.ppo.ppo import PPO
from xingtian_amd.register import Registers


@Registers.model
class PpoMlp(PPO):
    TRUNK_DEFAULTS = ((64, 64), "tanh")        # hidden_sizes, activation (get_mlp_default_settings)

    def __init__(self, model_info):
        self._read_trunk_options(model_info.get("model_config"), MLP_SHARE_LAYERS, *self.TRUNK_DEFAULTS)
        super().__init__(model_info)

    def build_spec(self):
        if self.input_dtype != "float32":
            raise ValueError("dtype: {} not supported automatically, please implement it yourself".format(
                self.input_dtype))
        return netspec.ppo_mlp(tuple(self.state_dim), self.action_dim, tuple(self.hidden_sizes), self.activation,
                               self.vf_share_layers, self.action_type)
