"""``PpoMlp`` (xt/model/ppo/ppo_mlp.py:29-49) on the HIP learner."""
from xingtian_amd.model import netspec
from xingtian_amd.model.ppo.default_config import MLP_SHARE_LAYERS
from xingtian_amd.model.ppo.ppo import PPO
from xingtian_amd.register import Registers

ACTIVATIONS = ("relu", "tanh")


@Registers.model
class PpoMlp(PPO):
    """Build PPO MLP network."""

    def __init__(self, model_info):
        model_config = model_info.get("model_config") or {}
        self.vf_share_layers = model_config.get("VF_SHARE_LAYERS", MLP_SHARE_LAYERS)
        self.hidden_sizes = model_config.get("hidden_sizes", [64, 64])  # get_mlp_default_settings, model_utils.py:100-107
        self.activation = model_config.get("activation", "tanh")
        if self.activation not in ACTIVATIONS:
            raise KeyError("activation {} not implemented.".format(self.activation))
        super().__init__(model_info)

    def build_spec(self):
        if self.input_dtype != "float32":
            raise ValueError("dtype: {} not supported automatically, please implement it yourself".format(
                self.input_dtype))
        return netspec.ppo_mlp(tuple(self.state_dim), self.action_dim, tuple(self.hidden_sizes), self.activation,
                               self.vf_share_layers, self.action_type)
