"""``PpoCnn`` (xt/model/ppo/ppo_cnn.py:29-50) on the HIP learner."""
from xingtian_amd.model import netspec
from xingtian_amd.model.ppo.default_config import CNN_SHARE_LAYERS
from xingtian_amd.model.ppo.ppo import PPO
from xingtian_amd.register import Registers

ACTIVATIONS = ("relu", "tanh")


@Registers.model
class PpoCnn(PPO):
    """Build PPO CNN network."""

    def __init__(self, model_info):
        model_config = model_info.get("model_config") or {}
        self.vf_share_layers = model_config.get("VF_SHARE_LAYERS", CNN_SHARE_LAYERS)
        self.hidden_sizes = model_config.get("hidden_sizes", [512])   # get_cnn_default_settings, model_utils.py:110-114
        self.activation = model_config.get("activation", "relu")
        if self.activation not in ACTIVATIONS:
            raise KeyError("activation {} not implemented.".format(self.activation))
        super().__init__(model_info)

    def build_spec(self):
        return netspec.ppo_cnn(tuple(self.state_dim), self.action_dim, tuple(self.hidden_sizes), self.activation,
                               self.vf_share_layers, self.input_dtype, self.action_type)
