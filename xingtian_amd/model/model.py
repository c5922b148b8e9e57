"""Model base class with the reference's ``XTModel`` surface (xt/model/model.py:30-136),
minus the TensorFlow graph/session: the network lives in a ``HipActorCritic``."""
import glob
import os
from collections import OrderedDict

import numpy as np


class XTModel(object):
    """Model Base class for model module (xt/model/model.py:30)."""

    def __init__(self, model_info):
        self.actor_var = None
        self._summary = model_info.get("summary", False)
        self.model_format = model_info.get("model_format")
        self.max_to_keep = model_info.get("max_to_keep", 100)
        # beyond the reference: Adam slots ride along in the .npz so that a restore is a true resume
        self.save_optimizer = bool((model_info.get("model_config") or {}).get("SAVE_OPTIMIZER", True))
        self.model = self.create_model(model_info)
        if "init_weights" in model_info:
            model_name = model_info["init_weights"]
            try:
                self.load_model(model_name)
                print("load weight: {} success.".format(model_name))
            except BaseException:
                print("load weight: {} failed!".format(model_name))

    def create_model(self, model_info):
        """Abstract method for creating model."""
        raise NotImplementedError

    def predict(self, state):
        raise NotImplementedError

    def train(self, state, label):
        raise NotImplementedError

    def set_weights(self, weights):
        """Set weight with memory tensor (name -> ndarray dict)."""
        self.net.set_weights(weights)

    def get_weights(self):
        """Get the weights (name -> ndarray dict)."""
        return self.net.get_weights()

    def save_model(self, file_name):
        """np.savez of {tf_var_name: ndarray} (TFVariables.save_weights, tf_utils.py:130-134; rotation
        xt/model/model.py:104-108).  With SAVE_OPTIMIZER (default) the Adam slots are added under TF1's own slot
        names; the reference's loader skips names it does not know, so the file stays loadable there."""
        if self.max_to_keep > -1:
            check_keep_model(os.path.dirname(file_name), self.max_to_keep)
        payload = OrderedDict(self.get_weights())
        if self.save_optimizer and hasattr(self.net, "get_optimizer_state"):
            payload.update(self.net.get_optimizer_state())
        np.savez(file_name + ".npz", **payload)
        return file_name + ".npz"

    def load_model(self, model_name, by_name=False):
        """Weights by TF variable name (TFVariables.set_weights_with_npz, tf_utils.py:141-144); Adam slots too
        when the file carries a complete set, else the optimizer is left as it is."""
        np_file = np.load(model_name)
        weights = OrderedDict(**np_file)
        self.set_weights(weights)
        if hasattr(self.net, "set_optimizer_state"):
            self.optimizer_restored = self.net.set_optimizer_state(weights)


def check_keep_model(model_path, keep_num):
    """Check model saved count under path (xt/model/model.py:130-136)."""
    target_file = glob.glob(os.path.join(model_path, "actor*"))
    if len(target_file) > keep_num:
        to_rm_model = sorted(target_file, reverse=True)[keep_num:]
        for item in to_rm_model:
            os.remove(item)
