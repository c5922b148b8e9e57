"""Base class of the HIP learner models.

Same outward contract as the reference's ``XTModel`` (xt/model/model.py:30-136) -- ``Model(model_info)``,
``create_model`` hook, ``predict`` / ``train``, name-keyed ``get_weights`` / ``set_weights``, ``save_model`` /
``load_model`` on ``.npz`` files keyed by TF variable name, ``max_to_keep`` rotation, optional ``init_weights`` --
without a TensorFlow graph or session: the network is a ``HipActorCritic`` held in ``self.net``.
"""
import glob
import os
from collections import OrderedDict

import numpy as np


def check_keep_model(model_path, keep_num):
    """Checkpoint rotation: called BEFORE a save, deletes all but the ``keep_num`` newest ``actor*`` files
    (newest = last in lexicographic order, which the zero-padded index makes chronological)."""
    files = sorted(glob.glob(os.path.join(model_path, "actor*")))
    for stale in files[:max(0, len(files) - keep_num)]:
        os.remove(stale)


def wants_cpu_replica(model_info):
    """True in the processes that only ever call ``predict``: explorers, evaluators and the predictor are started
    with ``CUDA_VISIBLE_DEVICES=-1`` (xt/framework/explorer.py:60, predictor.py:88) and only the learner marks its
    model ``type: learner`` (xt/framework/learner.py:544).  There the model is the numpy replica
    (``cpu_net.CpuActorCritic``: same names, same flat layout, no ``train``).  The LEARNER never takes this route:
    without a GPU it fails loudly in ``HipActorCritic``.  ``model_config.DEVICE`` ("cpu" | "gpu") overrides."""
    dev = (model_info.get("model_config") or {}).get("DEVICE")
    if dev is not None:
        if dev not in ("cpu", "gpu"):
            raise ValueError("model_config.DEVICE must be 'cpu' or 'gpu', got {!r}".format(dev))
        return dev == "cpu"
    if model_info.get("type") == "learner":
        return False
    import torch
    return not torch.cuda.is_available()


def learner_device(model_info):
    """cuda:0, or -- when the process is one rank of a data-parallel learner (``WORLD_SIZE > 1`` and ``model_config.DP`` not
    "off", xingtian_amd/parallel.py::LearnerDP) -- the rank's own GPU (``model_config.DP_DEVICE`` or ``LOCAL_RANK``)."""
    cfg = model_info.get("model_config") or {}
    implicit = model_info.get("type") == "learner" and cfg.get("DP", "auto") != "off"
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and (implicit or cfg.get("DP") in ("strict", "weak")):
        from xingtian_amd.parallel import LearnerDP
        return "cuda:{}".format(LearnerDP.device_index(cfg))
    return "cuda:0"


def build_net(model_info, spec, max_batch, seed, init="glorot"):
    """The network object behind a model: HIP learner network, or the inference-only CPU replica (see above)."""
    if wants_cpu_replica(model_info):
        from xingtian_amd.model.cpu_net import CpuActorCritic
        return CpuActorCritic(spec, seed=seed, init=init)
    from xingtian_amd.model.hip_net import HipActorCritic
    return HipActorCritic(spec, max_batch=max_batch, device=learner_device(model_info), seed=seed, init=init)


def as_numpy(x):
    """device tensor or ndarray -> ndarray (``predict`` post-processing is shared by both network kinds)"""
    return x if isinstance(x, np.ndarray) else x.cpu().numpy()


class XTModel(object):
    # every update enqueues the D2H of its new weights behind itself (the learner hands them out after every train when
    # train_per_checkpoint = 1, xt/framework/learner.py:361-363); the algorithm clears this when weights only go out every
    # k-th train (pong_impala_speedup.yaml: 3) -- the copy is then issued when they are asked for
    eager_snapshot = True

    def __init__(self, model_info):
        cfg = model_info.get("model_config") or {}
        self.actor_var = None
        self.model_format = model_info.get("model_format")
        self.max_to_keep = model_info.get("max_to_keep", 100)
        self._summary = model_info.get("summary", False)
        # beyond the reference: Adam slots ride along in the .npz so that a restore is a true resume
        self.save_optimizer = bool(cfg.get("SAVE_OPTIMIZER", True))
        self.optimizer_restored = False
        self.model = self.create_model(model_info)
        start_from = model_info.get("init_weights")
        if start_from is not None:
            try:
                self.load_model(start_from)
                print("load weight: {} success.".format(start_from))
            except BaseException:      # the reference also only reports a bad init_weights path
                print("load weight: {} failed!".format(start_from))

    # ---- hooks of the concrete models
    def create_model(self, model_info):
        raise NotImplementedError

    def predict(self, state):
        raise NotImplementedError

    def train(self, state, label):
        raise NotImplementedError

    def _require_learner(self):
        if getattr(self.net, "inference_only", False):
            raise RuntimeError("{}: this is the inference-only CPU replica (explorer / evaluator process); the learner "
                               "update runs only on the GPU (model_info['type'] == 'learner')".format(type(self).__name__))

    # ---- weights by TF variable name
    def get_weights(self):
        return self.net.get_weights()

    def set_weights(self, weights):
        self.net.set_weights(weights)

    def publish_weights(self, ring, ctr_info=None, lag=0):
        """Hand the current weights to the explorers through a ``transport.WeightsRing`` (the learner's
        ``get_weights`` + ``_dist_policy`` pair, xt/framework/learner.py:361-366) without building an intermediate
        dict of private arrays; on a page-locked ring the parameters travel HBM -> slot with one DMA."""
        if getattr(ring, "pinned", False) and getattr(self.net, "_wring", None) is not ring:
            self.net.attach_weights_ring(ring)      # later updates copy their weights straight into the ring
        return self.net.publish_weights(ring, ctr_info, lag=lag)

    # ---- checkpoints
    def save_model(self, file_name):
        """``<file_name>.npz`` = {tf_variable_name: ndarray} as ``TFVariables.save_weights`` writes it
        (xt/model/tf_utils.py:130-134).  With SAVE_OPTIMIZER (default) the optimizer slots are added under TF1's own
        slot names; the reference's loader skips names it does not know, so the file stays loadable there."""
        if self.max_to_keep > -1:
            check_keep_model(os.path.dirname(file_name), self.max_to_keep)
        # (the learner network hands out views into its pinned snapshot: np.savez consumes them at once)
        arrays = OrderedDict(self.get_weights() if getattr(self.net, "inference_only", False)
                             else self.net.get_weights(copy=False))
        if self.save_optimizer and hasattr(self.net, "get_optimizer_state"):
            arrays.update(self.net.get_optimizer_state())
            arrays.update(self.extra_optimizer_state())
        path = file_name + ".npz"
        np.savez(path, **arrays)
        return path

    def load_model(self, model_name, by_name=False):
        """Weights by variable name (``set_weights_with_npz``, tf_utils.py:141-144); optimizer slots too when the
        file carries a complete set (``self.optimizer_restored``), else the optimizer is left as it is."""
        arrays = OrderedDict(np.load(model_name).items())
        self.set_weights(arrays)
        if hasattr(self.net, "set_optimizer_state"):
            self.optimizer_restored = self.net.set_optimizer_state(arrays)
            if self.optimizer_restored:
                self.restore_extra_optimizer_state(arrays)

    # ---- optimizer state that lives on the host side of a concrete model (e.g. tf.keras Adam's ``iterations``)
    def extra_optimizer_state(self):
        return {}

    def restore_extra_optimizer_state(self, arrays):
        pass
