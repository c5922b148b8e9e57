"""PPO model: the reference's ``PPO(XTModel)`` (xt/model/ppo/ppo.py:36-132) on HIP kernels.

Same constructor contract (``model_info`` with ``state_dim/action_dim/input_dtype/
model_config``), same ``predict`` / ``train`` / ``get_weights`` / ``set_weights`` /
``save_model`` signatures and return shapes.  ``train`` uploads the rollout once (the
reference re-feeds every minibatch through ``feed_dict``), keeps it resident in HBM and
runs all NUM_SGD_ITER x ceil(N/BATCH_SIZE) steps from one C call; the minibatch gather is
an index indirection inside the first conv kernel.
"""
import numpy as np
import torch

from xingtian_amd.model.model import XTModel, as_numpy, build_net
from xingtian_amd.model.ppo.default_config import (  # noqa: F401
    LR, BATCH_SIZE, CRITIC_LOSS_COEF, ENTROPY_LOSS, LOSS_CLIPPING, MAX_GRAD_NORM, NUM_SGD_ITER, SUMMARY, VF_CLIP)
from xingtian_amd.register import Registers, import_config


@Registers.model
class PPO(XTModel):
    """Build PPO network (xt/model/ppo/ppo.py:36)."""

    def __init__(self, model_info):
        model_config = model_info.get("model_config") or {}
        import_config(globals(), model_config)

        self.state_dim = model_info["state_dim"]
        self.action_dim = model_info["action_dim"]
        self.input_dtype = model_info.get("input_dtype", "float32")

        self.action_type = model_config.get("action_type", "Categorical")
        self._lr = model_config.get("LR", LR)
        self._batch_size = model_config.get("BATCH_SIZE", BATCH_SIZE)
        self.critic_loss_coef = model_config.get("CRITIC_LOSS_COEF", CRITIC_LOSS_COEF)
        self.ent_coef = model_config.get("ENTROPY_LOSS", ENTROPY_LOSS)
        self.clip_ratio = model_config.get("LOSS_CLIPPING", LOSS_CLIPPING)
        self._max_grad_norm = model_config.get("MAX_GRAD_NORM", MAX_GRAD_NORM)
        self.num_sgd_iter = model_config.get("NUM_SGD_ITER", NUM_SGD_ITER)
        self.verbose = model_config.get("SUMMARY", SUMMARY)
        self.vf_clip = model_config.get("VF_CLIP", VF_CLIP)
        self.use_graph = bool(model_config.get("USE_HIP_GRAPH", True))
        self.copy_streams = int(model_config.get("COPY_STREAMS", 2))      # HIP streams the rollout's H2D copies alternate over
        # (adv - adv.mean()) / (adv.std() + 1e-8) over the whole rollout: a COMMENT in the reference
        # (xt/algorithm/ppo/ppo.py:73), hence off unless the configuration asks for it
        self.adv_norm = bool(model_config.get("ADV_NORM", False))
        self.seed = model_config.get("SEED")
        self._rng = np.random.default_rng(self.seed)

        if self.action_type is None:            # learner.py:487 injects it from the environment; Atari YAMLs are discrete
            self.action_type = "Categorical"
        if self.action_type not in ("Categorical", "DiagGaussian"):     # make_dist, xt/model/tf_dist.py:133-139
            raise NotImplementedError(
                "action type: {} not match any implemented distributions.".format(self.action_type))
        self.gauss = self.action_type == "DiagGaussian"
        self._resident = None
        self._ingest = None
        self._dp = None
        self._perm_dense = None
        self._perm_pin = None           # two pinned [NUM_SGD_ITER, n] blocks: this update's shuffles / the next one's
        self._perm_last, self._perm_next = 0, None    # block of the last H2D; block holding shuffles drawn ahead
        # the streaming ingest stages int32 actions: discrete actions; vector observations must be 4-aligned (image
        # observations with another channel count are expanded on the device, RolloutIngest.pad_channels)
        self.stream_ingest = bool(model_config.get("STREAM_INGEST", True)) and not self.gauss and \
            (len(self.state_dim) != 1 or int(self.state_dim[0]) % 4 == 0)
        self.gamma, self.lam = 0.99, 0.95       # GAE of raw trajectories; the PPO algorithm assigns its GAMMA / LAM
        super().__init__(model_info)

    # ---- trunk options shared by the CNN / MLP variants -------------------------------------------------------
    # the reference's ACTIVATION_MAP (xt/model/model_utils.py:8-20), all ten entries.  The eight monotonic ones are
    # fused into the layer kernels (derivative from the stored output); swish and gelu keep the pre-activation of their
    # layers as well (one extra buffer + one elementwise launch per layer: a slow path no bundled configuration takes)
    TRUNK_ACTIVATIONS = ("relu", "tanh", "sigmoid", "softsign", "softplus", "leaky_relu", "elu", "selu", "swish", "gelu")

    def _read_trunk_options(self, model_config, share_default, hidden_default, act_default):
        """VF_SHARE_LAYERS / hidden_sizes / activation with the reference's per-variant defaults
        (get_cnn_default_settings / get_mlp_default_settings, xt/model/model_utils.py:100-117)."""
        cfg = model_config or {}
        self.vf_share_layers = cfg.get("VF_SHARE_LAYERS", share_default)
        self.hidden_sizes = cfg.get("hidden_sizes", list(hidden_default))
        self.activation = cfg.get("activation", act_default)
        if self.activation not in self.TRUNK_ACTIVATIONS:
            raise KeyError("activation {} not implemented.".format(self.activation))

    # subclasses provide build_spec(); create_model wires the HIP network
    def build_spec(self):
        raise NotImplementedError

    def create_model(self, model_info):
        spec = self.build_spec()
        self.net = build_net(model_info, spec, self._batch_size, self.seed)
        self.actor_var = self.net
        if self.net.inference_only:
            self.stream_ingest = False
            return self.net
        base = dict(LR=self._lr, LOSS_CLIPPING=self.clip_ratio, ENTROPY_LOSS=self.ent_coef, VF_CLIP=self.vf_clip,
                    CRITIC_LOSS_COEF=self.critic_loss_coef, MAX_GRAD_NORM=self._max_grad_norm,
                    BATCH_SIZE=self._batch_size, NUM_SGD_ITER=self.num_sgd_iter)
        self._cfg = self.net.make_ppo_cfg(base)
        self._perm_rng = self._rng
        # one rank of a data-parallel learner (torchrun: WORLD_SIZE > 1, or model_config.DP): the replica starts from rank
        # 0's weights, the gradient exchange is installed on the network, the minibatch split follows DP / DP_FEED
        from xingtian_amd.parallel import LearnerDP
        self._dp = LearnerDP.from_config(model_info.get("model_config"), is_learner=model_info.get("type") == "learner")
        if self._dp is not None:
            # (weak: the reported loss is the mean of the ranks' minibatch means; strict: the ranks' shares of one mean)
            self._dp.attach(self.net, loss_scale=1.0 / self._dp.world if self._dp.mode == "weak" else 1.0)
            self._cfg, _ = self._dp.ppo_cfg(self.net, base)
            if self.adv_norm and not (self._dp.mode == "strict" and self._dp.feed == "replicated"):
                # every rank would normalise with the mean / std of ITS trajectories only: not the single-GPU update
                raise ValueError("ADV_NORM under data parallelism needs DP: strict with DP_FEED: replicated (every rank "
                                 "holds the whole rollout); got DP {} / DP_FEED {}".format(self._dp.mode, self._dp.feed))
            if not self._dp.graph_capable:
                self.use_graph = False
            if self._dp.mode == "strict" and self._dp.feed == "replicated":
                self._perm_rng = np.random.default_rng(self._dp.shared_seed(self.seed))      # the SAME shuffles everywhere
            else:
                self._perm_rng = np.random.default_rng(None if self.seed is None else [int(self.seed), self._dp.rank])
        return self.net

    def predict(self, state):
        """-> (action [B] int32 | [B,A] f32, logp [B,1] f32, value [B,1] f32), xt/model/ppo/ppo.py:104-109."""
        state = np.asarray(state)
        logits, value = self.net.forward(state)
        logits = as_numpy(logits)
        value = as_numpy(value).reshape(-1, 1)
        if self.gauss:
            # DiagGaussianDist.sample / log_prob (tf_dist.py:66-69,86-87) on the host; `logits` is the mean
            log_std = self.net.get_weights()["pi_logstd"].reshape(1, -1).astype(np.float32)
            std = np.exp(log_std)
            action = (logits + std * self._rng.standard_normal(logits.shape).astype(np.float32)).astype(np.float32)
            neglogp = np.float32(0.5 * np.log(2.0 * np.pi)) * np.float32(logits.shape[-1]) \
                + 0.5 * np.square((action - logits) / std).sum(-1, keepdims=True) + log_std.sum(-1, keepdims=True)
            return action, (-neglogp).astype(np.float32), value
        # tf.random.categorical (tf_dist.py:127-130): Gumbel-max on the host
        u = self._rng.random(logits.shape)
        action = np.argmax(logits - np.log(-np.log(u)), axis=-1).astype(np.int32)
        m = logits.max(axis=-1, keepdims=True)
        lsm = logits - m - np.log(np.exp(logits - m).sum(axis=-1, keepdims=True))
        logp = np.take_along_axis(lsm, action[:, None].astype(np.int64), axis=1).astype(np.float32)
        return action, logp, value

    def _upload(self, state, label):
        """Keep the rollout resident in HBM; reuse the buffers when the shapes repeat so that a
        captured hipGraph stays valid."""
        dev = self.net.device
        obs = np.ascontiguousarray(state[0])
        n = obs.shape[0]
        key = (obs.shape, str(obs.dtype))
        lay0 = self.net.spec.layers[0]
        pad = lay0.C - obs.shape[1] if (obs.ndim == 2 and lay0.H == lay0.W == 1) else 0   # netspec._mlp zero padding
        cpad = self.net.spec.obs_channels_padded if obs.ndim == 4 else None                # netspec._conv channel padding
        if self._resident is None or self._resident["key"] != key:
            odt = torch.uint8 if self.net.spec.input_xform[0] else torch.float32
            oshape = (n, lay0.C) if pad > 0 else (tuple(obs.shape[:3]) + (cpad,) if cpad else obs.shape)
            self._resident = dict(
                key=key,
                obs=torch.zeros(oshape, dtype=odt, device=dev),
                obs_raw=torch.empty(obs.shape, dtype=odt, device=dev) if cpad else None,
                action=(torch.empty((n, self.action_dim), dtype=torch.float32, device=dev) if self.gauss
                        else torch.empty((n,), dtype=torch.int32, device=dev)),
                old_logp=torch.empty((n,), dtype=torch.float32, device=dev),
                adv=torch.empty((n,), dtype=torch.float64, device=dev),
                old_v=torch.empty((n,), dtype=torch.float32, device=dev),
                target_v=torch.empty((n,), dtype=torch.float64, device=dev),
                perm=torch.empty((self.num_sgd_iter, n), dtype=torch.int32, device=dev))
        r = self._resident
        src = torch.from_numpy(obs).to(r["obs"].dtype) if obs.dtype != np.uint8 and r["obs"].dtype == torch.uint8 \
            else torch.from_numpy(obs)
        if cpad:
            r["obs_raw"].copy_(src, non_blocking=True)
            self.net.pad_obs_channels(r["obs_raw"], out=r["obs"])
        else:
            (r["obs"][:, :obs.shape[1]] if pad > 0 else r["obs"]).copy_(src, non_blocking=True)
        if self.gauss:
            r["action"].copy_(torch.from_numpy(
                np.ascontiguousarray(label[0], dtype=np.float32).reshape(n, self.action_dim)))
        else:
            r["action"].copy_(torch.from_numpy(np.ascontiguousarray(label[0], dtype=np.int32).reshape(-1)))
        r["old_logp"].copy_(torch.from_numpy(np.ascontiguousarray(label[1], dtype=np.float32).reshape(-1)))
        r["adv"].copy_(torch.from_numpy(np.ascontiguousarray(label[2], dtype=np.float64).reshape(-1)))
        r["old_v"].copy_(torch.from_numpy(np.ascontiguousarray(label[3], dtype=np.float32).reshape(-1)))
        r["target_v"].copy_(torch.from_numpy(np.ascontiguousarray(label[4], dtype=np.float64).reshape(-1)))
        return r

    # ---- streaming ingest: trajectories go to HBM as they arrive -------------------
    def ingest_trajectory(self, train_data, pinned=False, slot_guard=None):
        """Called by ``PPO.prepare_data`` for every trajectory; starts its pinned-staging + async H2D copy
        (``pinned``: the arrays already live in page-locked memory -- a pinned transport ring -- and are copied to HBM
        straight from there)."""
        if self._ingest is None:
            from xingtian_amd.ingest import PPO_FIELDS, PPO_RAW_FIELDS, RolloutIngest
            cpad = self.net.spec.obs_channels_padded
            self._ingest = RolloutIngest(self.net.device, self.num_sgd_iter,
                                         obs_u8=bool(self.net.spec.input_xform[0]), fields=PPO_FIELDS + PPO_RAW_FIELDS,
                                         copy_streams=self.copy_streams,
                                         pad_channels=(cpad, self.net.obs_fill_byte()) if cpad else None)
        if "adv" not in train_data:
            # value / reward / done as the explorer holds them before data_proc: GAE runs on the learner GPU, once per
            # rollout, inside train_ingested (no per-message launch, no read-back)
            self._ingest.put_raw(train_data["cur_state"], train_data["action"], train_data["logp"], train_data["value"],
                                 train_data["reward"], train_data["done"], pinned=pinned, slot_guard=slot_guard)
            return
        self._ingest.put(train_data["cur_state"], train_data["action"], train_data["logp"], train_data["adv"],
                         train_data["old_value"], train_data["target_value"], pinned=pinned, slot_guard=slot_guard)

    def ingested(self):
        return 0 if self._ingest is None else self._ingest.n

    def train_ingested(self, perms=None):
        """``train`` on the rollout that was streamed in through ``ingest_trajectory`` (no concat, no upload)."""
        self._require_learner()
        n, d = self._ingest.finish()
        perm = d["perm"][:, :n] if d["perm"].shape[1] == n else None
        if perm is None:
            # capacity > n: the kernel expects perm as a dense [epochs, n] array
            if self._perm_dense is None or self._perm_dense.shape[1] != n:
                self._perm_dense = torch.empty((self.num_sgd_iter, n), dtype=torch.int32, device=self.net.device)
            perm = self._perm_dense
        perm.copy_(self._take_perms(n, perms), non_blocking=True)
        if self._ingest.last_raw_traj:
            from xingtian_amd import lib as L
            self._ingest.gae_on_device(d, n, self.gamma, self.lam, L.stream_ptr())
        self._normalize_adv(d["adv"][:n])
        # the library keeps a small cache of hipGraphs: the two alternating buffer sets replay their own graph
        acc = self.net.ppo_train(self._cfg, d["obs"][:n], perm, d["action"][:n], d["old_logp"][:n], d["adv"][:n],
                                 d["old_v"][:n], d["target_v"][:n], use_graph=self.use_graph)
        self._ingest.mark_consumed()
        return self._finish_update(acc, n)

    def _normalize_adv(self, adv_dev):
        """ADV_NORM: the rollout's float64 advantages normalised in place on the device (C ABI xt_adv_normalize_f64)."""
        if self.adv_norm:
            from xingtian_amd import lib as L
            L.check(self.net.lib.xt_adv_normalize_f64(L.ptr(adv_dev), int(adv_dev.numel()), 1e-8, None, L.stream_ptr()),
                    "xt_adv_normalize_f64")

    def _finish_update(self, acc, n):
        """Everything the host has to do for the NEXT publish / update is done while the GPU runs this one: the D2H of
        the new weights is enqueued behind the update, the next update's epoch shuffles are drawn (0.3 ms of host RNG
        per 4 x 4096), and only then does the host block on the loss."""
        if self.eager_snapshot:
            self.net.snapshot_weights_async()
        self._draw_ahead(n)
        # data parallel: the GLOBAL loss (every rank's share travelled in the tail of the exchanged gradient, summed in
        # rank order on the device: the same bits on every rank) -- no host collective; error bits raise
        a = self.net.read_loss(acc)
        return np.float32(a[0] / max(a[1], 1.0))

    def _perm_block(self, n):
        """The pinned block the NEXT shuffles are written to: the one whose H2D is not the most recent (that copy may
        still be in flight; the other block's copy is at least one whole update old)."""
        if self._perm_pin is None or self._perm_pin[0].shape[1] != n:
            self._perm_pin = [torch.empty((self.num_sgd_iter, n), dtype=torch.int32, pin_memory=True) for _ in range(2)]
            self._perm_last, self._perm_next = 0, None
        return 1 - self._perm_last

    def _draw_ahead(self, n):
        slot = self._perm_block(n)
        self.make_perms(n, out=self._perm_pin[slot].numpy())
        self._perm_next = slot

    def _take_perms(self, n, perms=None):
        """-> pinned int32 [NUM_SGD_ITER, n] tensor with this update's shuffles: injected ``perms``, the ones drawn
        ahead during the previous update (same n), or fresh ones."""
        slot = self._perm_block(n)
        if perms is not None:
            np.copyto(self._perm_pin[slot].numpy(), np.asarray(perms, dtype=np.int32).reshape(self.num_sgd_iter, n))
        elif self._perm_next is None:
            self.make_perms(n, out=self._perm_pin[slot].numpy())
        # (else: block `slot` already holds the shuffles drawn ahead)
        self._perm_next = None
        self._perm_last = slot
        return self._perm_pin[slot]

    def make_perms(self, nbatch, out=None):
        """np.random.shuffle(inds) once per epoch, cumulatively (xt/model/ppo/ppo.py:114-118)."""
        inds = np.arange(nbatch)
        perms = np.empty((self.num_sgd_iter, nbatch), np.int32) if out is None else out
        for ep in range(self.num_sgd_iter):
            self._perm_rng.shuffle(inds)
            perms[ep] = inds
        return perms

    def train(self, state, label, perms=None):
        """state=[obs], label=[action, old_logp, adv, old_v, target_v] -> mean minibatch loss
        (xt/model/ppo/ppo.py:111-132).  ``perms`` ([NUM_SGD_ITER, N]) injects the shuffles."""
        self._require_learner()
        r = self._upload(state, label)
        nbatch = r["obs"].shape[0]
        r["perm"].copy_(self._take_perms(nbatch, perms), non_blocking=True)
        self._normalize_adv(r["adv"])
        acc = self.net.ppo_train(self._cfg, r["obs"], r["perm"], r["action"], r["old_logp"], r["adv"], r["old_v"],
                                 r["target_v"], use_graph=self.use_graph)
        return self._finish_update(acc, nbatch)
