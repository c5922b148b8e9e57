"""``ImpalaCnnOpt`` (xt/model/impala/impala_cnn_opt.py:64-296) on HIP kernels.

conv(SAME) -> conv(SAME) -> 11x11 conv (a dense 3872->256) -> 1x1-conv policy + dense
baseline, v-trace targets (xt/model/impala/vtrace.py) and the sum-form loss
``pi + 0.5*baseline + 0.01*entropy`` (:299-351), global-norm clip (:213-214) and Adam.
The reference pins v-trace to the CPU inside the TF graph (:336); here it is one small
GPU kernel between the forward and backward passes of the same step.
"""
import numpy as np
import torch

from xingtian_amd.model import netspec
from xingtian_amd.model.impala.default_config import GAMMA, LR  # noqa: F401
from xingtian_amd.model.model import XTModel, as_numpy, build_net
from xingtian_amd.register import Registers, import_config


def linear_cosine_decay(learning_rate, global_step, decay_steps, num_periods=0.5, alpha=0.0, beta=0.001):
    """tf.train.linear_cosine_decay in float32 (the reference's ``_get_lr``, impala_cnn_opt.py:236-249, calls it
    with learning_rate = lr_schedule[0][1], decay_steps = 20000 and beta = lr_schedule[1][1] / decay_steps)."""
    f = np.float32
    step = f(min(float(global_step), float(decay_steps)))
    linear_decayed = (f(decay_steps) - step) / f(decay_steps)
    cosine_decayed = f(0.5) * (f(1.0) + np.cos(f(np.pi) * f(2.0) * f(num_periods) * step / f(decay_steps), dtype=f))
    return f(learning_rate) * ((f(alpha) + linear_decayed) * cosine_decayed + f(beta))


@Registers.model
class ImpalaCnnOpt(XTModel):
    """Docstring for ActorNetwork (impala_cnn_opt.py:64)."""

    def __init__(self, model_info):
        model_config = model_info.get("model_config", dict()) or {}
        import_config(globals(), model_config)
        self.input_dtype = model_info.get("input_dtype", "float32")
        self.sta_mean = model_info.get("state_mean", 0.)
        self.sta_std = model_info.get("state_std", 255.)
        self.state_dim = model_info["state_dim"]
        self.action_dim = model_info["action_dim"]
        self.lr_schedule = model_config.get("lr_schedule", None)
        self.opt_type = model_config.get("opt_type", "adam")
        if self.opt_type not in ("adam", "rmsprop"):
            raise KeyError("invalid opt_type: {}".format(self.opt_type))
        if self.lr_schedule and len(self.lr_schedule) != 2:
            raise ValueError("lr_schedule invalid: {} (need 2 elements, like [[0, 0.01], [20000, 0.000001]])".format(
                self.lr_schedule))
        self.lr = LR
        self._global_step = 0          # tf.Variable global_step of apply_gradients (impala_cnn_opt.py:198,217)
        self.grad_norm_clip = model_config.get("grad_norm_clip", 40.0)
        self.sample_batch_steps = model_config.get("sample_batch_step", 50)
        self.max_batch = int(model_config.get("MAX_BATCH", model_info.get("max_batch", 1024)))
        self.seed = model_config.get("SEED")
        self._rng = np.random.default_rng(self.seed)
        self.use_graph = bool(model_config.get("USE_HIP_GRAPH", True))
        self.stream_ingest = bool(model_config.get("STREAM_INGEST", True))
        # ASYNC_LOSS (default off): train() does not wait for the update it has just enqueued -- it returns the loss of the
        # PREVIOUS train (the first call still waits), so the next message is staged and copied to HBM while the GPU runs this
        # one.  Only the reported number lags (xt/framework/learner.py:348-351 logs it); weights handed out afterwards are
        # always the ones this train produced.  Pays when weights do not go out after every train (train_per_checkpoint > 1).
        self.async_loss = bool(model_config.get("ASYNC_LOSS", False))
        # (not with ASYNC_LOSS: the staging block of train k may then be rewritten for train k + 2 while train k still runs)
        self.zero_copy_labels = bool(model_config.get("ZERO_COPY_LABELS", True)) and not self.async_loss
        # IO_TAIL_IN_GRAPH (default on; synchronous loss only): the loss read-back and the weights-ring copy are the train's own
        # last kernels inside its replayed hipGraph, the loss is awaited by polling a page-locked word (xt_train_io.tail_in_graph)
        # (2: the weights copy leaves the graph again -- a device-side snapshot is the train's last kernel and the bus-bound
        # copy snapshot -> ring slot runs on the SDMA engine UNDER THE NEXT TRAIN, issued through the HSA runtime by the ring's
        # committer thread when it waits for the publish: csrc/xt_sdma.hip)
        self.io_tail_in_graph = int(model_config.get("IO_TAIL_IN_GRAPH", 2))
        # HIP streams the frames of consecutive messages alternate between (joined by two event calls per train): ONE for an
        # IMPALA learner -- its few messages per train share the bus anyway, and the join costs the staging thread ~10 us
        self.ingest_copy_streams = int(model_config.get("INGEST_COPY_STREAMS", 1))
        self.ingest_dma = bool(model_config.get("INGEST_DMA", True))
        self._ingest = None
        self._dp = None
        self._lr_host = self._lr_dev = None
        super().__init__(model_info)

    def create_model(self, model_info):
        spec = netspec.impala_cnn_opt(tuple(self.state_dim), self.action_dim, self.sta_mean, self.sta_std,
                                      self.input_dtype)
        self.net = build_net(model_info, spec, self.max_batch, self.seed, init="none")
        self.net.init_weights(self.seed, baseline_norm_std=0.01)   # custom_norm_initializer(0.01), :149
        self.actor_var = self.net
        if self.net.inference_only:
            self.stream_ingest = False
            return True
        self.net.set_optimizer(self.opt_type)
        self._cfg = self.net.make_impala_cfg(self.lr, self.grad_norm_clip, self.sample_batch_steps, GAMMA,
                                             opt_type=self.opt_type)
        # one rank of a data-parallel learner (xingtian_amd/parallel.py::LearnerDP): the sum-form loss makes the SUM of the
        # ranks' gradients the chunk's gradient, no scaling.  strict + replicated feed: whole-trajectory shards of every
        # chunk, taken inside xt_net_impala_train; sharded feeds: every rank trains BATCH_SIZE / N frames of its own
        # messages per chunk (strict) or full BATCH_SIZE chunks (weak: global chunk N x BATCH_SIZE, flagged)
        from xingtian_amd.parallel import LearnerDP
        self._dp = LearnerDP.from_config(model_info.get("model_config"), is_learner=model_info.get("type") == "learner")
        if self._dp is not None:
            self._dp.attach(self.net, loss_scale=1.0)       # sum-form loss: the ranks' shares of one sum
            if not self._dp.graph_capable:
                self.use_graph = False
            if self._dp.mode == "strict" and self._dp.feed == "replicated":
                self._cfg = self.net.make_impala_cfg(self.lr, self.grad_norm_clip, self.sample_batch_steps, GAMMA,
                                                     opt_type=self.opt_type, shard_rank=self._dp.rank,
                                                     shard_world=self._dp.world)
        return True

    # ---- resident rollout: every train goes pinned staging -> async H2D -> ONE C call (hipGraph replay) -------
    def _ingest_obj(self):
        if self._ingest is None:
            from xingtian_amd.ingest import RolloutIngest, impala_fields
            cpad = self.net.spec.obs_channels_padded
            self._ingest = RolloutIngest(self.net.device, 0, initial_capacity=self.max_batch,
                                         obs_u8=bool(self.net.spec.input_xform[0]),
                                         fields=impala_fields(self.action_dim),
                                         pad_channels=(cpad, self.net.obs_fill_byte()) if cpad else None,
                                         copy_streams=self.ingest_copy_streams)
            # the few KB of labels of a train are read by the v-trace kernel straight out of the page-locked staging block
            # (only if page-locked host memory is mapped into the device's address space here: probed once)
            from xingtian_amd import lib as L
            probe = torch.empty(64, dtype=torch.uint8, pin_memory=True)
            self._ingest.zero_copy_labels = bool(self.zero_copy_labels) and L.host_device_ptr(probe.data_ptr()) is not None
            # frames out of page-locked transport slots: SDMA copies with tickets (xt_dma_h2d_async), no stream, no events
            self._ingest.dma_h2d = bool(self.ingest_dma) and self._ingest.zero_copy_labels
        return self._ingest

    def ingest_message(self, states, bp_logic_outs, actions, dones, rewards, pinned=False, slot_guard=None):
        """Called by ``IMPALAOpt.prepare_data`` for every rollout message: its pinned-staging + asynchronous H2D
        copy starts now, so ``train`` finds the rollout resident."""
        self._ingest_obj().put(states, bp_logic_outs, actions, dones, rewards, pinned=pinned, slot_guard=slot_guard)

    def ingested(self):
        return 0 if self._ingest is None else self._ingest.n

    def ingest_generation(self):
        """rollouts the learner has taken over so far (``RolloutIngest.finish`` calls): a ``transport.Prefetcher`` stages
        at most one train ahead of it"""
        return self._ingest_obj().generation

    def _lr_steps(self, n_chunks):
        """Step size of the next ``n_chunks`` updates as a device array, or None when it is the constant LR (the
        reference's rmsprop branch ignores lr_schedule, impala_cnn_opt.py:204-206)."""
        if not (self.lr_schedule and self.opt_type == "adam"):
            return None
        if self._lr_host is None or self._lr_host[0].numel() < n_chunks:
            # two pinned blocks, alternating: with ASYNC_LOSS the host may be a whole train ahead of the copy it enqueued
            self._lr_host = [torch.empty((max(n_chunks, 16),), dtype=torch.float32, pin_memory=True) for _ in range(2)]
            self._lr_dev = torch.empty((max(n_chunks, 16),), dtype=torch.float32, device=self.net.device)
            self._lr_turn = 0
            torch.cuda.current_stream(self.net.device).synchronize()
        host = self._lr_host[self._lr_turn]
        self._lr_turn ^= 1
        step0 = self._global_step
        for i in range(n_chunks):
            self._global_step = step0 + i
            host[i] = float(self.current_lr())
        self._global_step = step0
        self._lr_dev[:n_chunks].copy_(host[:n_chunks], non_blocking=True)
        return self._lr_dev

    def train_ingested(self, batch_size):
        """``IMPALAOpt.train`` (impala_opt.py:73-106) on the rollout streamed in through ``ingest_message``: all
        sequential BATCH_SIZE chunks in one C call -> mean of the chunk losses."""
        self._require_learner()
        n, d = self._ingest.finish(wait_on_stream=False)       # (the train's C call makes the stream wait for the copies)
        dp = self._dp
        if dp is not None:
            if dp.mode == "strict" and dp.feed != "replicated":
                # the reference's chunk of BATCH_SIZE frames = N local chunks of BATCH_SIZE / N frames (whole trajectories)
                if batch_size % (dp.world * self.sample_batch_steps):
                    raise ValueError("strict data parallelism over sharded messages needs BATCH_SIZE ({}) divisible by "
                                     "ranks x sample_batch_step ({} x {})".format(batch_size, dp.world, self.sample_batch_steps))
                batch_size //= dp.world
        n_chunks = (n + batch_size - 1) // batch_size
        lr_steps = self._lr_steps(n_chunks)
        # ONE C call: wait for the rollout's copies -> all chunks (hipGraph replay) -> mark the buffer set consumed -> loss
        # read-back into a pinned block -> [the new parameters into the weights ring's slot, D2H in stream order] -> wait for
        # the loss (GIL released).  (Round 6: these were ~10 Python-level runtime calls around an 85 us train.)
        ring = getattr(self.net, "_wring", None) if self.eager_snapshot else None
        ticket = None
        if ring is not None and getattr(ring, "async_commit", False) and getattr(ring, "pinned", False):
            ticket = ring.publish_reserve(self.net, getattr(self.net, "_wring_ctr", None))
        ing = self._ingest
        lab = ing.mapped_labels(n) if ing.zero_copy_labels else None
        if lab is None:
            if ing.zero_copy_labels:
                raise RuntimeError("ImpalaCnnOpt: the label staging block is not mapped into the device's address space "
                                   "(set model_config ZERO_COPY_LABELS: false)")
            lab = {k: d[k][:n] for k in ("logit", "action", "done", "reward")}
        # (in-graph tail: the call returns right behind the launch; what does not depend on the loss -- handing the publish
        # to the ring's committer, reserving the NEXT publish's slot and header -- happens while the device trains)
        # No event records behind the graph either (each delays the NEXT graph on the stream): the buffer set is consumed once
        # the loss has been seen (the loss kernel runs behind every kernel that reads it), the parameter copy reports its own
        # completion through the mailbox (net.io_publish_done()).
        defer = bool(self.io_tail_in_graph) and not self.async_loss
        a = self.net.impala_train_io(self._cfg, d["obs"][:n], batch_size, lab["logit"], lab["action"], lab["done"],
                                     lab["reward"], lr_steps=lr_steps, use_graph=self.use_graph,
                                     wait_event=ing.last.done if getattr(ing.last, "wait_stream", True) else None,
                                     wait_dma_ticket=ing.last.dma_ticket,
                                     consumed_event=None if defer else ing.consumed_event(),
                                     publish=None if ticket is None else (ticket[3], None if defer else ticket[4]),
                                     wait_loss=not self.async_loss, tail_in_graph=self.io_tail_in_graph, defer=defer)
        try:
            self._global_step += n_chunks
            if a is None:
                ing.last.free = None             # (host-confirmed below: impala_wait_loss returns behind the loss kernel)
            if ticket is not None:
                if a is None:
                    # the ring's committer thread waits through the mailbox: for the in-graph copy kernel's report (mode 1),
                    # or for the snapshot's and then makes the SDMA copy itself, under the next train (mode 2)
                    ticket = ticket[:5] + (self.net.io_publish_done(),)
                ring.publish_enqueued(ticket)
                self.net._wring_version = getattr(self.net, "_version", 0)
                if a is None:
                    ring.publish_prereserve(self.net, getattr(self.net, "_wring_ctr", None))
            elif self.eager_snapshot:
                self.net.snapshot_weights_async()   # (no committer ring: the D2H of the new weights is enqueued behind the train)
        finally:
            if a is None:
                a = self.net.impala_wait_loss()
        # Data parallel: the sum is the GLOBAL one (the ranks' shares travelled in the tail of the exchanged gradient)
        return np.float32(float(a[0]) / max(float(a[1]), 1.0))

    def train(self, state, label):
        """One chunk: state [n,...], label=[bp_logits, actions, dones, rewards] -> loss
        (impala_cnn_opt.py:251-265).  n must be a multiple of sample_batch_step."""
        self._require_learner()
        bp_logic_outs, actions, dones, rewards = label
        ing = self._ingest_obj()
        ing.reset()
        ing.put(state, bp_logic_outs, np.asarray(actions).reshape(-1), np.asarray(dones, dtype=bool).reshape(-1),
                np.asarray(rewards).reshape(-1))
        return self.train_ingested(int(np.asarray(state).shape[0]))

    def extra_optimizer_state(self):
        """``global_step`` drives ``lr_schedule`` (impala_cnn_opt.py:198-203): it belongs to a true resume."""
        return {"global_step": np.int64(self._global_step)}

    def restore_extra_optimizer_state(self, arrays):
        if "global_step" in arrays:
            self._global_step = int(arrays["global_step"])

    def current_lr(self, decay_step=20000.0):
        """Learning rate of the NEXT update (``_get_lr``, impala_cnn_opt.py:236-249)."""
        if not self.lr_schedule:
            return np.float32(self.lr)
        return linear_cosine_decay(self.lr_schedule[0][1], self._global_step, decay_step,
                                   beta=self.lr_schedule[1][1] / float(decay_step))

    def predict(self, state):
        """-> [logits [B,A], baseline [B], action [B]] (impala_cnn_opt.py:267-277)."""
        logits, value = self.net.forward(np.asarray(state))
        logits = as_numpy(logits)
        u = self._rng.random(logits.shape)
        action = np.argmax(logits - np.log(-np.log(u)), axis=-1).astype(np.int32)   # tf.multinomial, :153-157
        return [logits, as_numpy(value), action]
