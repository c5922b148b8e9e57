"""Device-resident actor-critic network driven through the C ABI.

PyTorch-ROCm tensors are storage only (flat parameter / gradient / Adam buffers, the
activation workspace, the resident rollout); every arithmetic op is a HIP kernel in
``libxt_mi355x.so``.  Used by the ``PPO`` / ``ImpalaCnnOpt`` model classes.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from xingtian_amd import lib as L
from xingtian_amd.model.cpu_net import initial_weights


class _MailboxDone(object):
    """see ``HipActorCritic.io_publish_done``"""
    __slots__ = ("net", "seq")

    def __init__(self, net, seq):
        self.net, self.seq = net, seq

    def synchronize(self):
        rc = self.net.lib.xt_net_io_publish_wait(self.net.handle, self.seq, 30000)        # (ctypes releases the GIL)
        if rc == 1:
            raise RuntimeError("xingtian_amd: the parameter copy of train {} did not land within 30 s".format(self.seq))
        L.check(rc, "xt_net_io_publish_wait")

    def query(self):
        return self.net.lib.xt_net_io_publish_wait(self.net.handle, self.seq, 0) == 0


class HipActorCritic(object):
    def __init__(self, spec, max_batch, device="cuda:0", seed=None, init="glorot"):
        L.require_gpu()
        self.lib = L.load()
        self.spec, self.max_batch = spec, int(max_batch)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        n = spec.n_flat
        self.params = torch.zeros(n, dtype=torch.float32, device=self.device)
        # the gradient buffer carries XT_DP_TAIL_FLOATS behind align4(n): the data-parallel tail (rows / loss share of every
        # rank, include/xt_mi355x.h `xt_net_set_dp`) travels with the gradient in ONE exchange; `grads` is the gradient view
        self.grads_xchg = torch.zeros(((n + 3) // 4) * 4 + L.DP_TAIL_FLOATS, dtype=torch.float32, device=self.device)
        self.grads = self.grads_xchg[:n]
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=self.device)
        self.adam_state = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.loss_acc = torch.zeros(8, dtype=torch.float32, device=self.device)
        self.loss_out = torch.zeros(8, dtype=torch.float32, device=self.device)

        lay_arr = (L.LayerDesc * len(spec.layers))()
        for i, lay in enumerate(spec.layers):
            g = lay_arr[i].g
            g.H, g.W, g.C, g.KH, g.KW, g.S = lay.H, lay.W, lay.C, lay.KH, lay.KW, lay.S
            g.PT, g.PL, g.OH, g.OW, g.N, g.act = lay.PT, lay.PL, lay.OH, lay.OW, lay.N, L.ACT[lay.act]
            lay_arr[i].param_off = lay.param_off
            lay_arr[i].trunk = lay.trunk
        self._lay_arr = lay_arr
        d = L.NetDesc()
        d.n_layers, d.layers, d.n_trunks = len(spec.layers), lay_arr, spec.n_trunks
        d.feat, d.action_dim, d.pi_off, d.v_off, d.n_params = spec.feat, spec.action_dim, spec.pi_off, spec.v_off, n
        d.xf = L.InputXform(*spec.input_xform)
        d.in_h, d.in_w, d.in_c = spec.layers[0].H, spec.layers[0].W, spec.layers[0].C      # (C: as the kernels read it, padded)
        d.action_type = L.ACTION_TYPE[getattr(spec, "action_type", "Categorical")]
        d.logstd_off = getattr(spec, "logstd_off", 0)
        self._desc = d
        h = ctypes.c_void_p()
        L.check(self.lib.xt_net_create(ctypes.byref(d), self.max_batch, ctypes.byref(h)), "xt_net_create")
        self.handle = h
        ws_bytes = self.lib.xt_net_workspace_bytes(h)
        self.workspace = torch.empty(ws_bytes // 4, dtype=torch.float32, device=self.device)
        L.check(self.lib.xt_net_bind(h, L.ptr(self.params), L.ptr(self.grads), L.ptr(self.adam_m), L.ptr(self.adam_v),
                                     L.ptr(self.adam_state), L.ptr(self.workspace), ws_bytes), "xt_net_bind")
        L.check(self.lib.xt_adam_state_init(L.ptr(self.adam_state), L.stream_ptr()), "xt_adam_state_init")
        if init == "glorot":
            self.init_weights(seed)

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.xt_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    inference_only = False

    def init_weights(self, seed=None, baseline_norm_std=None):
        self.set_weights(initial_weights(self.spec, seed, baseline_norm_std))

    def _flat_to_host(self, dev_flat, tag):
        """One D2H of a flat device buffer into a persistent PINNED host buffer (the weight publish
        after every PPO update is on the critical path; a pageable .cpu() bounces through a driver staging copy)."""
        pin = getattr(self, "_pin", None)
        if pin is None:
            pin = self._pin = {}
        if tag not in pin:
            pin[tag] = torch.empty(dev_flat.shape, dtype=dev_flat.dtype, pin_memory=True)
        pin[tag].copy_(dev_flat.detach(), non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return pin[tag].numpy()

    # ---- weight publish: the D2H of the new parameters is enqueued by the update itself
    SNAP_SLOTS = 4

    def touch(self):
        """The parameters are about to change (or have been assigned): a snapshot taken earlier is stale."""
        self._version = getattr(self, "_version", 0) + 1

    def snapshot_weights_async(self):
        """Enqueue the D2H of the flat parameter buffer into the next of ``SNAP_SLOTS`` pinned host blocks on a side
        stream, ordered after everything already enqueued on the compute stream (the update whose result it
        publishes).  Returns immediately: the copy runs under whatever the host does next (the loss read-back, the next
        rollout's ingest).  ``get_weights`` picks the block up once its event has fired."""
        ring = getattr(self, "_wring", None)
        if ring is not None and not getattr(ring, "pinned", False):
            ring = self._wring = None       # the ring was closed (or un-pinned) behind our back: private snapshots again
        if ring is not None and getattr(ring, "async_commit", False):
            # asynchronous commit (WeightsRing.start_committer): every begun copy IS a publish; the ring's helper thread makes
            # it visible when it lands (begin blocks only when slots - 1 copies are still in flight)
            ring.begin_flat_publish(self, getattr(self, "_wring_ctr", None))
            self._wring_version = getattr(self, "_version", 0)
            return
        if ring is not None:
            # a page-locked WeightsRing is attached: the copy goes straight into its next slot (no pinned bounce block).
            # Begun publishes beyond the allowed lag belong to updates whose weights were never handed out: reuse their slot
            while len(ring._pending) > getattr(self, "_wring_lag", 0):
                ring.retarget_flat_publish()
            ring.begin_flat_publish(self, getattr(self, "_wring_ctr", None))
            self._wring_version = getattr(self, "_version", 0)
            return
        snap = getattr(self, "_snap", None)
        if snap is None:
            snap = self._snap = dict(stream=torch.cuda.Stream(device=self.device), slot=-1, version=-1, events=[],
                                     host=[torch.empty(self.params.shape, dtype=torch.float32, pin_memory=True)
                                           for _ in range(self.SNAP_SLOTS)])
            snap["events"] = [torch.cuda.Event() for _ in range(self.SNAP_SLOTS)]
            snap["ready"] = [torch.cuda.Event() for _ in range(self.SNAP_SLOTS)]
        i = (snap["slot"] + 1) % self.SNAP_SLOTS
        ready = snap["ready"][i]
        ready.record(L.current_stream(self.device))
        snap["stream"].wait_event(ready)
        L.memcpy_async(snap["host"][i].data_ptr(), self.params.data_ptr(), self.params.numel() * 4, L.D2H, snap["stream"])
        snap["events"][i].record(snap["stream"])
        snap["slot"], snap["version"] = i, getattr(self, "_version", 0)

    def get_weights(self, copy=True):
        """dict TF-variable-name -> ndarray (TFVariables.get_weights, xt/model/tf_utils.py:99-102).  The flat
        parameter buffer IS the packed form: ONE pinned D2H -- already in flight when the last update enqueued it
        (``snapshot_weights_async``), else issued now.  By default the arrays are private copies, as the reference hands
        them out (a caller may keep them: best-checkpoint retention, PBT).  ``copy=False`` returns per-variable VIEWS into
        the pinned block, valid until ``SNAP_SLOTS - 1`` further snapshots have been taken: for consumers that
        serialise them at once (``transport.WeightsRing.publish``, ``save_model``)."""
        flat = self._snapshot_flat()
        if copy:
            # ONE private block per call (the native staging pool moves the 3.4 MB at ~100 GB/s instead of a dozen
            # single-threaded numpy copies: 0.24 -> ~0.1 ms per publish); the dict's arrays are views of that block, which
            # nobody else holds
            own = np.empty(flat.shape, np.float32)
            if not getattr(self, "_tuned_copy", False):
                from xingtian_amd.ingest import staging_report
                staging_report()
                self._tuned_copy = True
            L.check(self.lib.xt_stage_rows(ctypes.c_void_p(own.ctypes.data), ctypes.c_void_p(flat.ctypes.data), own.nbytes,
                                           None, 0, 0, -1, None), "xt_stage_rows")
            flat = own
        out = OrderedDict()
        for name in self.spec.names:
            out[name] = self.spec.var_view(flat, name)
        return out

    def _snapshot_flat(self):
        """the flat float32 parameter buffer in pinned host memory, as of everything enqueued so far"""
        snap = getattr(self, "_snap", None)
        if snap is None or snap["version"] != getattr(self, "_version", 0):
            ring, self._wring = getattr(self, "_wring", None), None       # (a private snapshot, not a ring publish)
            try:
                self.snapshot_weights_async()
            finally:
                self._wring = ring
            snap = self._snap
        snap["version"] = -1            # a pre-enqueued snapshot serves ONE publish; later calls copy again
        snap["events"][snap["slot"]].synchronize()
        return snap["host"][snap["slot"]].numpy()

    def publish_weights(self, ring, ctr_info=None, lag=0):
        """The learner's weight hand-over (xt/framework/learner.py:361-363) without an intermediate dict: the packed
        parameter block goes pinned block -> ring slot with one copy per variable, or -- when the ring is page-locked
        (``WeightsRing.pin``) -- by ONE D2H straight from HBM into the slot (no host copy at all).  Returns the
        publish's sequence number."""
        if getattr(ring, "pinned", False):
            attached = getattr(self, "_wring", None) is ring
            if getattr(ring, "async_commit", False):
                # the update enqueued the D2H into the ring slot (or it is enqueued now); the ring's committer thread makes it
                # visible when it lands -- this thread goes on (content = THIS update's weights, no lag)
                if not attached:
                    raise ValueError("publish_weights on a ring with an asynchronous committer needs attach_weights_ring(ring)")
                if getattr(self, "_wring_version", -1) != getattr(self, "_version", 0) or ctr_info:
                    ring.begin_flat_publish(self, ctr_info)
                    self._wring_version = getattr(self, "_version", 0)
                return ring._last_begun
            if lag > 0:
                # ``lag = 1`` (asynchronous algorithms, flagged deviation): hand out the weights whose copy was begun at the
                # PREVIOUS publish and never wait for the one just enqueued -- the next rollout message is ingested while
                # the GPU still runs this update.  The copy of the current update stays begun.  With
                # train_per_checkpoint = 1 every train begins a copy (eager_snapshot), so the lag is ONE TRAIN; with
                # train_per_checkpoint = k > 1 copies are only begun here, so the lag is one publish INTERVAL = k trains
                # (pong_impala_speedup.yaml: 3) -- bench.py labels it so.
                if not attached or ring.slots < 3:
                    raise ValueError("publish_weights(lag=1) needs attach_weights_ring(ring) and a ring of >= 3 slots")
                self._wring_lag = 1
                if not ring._pending or getattr(self, "_wring_version", -1) != getattr(self, "_version", 0):
                    ring.begin_flat_publish(self, ctr_info)
                    self._wring_version = getattr(self, "_version", 0)
                if len(ring._pending) > 1 or ring.latest() == 0:
                    return ring.commit_flat_publish()
                return ring.latest()
            self._wring_lag = 0
            if attached and ring._pending and getattr(self, "_wring_version", -1) == getattr(self, "_version", 0) and not ctr_info:
                while len(ring._pending) > 1:
                    ring.commit_flat_publish()
                return ring.commit_flat_publish()       # the update itself enqueued the copy (snapshot_weights_async)
            while ring._pending:
                ring.retarget_flat_publish()
            return ring.publish_flat_from_device(self, ctr_info)
        return ring.publish(self.get_weights(copy=False), ctr_info)

    def attach_weights_ring(self, ring):
        """From now on the weight snapshot that every update enqueues (``snapshot_weights_async``) lands directly in
        ``ring``'s next slot (``ring`` must be page-locked, ``WeightsRing.pin``): ``publish_weights(ring)`` then only
        waits for that copy and commits it.  ``None`` detaches."""
        if ring is not None and not getattr(ring, "pinned", False):
            raise ValueError("attach_weights_ring: the ring must be page-locked (WeightsRing.pin())")
        self._wring = ring

    def set_weights(self, weights):
        """Assign by name; unknown names are ignored, KeyError if nothing matches
        (TFVariables.set_weights, xt/model/tf_utils.py:104-128)."""
        hit = [k for k in weights.keys() if k in self.spec.names]
        if not hit:
            raise KeyError("NO node's weights could assign in self.graph {} vs {}".format(
                list(self.spec.names.keys()), list(weights.keys())))
        flat = self.params.detach().cpu().numpy().copy()
        for name in hit:
            off, shape = self.spec.names[name]
            val = np.asarray(weights[name], np.float32)
            if tuple(val.shape) != tuple(shape):
                raise KeyError("update {} encounter error: shape {} vs {}".format(name, val.shape, shape))
            self.spec.var_view(flat, name)[...] = val
        self.params.copy_(torch.from_numpy(flat))
        self.touch()

    # ------------------------------------------------------------------ optimizer state
    # The reference never checkpoints its AdamOptimizer slots (TFVariables only walks the trainable variables,
    # xt/model/tf_utils.py:60-97), so a restored run restarts Adam from zero.  For a true resume the slots are
    # exported under the names TF1 itself gives them ("<var>/Adam" = m, "<var>/Adam_1" = v, "beta1_power",
    # "beta2_power"), next to the weights: TFVariables.set_weights ignores names it does not know
    # (tf_utils.py:106-109), so the same .npz still loads in the reference.
    OPT_M, OPT_V, OPT_B1, OPT_B2, OPT_STEP = "/Adam", "/Adam_1", "beta1_power", "beta2_power", "adam_step"
    opt_kind = "adam"        # "rmsprop": the two slot buffers hold mg / ms; saved as "<var>/RMSProp_1" / "<var>/RMSProp"

    def _slot_suffixes(self):
        return ("/RMSProp_1", "/RMSProp") if self.opt_kind == "rmsprop" else (self.OPT_M, self.OPT_V)

    def get_optimizer_state(self):
        m = self.adam_m.detach().cpu().numpy()
        v = self.adam_v.detach().cpu().numpy()
        st = self.adam_state.detach().cpu().numpy()
        out = OrderedDict()
        for name in self.spec.names:
            sm, sv = self._slot_suffixes()
            out[name + sm] = self.spec.var_view(m, name).copy()
            out[name + sv] = self.spec.var_view(v, name).copy()
        out[self.OPT_B1] = np.float32(st[0])
        out[self.OPT_B2] = np.float32(st[1])
        out[self.OPT_STEP] = np.int64(round(float(st[5])))
        return out

    def set_optimizer_state(self, state):
        """Restore Adam slots saved by get_optimizer_state; returns False (state untouched) unless EVERY slot
        and both beta powers are present with the right shapes."""
        need = [n + sfx for n in self.spec.names for sfx in self._slot_suffixes()] + [self.OPT_B1, self.OPT_B2]
        if any(k not in state for k in need):
            return False
        m = np.zeros(self.spec.n_flat, np.float32)
        v = np.zeros(self.spec.n_flat, np.float32)
        for name, (off, shape) in self.spec.names.items():
            for sfx, dst in zip(self._slot_suffixes(), (m, v)):
                val = np.asarray(state[name + sfx], np.float32)
                if tuple(val.shape) != tuple(shape):
                    raise KeyError("optimizer slot {} shape {} vs {}".format(name + sfx, val.shape, shape))
                self.spec.var_view(dst, name)[...] = val
        st = self.adam_state.detach().cpu().numpy().copy()
        st[0], st[1] = np.float32(state[self.OPT_B1]), np.float32(state[self.OPT_B2])
        st[5] = np.float32(state[self.OPT_STEP]) if self.OPT_STEP in state else 0.0
        self.adam_m.copy_(torch.from_numpy(m))
        self.adam_v.copy_(torch.from_numpy(v))
        self.adam_state.copy_(torch.from_numpy(st))
        return True

    def check_device_errors(self):
        """Raise if a kernel flagged an error in the optimiser state block (state[6], include/xt_mi355x.h): today only
        the fused update tail (``xt_tuning.tail_fused``, off by default) whose grid barrier timed out and skipped an update."""
        flag = float(self.adam_state[6].item())
        if flag != 0.0:
            raise RuntimeError("xingtian_amd: device-side error word {} (1: the fused update tail's grid barrier timed "
                               "out, the update was skipped)".format(flag))

    def reset_optimizer(self):
        self.adam_m.zero_()
        self.adam_v.fill_(1.0) if self.opt_kind == "rmsprop" else self.adam_v.zero_()
        L.check(self.lib.xt_adam_state_init(L.ptr(self.adam_state), L.stream_ptr()), "xt_adam_state_init")

    # ------------------------------------------------------------------ compute
    def to_device_obs(self, obs):
        t = torch.as_tensor(np.ascontiguousarray(obs)) if not torch.is_tensor(obs) else obs
        want = torch.uint8 if self.spec.input_xform[0] else torch.float32
        t = t.to(device=self.device, dtype=want)
        lay0 = self.spec.layers[0]
        if t.dim() == 2 and lay0.H == lay0.W == 1 and t.shape[1] < lay0.C:      # zero-pad odd feature counts (netspec._mlp)
            t = torch.nn.functional.pad(t, (0, lay0.C - t.shape[1]))
        elif t.dim() == 4 and t.shape[3] < lay0.C:                                 # image channels -> multiple of 4
            t = self.pad_obs_channels(t.contiguous())
        return t.contiguous()

    def obs_fill_byte(self):
        """the uint8 value the input transform maps to 0 (what padded channel planes are filled with)"""
        is_u8, mean, _ = self.spec.input_xform
        if not is_u8 or mean == 0.0:
            return 0
        if float(mean) != int(mean) or not 0 <= int(mean) <= 255:
            raise ValueError("channel padding of uint8 observations needs an integral state_mean in [0, 255], got {}".format(mean))
        return int(mean)

    def pad_obs_channels(self, t, out=None):
        """[n, H, W, c] device tensor -> [n, H, W, pad4(c)] with neutral extra planes (C ABI xt_pad_channels)."""
        c_dst = self.spec.layers[0].C
        n, h, w, c = t.shape
        if out is None:
            out = torch.empty((n, h, w, c_dst), dtype=t.dtype, device=t.device)
        L.check(self.lib.xt_pad_channels(L.ptr(t), L.ptr(out), n * h * w, c, c_dst, t.element_size(), self.obs_fill_byte(),
                                         L.stream_ptr()), "xt_pad_channels")
        return out

    def forward(self, obs):
        """obs [B, ...] (host or device) -> (logits [B,A], value [B]) device tensors."""
        x = self.to_device_obs(obs)
        b = x.shape[0]
        a = self.spec.action_dim
        logits = torch.empty((b, a), dtype=torch.float32, device=self.device)
        value = torch.empty((b,), dtype=torch.float32, device=self.device)
        for s in range(0, b, self.max_batch):
            e = min(b, s + self.max_batch)
            L.check(self.lib.xt_net_forward(self.handle, L.ptr(x[s:e]), None, e - s, L.ptr(logits[s:e]),
                                            L.ptr(value[s:e]), L.stream_ptr()), "xt_net_forward")
        return logits, value

    def set_dp(self, rank, world, loss_scale=1.0):
        """C ABI ``xt_net_set_dp``: switch the data-parallel tail on (``world`` >= 1: rows + loss shares of every rank travel
        behind the gradient in the same exchange, the optimiser adds the GLOBAL loss to ``loss_acc``) or off (``world`` = 0)"""
        L.check(self.lib.xt_net_set_dp(self.handle, int(rank), int(world), float(loss_scale)), "xt_net_set_dp")

    def read_loss(self, acc=None, wait=True):
        """[sum of step losses, number of steps, data-parallel error bits, -] of the train(s) enqueued so far: ONE 16-byte D2H
        into a pinned block + an event wait instead of a pageable ``.cpu()`` (a staging copy, an allocation and a device-wide
        synchronisation: ~30 us of a 128-frame IMPALA train).  Two blocks alternate; ``wait=False`` returns the block of the
        PREVIOUS call instead (ASYNC_LOSS: its copy landed long ago; the very first call still waits for its own).  Raises if
        the train left error bits (the update was skipped on this rank: include/xt_mi355x.h ``xt_net_set_direct``)."""
        acc = self.loss_acc if acc is None else acc
        rb = getattr(self, "_loss_rb", None)
        if rb is None:
            pin = torch.zeros((2, 4), dtype=torch.float32, pin_memory=True)
            rb = self._loss_rb = dict(pin=pin, np=pin.numpy(), ev=[torch.cuda.Event(), torch.cuda.Event()], slot=0, n=0)
        i = rb["slot"]
        cur = L.current_stream(self.device)
        L.memcpy_async(rb["pin"][i].data_ptr(), acc.data_ptr(), 16, L.D2H, cur)
        rb["ev"][i].record(cur)
        rb["slot"] = i ^ 1
        j = i if (wait or rb["n"] == 0) else i ^ 1
        rb["n"] += 1
        gate = getattr(self, "idle_gate", None)        # (transport.Prefetcher: its staging thread works while we wait)
        if gate is not None:
            gate.set()
            rb["ev"][j].synchronize()
            gate.clear()
        else:
            rb["ev"][j].synchronize()
        a = rb["np"][j]
        if a[2] != 0.0:
            raise RuntimeError("xingtian_amd: the data-parallel update failed on this rank -- {} (error bits {}); the "
                               "optimiser skipped the update, parameters are those of the last good step".format(
                                   L.dp_error_text(a[2]), int(a[2])))
        return a

    def make_ppo_cfg(self, cfg, grad_scale=1.0, global_batch=0, shard_rank=0, shard_world=0):
        c = L.PpoCfg()
        c.lr, c.beta1, c.beta2, c.eps = cfg["LR"], 0.9, 0.999, 1e-8
        c.clip_ratio, c.ent_coef, c.vf_clip = cfg["LOSS_CLIPPING"], cfg["ENTROPY_LOSS"], cfg["VF_CLIP"]
        c.critic_coef, c.max_grad_norm = cfg["CRITIC_LOSS_COEF"], cfg["MAX_GRAD_NORM"]
        c.batch_size, c.num_sgd_iter = int(cfg["BATCH_SIZE"]), int(cfg["NUM_SGD_ITER"])
        c.grad_scale, c.global_batch = grad_scale, int(global_batch)
        # strict data parallelism inside xt_net_ppo_train: this rank's shard of every global minibatch (ABI >= 9)
        c.shard_rank, c.shard_world = int(shard_rank), int(shard_world)
        return c

    def ppo_step(self, c, obs, idx, action, old_logp, adv, old_v, target_v, apply=True):
        """One SGD step on rows idx (int32 device tensor or None) of device-resident data."""
        b = int(idx.numel()) if idx is not None else int(obs.shape[0])
        self.touch()
        L.check(self.lib.xt_net_ppo_step(self.handle, ctypes.byref(c), L.ptr(obs), L.ptr(idx), b, L.ptr(action),
                                         L.ptr(old_logp), L.ptr(adv), L.ptr(old_v), L.ptr(target_v),
                                         1 if apply else 0, L.ptr(self.loss_out), None, L.stream_ptr()),
                "xt_net_ppo_step")
        return self.loss_out

    def ppo_train(self, c, obs, perm, action, old_logp, adv, old_v, target_v, use_graph=False):
        n = int(obs.shape[0])
        self.touch()
        L.check(self.lib.xt_net_ppo_train(self.handle, ctypes.byref(c), L.ptr(obs), n, L.ptr(perm), L.ptr(action),
                                          L.ptr(old_logp), L.ptr(adv), L.ptr(old_v), L.ptr(target_v),
                                          L.ptr(self.loss_acc), 1 if use_graph else 0, L.stream_ptr()),
                "xt_net_ppo_train")
        return self.loss_acc

    def make_impala_cfg(self, lr, grad_norm_clip, sample_batch_step, gamma=0.99, grad_scale=1.0, opt_type="adam",
                        shard_rank=0, shard_world=0):
        c = L.ImpalaCfg()
        c.shard_rank, c.shard_world = int(shard_rank), int(shard_world)
        c.lr, c.beta1, c.beta2, c.eps = lr, 0.9, 0.999, 1e-8
        c.grad_norm_clip, c.gamma, c.sample_batch_step, c.grad_scale = grad_norm_clip, gamma, int(sample_batch_step), grad_scale
        c.opt_type = L.OPT_TYPE[opt_type]
        c.rms_decay, c.rms_eps = 0.99, 0.1          # RMSPropOptimizer(LR, decay=0.99, epsilon=0.1, centered=True)
        if opt_type != self.opt_kind:
            raise ValueError("call set_optimizer({!r}) before building its config".format(opt_type))
        return c

    def set_optimizer(self, kind):
        """'adam' (default) or 'rmsprop': the latter reuses the two slot buffers as mean gradient / mean square and
        initialises them as TF does (rms = ones, mg = zeros)."""
        if kind not in L.OPT_TYPE:
            raise KeyError("invalid opt_type: {}".format(kind))
        self.opt_kind = kind
        self.reset_optimizer()

    def impala_step(self, c, obs, bp_logits, action, done, reward, apply=True, loss_acc=None):
        n = int(obs.shape[0])
        self.touch()
        L.check(self.lib.xt_net_impala_step(self.handle, ctypes.byref(c), L.ptr(obs), n, L.ptr(bp_logits),
                                            L.ptr(action), L.ptr(done), L.ptr(reward), 1 if apply else 0,
                                            L.ptr(self.loss_out), L.ptr(loss_acc), L.stream_ptr()),
                "xt_net_impala_step")
        return self.loss_out

    def impala_train(self, c, obs, batch_size, bp_logits, action, done, reward, lr_steps=None, use_graph=False):
        """IMPALAOpt.train in one call (C ABI xt_net_impala_train): sequential BATCH_SIZE chunks of the resident
        rollout; ``lr_steps`` (float32 device tensor, one step size per chunk) or None.  Returns the device tensor
        [sum of chunk losses, number of chunks]."""
        n = int(obs.shape[0])
        self.touch()
        L.check(self.lib.xt_net_impala_train(self.handle, ctypes.byref(c), L.ptr(obs), n, int(batch_size),
                                             L.ptr(bp_logits), L.ptr(action), L.ptr(done), L.ptr(reward),
                                             L.ptr(lr_steps), L.ptr(self.loss_acc), 1 if use_graph else 0,
                                             L.stream_ptr()), "xt_net_impala_train")
        return self.loss_acc

    def impala_train_io(self, c, obs, batch_size, bp_logits, action, done, reward, lr_steps=None, use_graph=False,
                        wait_event=None, consumed_event=None, publish=None, wait_loss=True, tail_in_graph=True, defer=False,
                        wait_dma_ticket=0):
        """``impala_train`` + the runtime calls of the learner loop around it in ONE C call (``xt_net_impala_train_io``): the
        compute stream waits for ``wait_event`` (the rollout's copies), ``consumed_event`` is recorded behind the train,
        ``loss_acc`` is copied into the next pinned read-back block and -- ``wait_loss`` -- awaited with the GIL released,
        ``publish`` = (host address of a pinned weights-ring slot, raw event handle) receives the new parameters by a D2H in
        stream order behind the loss copy (``WeightsRing.publish_reserve``).  ``tail_in_graph`` (with ``wait_loss``): both
        copies are the train's own last kernels, inside its replayed hipGraph, and the loss is awaited by polling a
        page-locked word (``xt_train_io.tail_in_graph``); ``defer`` (with both): return right behind the launch -- the caller
        does its loss-independent book-keeping under the device's train and fetches the loss with ``impala_wait_loss()``.
        Returns the pinned [sum, count, error bits, -] block of THIS train (``wait_loss``) or of the previous one (None when
        deferred); raises on error bits."""
        n = int(obs.shape[0])
        self._version = getattr(self, "_version", 0) + 1          # touch()
        rb = getattr(self, "_loss_rb", None)
        if rb is None:
            pin = torch.zeros((2, 4), dtype=torch.float32, pin_memory=True)
            rb = self._loss_rb = dict(pin=pin, np=pin.numpy(), ev=[torch.cuda.Event(), torch.cuda.Event()], slot=0, n=0)
        if "raw" not in rb:
            cur = L.current_stream(self.device)
            for ev in rb["ev"]:
                ev.record(cur)                 # (a torch event gets its handle at its first record)
            rb["raw"] = [ev.cuda_event for ev in rb["ev"]]
            rb["ptr"] = [rb["pin"][k].data_ptr() for k in range(2)]
            rb["io"] = L.TrainIO()             # one descriptor, rewritten per train
            rb["io_ref"] = ctypes.byref(rb["io"])
            rb["fn"] = self.lib.xt_net_impala_train_io
            rb["dev"] = self.device.index if self.device.index is not None else torch.cuda.current_device()
            rb["acc"] = self.loss_acc.data_ptr()
        i = rb["slot"]
        io = rb["io"]
        io.wait_event = wait_event.cuda_event if wait_event is not None else None
        io.wait_dma_ticket = int(wait_dma_ticket)      # (xt_dma_h2d_async copies of the rollout: awaited before the launch)
        io.consumed_event = consumed_event.cuda_event if consumed_event is not None else None
        if publish is not None:
            io.publish_dst, io.publish_event = publish[0], publish[1]
        else:
            io.publish_dst, io.publish_event = None, None
        sync = bool(wait_loss or rb["n"] == 0)
        tail = bool(tail_in_graph and sync)
        deferred = bool(defer and tail)
        io.loss_host, io.loss_event, io.wait_loss = rb["ptr"][i], rb["raw"][i], (2 if deferred else 1) if sync else 0
        io.tail_in_graph = (2 if int(tail_in_graph) == 2 else 1) if tail else 0      # (2: snapshot + side-stream D2H)
        ptr = L.ptr
        # the staging thread may work while this thread is inside C with the GIL released -- in the deferred form only from
        # impala_wait_loss() on: the launch is ~15 us and the book-keeping behind it needs the GIL to itself (two Python
        # threads hand it back and forth at every runtime call: measured, the 16 us C call took 56-67 us with the gate open)
        gate = getattr(self, "idle_gate", None)
        if deferred and not getattr(self, "gate_at_launch", True):
            gate = None
        if gate is not None:
            gate.set()
        try:
            rc = rb["fn"](self.handle, ctypes.byref(c), ptr(obs), n, int(batch_size), ptr(bp_logits), ptr(action), ptr(done),
                          ptr(reward), ptr(lr_steps), rb["acc"], 1 if use_graph else 0, rb["io_ref"],
                          torch._C._cuda_getCurrentRawStream(rb["dev"]))
            if rc:
                L.check(rc, "xt_net_impala_train_io")
        finally:
            if gate is not None and not deferred:
                gate.clear()
        rb["slot"] = i ^ 1
        rb["n"] += 1
        if deferred:
            rb["deferred"] = i
            return None
        j = i if sync else i ^ 1
        if j != i:
            rb["ev"][j].synchronize()
        return self._loss_block(rb["np"][j])

    def io_publish_done(self):
        """completion handle (``synchronize()`` / ``query()``, the two calls a ``transport.WeightsRing`` makes on a publish's
        event) of the parameter copy of the most recent ``tail_in_graph`` train: the copy kernel's last workgroup writes the
        train's sequence number into the mailbox (C ABI ``xt_net_io_seq`` / ``xt_net_io_publish_wait``) -- no event record
        behind the replayed graph"""
        return _MailboxDone(self, int(self.lib.xt_net_io_seq(self.handle)))

    def impala_wait_loss(self):
        """second half of ``impala_train_io(..., defer=True)``: wait for that train's loss (C ABI ``xt_net_io_wait``, GIL
        released) -> its pinned [sum, count, error bits, -] block"""
        rb = self._loss_rb
        i = rb.pop("deferred")
        # an INLINE prefetcher (transport.Prefetcher(inline=True)): this thread stages the next rollout message(s) itself while
        # the device trains -- one interpreter thread, nobody to fight for the lock -- and looks at the mailbox in between
        hook = getattr(self, "idle_hook", None)
        if hook is not None:
            ready = self.lib.xt_net_io_loss_ready
            while not ready(self.handle):
                if not hook():
                    break               # (nothing to stage: wait in C below)
        gate = getattr(self, "idle_gate", None)
        if gate is not None:
            gate.set()
        try:
            rc = self.lib.xt_net_io_wait(self.handle, rb["ptr"][i], torch._C._cuda_getCurrentRawStream(rb["dev"]))
            if rc:
                L.check(rc, "xt_net_io_wait")
        finally:
            if gate is not None:
                gate.clear()
        return self._loss_block(rb["np"][i])

    @staticmethod
    def _loss_block(a):
        if a[2] != 0.0:
            raise RuntimeError("xingtian_amd: the data-parallel update failed on this rank -- {} (error bits {}); the "
                               "optimiser skipped the update, parameters are those of the last good step".format(
                                   L.dp_error_text(a[2]), int(a[2])))
        return a

    def io_times(self, reset=True):
        """host time per call (us) of the phases of ``impala_train_io`` -- before the launch, the launch, the runtime calls
        behind it, the wait for the loss (C ABI ``xt_net_io_times``; diagnostic)"""
        us, calls = (ctypes.c_double * 4)(), ctypes.c_int64()
        L.check(self.lib.xt_net_io_times(self.handle, us, ctypes.byref(calls), 1 if reset else 0), "xt_net_io_times")
        n = max(int(calls.value), 1)
        return dict(calls=int(calls.value), pre_us=us[0] / n, launch_us=us[1] / n, post_us=us[2] / n, wait_us=us[3] / n)

    def keras_impala_step(self, obs, idx, adv, onehot, target_v, ent_coef, loss_acc=None):
        """One ``model.fit`` minibatch of the non-opt IMPALA models (C ABI xt_net_keras_impala_step): forward, Keras
        impala_loss + 0.5 mse, backward; the gradient stays in ``self.grads`` for ``adam_keras``.  Returns the device
        tensor [loss, policy loss, mse]."""
        b = int(idx.numel()) if idx is not None else int(obs.shape[0])
        L.check(self.lib.xt_net_keras_impala_step(self.handle, L.ptr(obs), L.ptr(idx), b, L.ptr(adv), L.ptr(onehot),
                                                  L.ptr(target_v), float(ent_coef), L.ptr(self.loss_out),
                                                  L.ptr(loss_acc) if loss_acc is not None else None, L.stream_ptr()),
                "xt_net_keras_impala_step")
        return self.loss_out

    def adam_keras(self, lr, iterations, clipnorm=0.0, decay=0.0, beta1=0.9, beta2=0.999, eps=1e-7):
        """tf.keras Adam on the flat buffers: per-TENSOR clip_by_norm (kernel and bias are separate tensors), time-decayed
        learning rate; ``iterations`` = number of updates applied so far."""
        if not hasattr(self, "_keras_segs"):
            offs, sizes = [], []
            for name in self.spec.names:      # (a channel-padded kernel: its storage block; the padding is zero)
                off, size = self.spec.var_extent(name)
                offs.append(int(off))
                sizes.append(int(size))
            dev = self.device
            self._keras_segs = (torch.tensor(offs, dtype=torch.int64), torch.tensor(sizes, dtype=torch.int64), len(offs))
            self._keras_scratch = torch.zeros((16 * len(offs),), dtype=torch.float32, device=dev)
        offs, sizes, n = self._keras_segs
        t = iterations + 1
        lr_t = np.float32(lr) / (np.float32(1.0) + np.float32(decay) * np.float32(iterations))
        lr_t = lr_t * np.sqrt(np.float32(1.0) - np.float32(beta2) ** t) / (np.float32(1.0) - np.float32(beta1) ** t)
        self.touch()
        L.check(self.lib.xt_adam_keras(L.ptr(self.params), L.ptr(self.grads), L.ptr(self.adam_m), L.ptr(self.adam_v), n,
                                       offs.data_ptr(), sizes.data_ptr(), float(clipnorm), float(lr_t), beta1, beta2,
                                       eps, L.ptr(self._keras_scratch), L.stream_ptr()), "xt_adam_keras")

    def apply(self, lr, clip_norm, grad_scale=1.0):
        self.touch()
        L.check(self.lib.xt_net_apply(self.handle, lr, 0.9, 0.999, 1e-8, clip_norm, grad_scale, L.stream_ptr()),
                "xt_net_apply")

    def layer_buffers(self, layer, b):
        """(activation, d-pre-activation) views of the workspace for the first ``b`` samples of ``layer`` (tests)."""
        off = (ctypes.c_int64 * 4)()
        L.check(self.lib.xt_net_layer_offsets(self.handle, layer, off), "xt_net_layer_offsets")
        lay = self.spec.layers[layer]
        n = b * lay.OH * lay.OW * lay.N
        return self.workspace[off[0]:off[0] + n], self.workspace[off[1]:off[1] + n]

    def time_layer(self, layer, which, obs, idx, b, reps=20):
        ms = ctypes.c_float()
        L.check(self.lib.xt_net_time_layer(self.handle, layer, which, L.ptr(obs), L.ptr(idx), b, reps,
                                           ctypes.byref(ms), L.stream_ptr()), "xt_net_time_layer")
        return ms.value

    def time_tail(self, lr=2.5e-4, clip_norm=5.0, reps=50):
        """average ms of the step's tail (gradient reduction, [exchange], clip + Adam) in the net's current mode, on the
        slabs of the most recent gradient step (C ABI xt_net_time_tail)"""
        ms = ctypes.c_float()
        self.touch()
        L.check(self.lib.xt_net_time_tail(self.handle, float(lr), float(clip_norm), int(reps), ctypes.byref(ms),
                                          L.stream_ptr()), "xt_net_time_tail")
        return ms.value

    def grads_dict(self):
        flat = self.grads.detach().cpu().numpy()
        out = OrderedDict()
        for name in self.spec.names:
            out[name] = self.spec.var_view(flat, name).copy()
        return out
