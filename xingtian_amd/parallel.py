"""Data-parallel learner plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the multi-process tests, which also accepts device tensors).

The learner update shards naturally: samples of a minibatch are independent through
forward/backward and GAE / v-trace recur only along time, so every rank owns whole trajectories and a
slice of every minibatch; the ONLY exchange is one all-reduce (SUM) of the flat fp32 gradient buffer per SGD
step, after which every rank applies the identical clip + Adam update (replicas stay bit-identical because
the reduced buffer is identical on all ranks).

Two PPO modes (``dp_ppo_update``):

* ``strict`` -- what north_star / SURVEY 8(e) specify: the reference's GLOBAL minibatch of BATCH_SIZE rows
  (xt/model/ppo/ppo.py:119-124) is split into ``world`` equal shards (320 -> 40 rows per GPU at 8 GPUs), every rank
  walks the SAME epoch permutations, the loss means (xt/model/ppo/__init__.py:13,24) run over the global minibatch
  (``cfg.global_batch``), so the summed gradient IS the single-GPU gradient (up to fp32 summation order) and
  ``grad_scale`` = 1.
* ``weak`` -- every rank owns its own env_num trajectories and a full BATCH_SIZE local minibatch (global minibatch
  BATCH_SIZE * world: a flagged deviation that keeps the per-GPU work constant); the local means are averaged:
  ``grad_scale`` = 1 / world.

IMPALA's loss is a SUM over (T-1) x B (impala_cnn_opt.py:299-318,351): ranks own whole trajectories of a chunk,
the gradients are summed, no scaling.

The reference has no working counterpart: its multi-process trainer (xt/framework/trainer.py:32-136) averages
gradients on the host in float64 through a RawArray and is dead code (no Algorithm implements get_grad).
"""
import numpy as np
import torch
import torch.distributed as dist


def world_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(n_items, rank, world):
    """Contiguous, balanced [begin, end) shard of n_items (trajectories / minibatch rows) for ``rank``."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def split_minibatch(perm_row, start, batch_size, rank, world):
    """Rows of the GLOBAL minibatch perm_row[start:start+batch_size] owned by ``rank`` (balanced contiguous shards;
    the global permutation is drawn once with a shared seed so that every rank partitions it identically).  Works on
    numpy arrays and on (device) tensors; a tensor slice is a view, i.e. an index indirection without a copy."""
    stop = min(start + batch_size, len(perm_row))
    b, e = shard_range(stop - start, rank, world)
    return perm_row[start + b:start + e]


def allreduce_sum_(flat_grad):
    """In-place SUM all-reduce of the flat gradient buffer (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def grad_scale(loss_reduction, world):
    """Factor applied to the summed gradient: 'mean' losses averaged over equal local minibatches (PPO, weak mode)
    -> 1/world; 'sum' losses (IMPALA) and means already taken over the global minibatch (PPO, strict mode) -> 1."""
    if loss_reduction == "mean":
        return 1.0 / world
    if loss_reduction in ("sum", "global_mean"):
        return 1.0
    raise ValueError(loss_reduction)


def broadcast_weights_(flat_params, src=0):
    """Make replicas identical at start-up (random init differs per process unless seeded identically)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params


def dp_ppo_step(net, cfg_struct, lr, max_grad_norm, obs, idx, action, old_logp, adv, old_v, target_v, world,
                scale=None):
    """One data-parallel PPO SGD step on a HipActorCritic: local fwd/bwd on this rank's rows ``idx`` -> all-reduce
    (SUM) of net.grads -> identical clip+Adam on every rank.  ``scale`` multiplies the summed gradient
    (default: 1/world, the weak mode; strict mode passes 1 with ``cfg_struct.global_batch`` = the global rows)."""
    net.ppo_step(cfg_struct, obs, idx, action, old_logp, adv, old_v, target_v, apply=False)
    allreduce_sum_(net.grads)
    net.apply(lr, max_grad_norm, grad_scale=grad_scale("mean", world) if scale is None else scale)
    return net.loss_out


def dp_ppo_update(net, cfg, obs, perm, action, old_logp, adv, old_v, target_v, rank, world, mode="strict"):
    """One whole ``Model.train`` (xt/model/ppo/ppo.py:111-132) data parallel over ``world`` ranks.

    ``cfg``: the reference's model_config keys (LR, BATCH_SIZE, NUM_SGD_ITER, ...); ``perm`` int32 DEVICE tensor
    [NUM_SGD_ITER, n] with the epoch permutations (identical on every rank in strict mode); the rollout tensors are
    device resident.  strict: every rank holds the same n-row rollout and processes its shard of every global
    minibatch; weak: every rank holds its own rollout and processes full local minibatches.
    Returns the number of SGD steps."""
    bsz, epochs = int(cfg["BATCH_SIZE"]), int(cfg["NUM_SGD_ITER"])
    n = int(perm.shape[1])
    steps = 0
    structs = {}
    for ep in range(epochs):
        for start in range(0, n, bsz):
            rows = min(bsz, n - start)
            if mode == "strict":
                if rows < world:        # the same test on EVERY rank, before any collective: nobody is left waiting
                    raise ValueError("strict sharding: a minibatch of {} rows cannot be split over {} ranks".format(
                        rows, world))
                idx = split_minibatch(perm[ep], start, bsz, rank, world)
                key, scale = rows, grad_scale("global_mean", world)
                if key not in structs:
                    structs[key] = net.make_ppo_cfg(cfg, grad_scale=scale, global_batch=rows)
            elif mode == "weak":
                idx = perm[ep, start:start + rows]
                key, scale = 0, grad_scale("mean", world)
                if key not in structs:
                    structs[key] = net.make_ppo_cfg(cfg, grad_scale=scale, global_batch=0)
            else:
                raise ValueError(mode)
            dp_ppo_step(net, structs[key], cfg["LR"], cfg["MAX_GRAD_NORM"], obs, idx, action, old_logp, adv, old_v,
                        target_v, world, scale=scale)
            steps += 1
    return steps


def dp_impala_step(net, cfg_struct, lr, grad_norm_clip, obs, bp_logits, action, done, reward, n_traj, t_len, rank,
                   world, exchange=None, lr_steps=None):
    """One data-parallel ImpalaCnnOpt step: the chunk's ``n_traj`` trajectories (flat env-major rows b*T+t) are
    split into whole-trajectory shards, gradients of the sum-form loss are SUMMED, no scaling.

    Adam with the fixed step size goes step-wise (local gradient -> ``torch.distributed`` all-reduce -> clip + Adam).
    ``opt_type: rmsprop``, an ``lr_schedule`` step size (``lr_steps``: one-element float32 device tensor) or an explicit
    ``exchange`` (``RcclComm`` / ``TorchDistExchange``) go through the gradient-exchange hook of ``xt_net_impala_train``:
    the library applies the configured optimiser to the EXCHANGED gradient itself
    (xt/model/impala/impala_cnn_opt.py:198-217,234-249)."""
    from xingtian_amd import lib as L
    b, e = shard_range(n_traj, rank, world)
    hook = exchange is not None or lr_steps is not None or int(cfg_struct.opt_type) != L.OPT_TYPE["adam"]
    if hook:
        # one exchange object per net (a fresh ctypes trampoline per step cost ~0.2 ms of host set-up)
        ex = exchange
        if ex is None:
            ex = getattr(net, "_dp_exchange", None)
            if ex is None:
                ex = net._dp_exchange = TorchDistExchange(net)
        native = isinstance(ex, (RcclComm, DirectComm))
        ex.attach(net) if native else ex.attach()
        try:
            if e > b:
                sl = slice(b * t_len, e * t_len)
                net.impala_train(cfg_struct, obs[sl], (e - b) * t_len, bp_logits[sl], action[sl], done[sl], reward[sl],
                                 lr_steps=lr_steps, use_graph=False)
            else:
                # an empty shard (fewer trajectories than ranks) contributes a ZERO gradient, like the step-wise branch
                # below: the library does that itself when it shards (xt_impala_cfg.shard_world, ABI 10)
                import copy
                c2 = copy.copy(cfg_struct)
                c2.shard_rank, c2.shard_world = int(rank), int(world)
                n_all = n_traj * t_len
                net.impala_train(c2, obs[:n_all], n_all, bp_logits[:n_all], action[:n_all], done[:n_all], reward[:n_all],
                                 lr_steps=lr_steps, use_graph=False)
        finally:
            ex.detach(net) if native else ex.detach()
        return
    if e > b:
        sl = slice(b * t_len, e * t_len)
        net.impala_step(cfg_struct, obs[sl], bp_logits[sl], action[sl], done[sl], reward[sl], apply=False)
    else:
        net.grads.zero_()
    allreduce_sum_(net.grads)
    net.apply(lr, grad_norm_clip, grad_scale=grad_scale("sum", world))


class RcclComm(object):
    """A raw RCCL communicator for the gradient-exchange hook of ``xt_net_ppo_train`` (C ABI >= 4).

    ``torch.distributed`` owns its communicator and issues collectives on its own stream from Python, one call per
    SGD step.  The hook form instead enqueues ``ncclAllReduce`` on the stream the learner kernels run on, from inside
    ``xt_net_ppo_train`` -- so the 52 all-reduces of an update are captured into the update's hipGraph and replayed
    without any host involvement.  The communicator is created with ctypes on the ``librccl.so`` that torch ships
    (already loaded in the process); the 128-byte unique id travels through the existing ``torch.distributed`` group.

    Validated with a 1-rank communicator on one GPU (bit-identical to the step-wise path); it has not met a second rank
    yet (no multi-GPU box was available to the builders), so ``bench.py`` validates every hook variant against the
    step-wise path before it may carry a number.
    """
    NCCL_FLOAT32, NCCL_SUM = 7, 0

    def __init__(self, rank, world):
        import ctypes
        import os
        self._ct = ctypes
        self.rank, self.world = int(rank), int(world)
        self.lib = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

        class UniqueId(ctypes.Structure):
            _fields_ = [("internal", ctypes.c_byte * 128)]

        uid = UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        if self.world > 1:
            t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device="cuda")
            dist.broadcast(t, src=0)
            ctypes.memmove(ctypes.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        self.comm = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
        self.lib.ncclCommInitRank.restype = ctypes.c_int
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")
        self.lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p]
        self.lib.ncclAllReduce.restype = ctypes.c_int
        self.errors = []

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with ncclResult %d" % (what, rc))

    def all_reduce_(self, flat, stream_ptr):
        """Eager in-place SUM all-reduce of a float32 device tensor on the given stream (also warms RCCL up before
        the first capture: its lazy allocations must not happen under stream capture)."""
        self._check(self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), self.NCCL_FLOAT32,
                                           self.NCCL_SUM, self.comm, stream_ptr), "ncclAllReduce")

    def attach(self, net, overlap=False):
        """Install this communicator as the network's gradient exchange (C ABI ``xt_net_set_rccl``): the library calls
        ``ncclAllReduce`` itself through the function pointer resolved here -- no Python frame on the enqueue path.
        ``overlap``: two buckets per step, the last trunk layer + heads exchanged on the library's side stream while the
        conv backward runs (``XT_XCHG_OVERLAP``)."""
        from xingtian_amd import lib as L
        fn = self._ct.cast(self.lib.ncclAllReduce, self._ct.c_void_p)
        L.check(net.lib.xt_net_set_rccl(net.handle, self.comm, fn, L.XCHG_OVERLAP if overlap else 0), "xt_net_set_rccl")

    def status(self, net):
        """(calls, last non-zero ncclResult) of the exchange installed on ``net`` (C ABI ``xt_net_rccl_status``)."""
        from xingtian_amd import lib as L
        calls, err = self._ct.c_int32(0), self._ct.c_int32(0)
        L.check(net.lib.xt_net_rccl_status(net.handle, self._ct.byref(calls), self._ct.byref(err)), "xt_net_rccl_status")
        if err.value:
            self.errors.append(int(err.value))
        return int(calls.value), int(err.value)

    def count(self):
        """ranks RCCL itself reports for this communicator (ncclCommCount)"""
        n = self._ct.c_int(0)
        self.lib.ncclCommCount.argtypes = [self._ct.c_void_p, self._ct.POINTER(self._ct.c_int)]
        self._check(self.lib.ncclCommCount(self.comm, self._ct.byref(n)), "ncclCommCount")
        return int(n.value)

    def detach(self, net):
        from xingtian_amd import lib as L
        self.status(net)                     # (collects an error the library saw inside the hook)
        L.check(net.lib.xt_net_set_rccl(net.handle, None, None, 0), "xt_net_set_rccl")

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [self._ct.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = self._ct.c_void_p()


class TorchDistExchange(object):
    """The gradient-exchange hook served by ``torch.distributed`` from a host callback: the hook synchronises the
    stream it is given, all-reduces the sub-range of ``net.grads`` it was called for, and returns -- host-synchronous,
    so it cannot be captured into a hipGraph.  This is the TEST vehicle of the hook plumbing (bucket boundaries, the
    side-stream fork / join of ``XT_XCHG_OVERLAP``) on boxes where RCCL cannot form a group (two ranks on one GPU go
    through gloo); measurements use ``RcclComm``."""

    def __init__(self, net):
        import ctypes
        self._ct = ctypes
        import weakref
        net_ref = weakref.ref(net)           # (the net may cache this object: no reference cycle)
        self.calls = []                      # (offset, count) of every call -- only while `record_calls` (the tests)
        self.record_calls = False
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

        def _exchange(grads, count, user, stream):
            try:
                net = net_ref()
                hip.hipStreamSynchronize(stream)
                off = (int(grads) - net.grads.data_ptr()) // 4
                if self.record_calls:
                    self.calls.append((off, int(count)))
                view = net.grads_xchg[off:off + int(count)]       # (gradient + the data-parallel tail behind it)
                allreduce_sum_(view)
                torch.cuda.synchronize()
                return 0
            except Exception:        # noqa: BLE001 -- an exception must not unwind through the C frame
                return 1

        self._cb = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)(_exchange)

        self._net_ref = net_ref

    @property
    def net(self):
        return self._net_ref()

    def attach(self, overlap=False):
        from xingtian_amd import lib as L
        net = self.net
        L.check(net.lib.xt_net_set_grad_exchange_ex(net.handle, self._ct.cast(self._cb, self._ct.c_void_p), None,
                                                    L.XCHG_OVERLAP if overlap else 0), "xt_net_set_grad_exchange_ex")

    def detach(self):
        from xingtian_amd import lib as L
        net = self.net
        if net is not None:
            L.check(net.lib.xt_net_set_grad_exchange(net.handle, None, None), "xt_net_set_grad_exchange")


class DirectComm(object):
    """The direct 2-phase all-reduce over peer-mapped device memory (C ABI ``xt_direct_*`` / ``xt_allreduce_direct``,
    ``csrc/xt_xgmi.hip``): every rank pushes slice q of its gradient into peer q's inbox, rank q sums its slice in fixed
    rank order and pushes the result to everybody -- two hops over the xGMI mesh instead of the 2 (N-1) hops of a ring,
    replicas bit-identical by construction.  Kernels only, so the exchange is captured into the hipGraph of an update.

    ``DirectComm(rank, world, max_count)`` creates this rank's exchange block; ``connect()`` gathers the 64-byte IPC
    handles through the ``torch.distributed`` group that already exists (any backend; ``handles=`` passes them
    explicitly) and maps the peers' blocks.  N processes on ONE GPU take exactly the same code path (the tests)."""

    def __init__(self, rank, world, max_count, timeout_ms=None):
        import ctypes
        from xingtian_amd import lib as L
        self._ct, self._L = ctypes, L
        self.lib = L.load()
        self.rank, self.world, self.max_count = int(rank), int(world), int(max_count)
        self._handle_buf = ctypes.create_string_buffer(L.DIRECT_HANDLE_BYTES)
        self.comm = ctypes.c_void_p()
        L.check(self.lib.xt_direct_create(self.rank, self.world, self.max_count, self._handle_buf, ctypes.byref(self.comm)),
                "xt_direct_create")
        if timeout_ms:
            L.check(self.lib.xt_direct_set_timeout_ms(self.comm, int(timeout_ms)), "xt_direct_set_timeout_ms")
        self.connected = self.world == 1
        if self.world == 1:       # a one-rank group is its own peer (the fused step then runs against the rank's own block)
            L.check(self.lib.xt_direct_connect(self.comm, None), "xt_direct_connect")

    @property
    def handle(self):
        """this rank's 64-byte hipIpcMemHandle_t"""
        return bytes(self._handle_buf.raw)

    def connect(self, handles=None, group=None):
        """Map every peer's exchange block.  ``handles``: list of ``world`` 64-byte handles in rank order; default: gathered
        over ``torch.distributed`` (``group`` or the default group)."""
        if self.connected:
            return self
        if handles is None:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, self.handle, group=group)
            handles = gathered
        if len(handles) != self.world or any(len(h) != self._L.DIRECT_HANDLE_BYTES for h in handles):
            raise ValueError("DirectComm.connect: need {} handles of {} bytes".format(self.world, self._L.DIRECT_HANDLE_BYTES))
        blob = b"".join(bytes(h) for h in handles)
        self._L.check(self.lib.xt_direct_connect(self.comm, self._ct.c_char_p(blob)), "xt_direct_connect")
        self.connected = True
        return self

    @staticmethod
    def local_group(world, max_count, timeout_ms=None):
        """``world`` logical ranks inside THIS process (one device): tests, and single-process multi-stream drivers."""
        import ctypes
        ranks = [DirectComm(r, world, max_count, timeout_ms) for r in range(world)]
        arr = (ctypes.c_void_p * world)(*[c.comm for c in ranks])
        for c in ranks:
            if world > 1:
                c._L.check(c.lib.xt_direct_connect_local(c.comm, arr), "xt_direct_connect_local")
            c.connected = True
        return ranks

    @staticmethod
    def all_reduce_group_(ranks, bufs, streams):
        """in-process group: one all-reduce over ``bufs[r]`` (float32 device tensors) on ``streams[r]``, phase-ordered"""
        import ctypes
        n = len(ranks)
        c0 = ranks[0]
        comms = (ctypes.c_void_p * n)(*[c.comm for c in ranks])
        ptrs = (ctypes.c_void_p * n)(*[b.data_ptr() for b in bufs])
        sts = (ctypes.c_void_p * n)(*[s.cuda_stream for s in streams])
        c0._L.check(c0.lib.xt_allreduce_direct_group(n, comms, ptrs, int(bufs[0].numel()), sts), "xt_allreduce_direct_group")

    def set_fused(self, fused):
        """one launch per all-reduce (default) or the three-launch form (``xt_direct_set_fused``)"""
        self._L.check(self.lib.xt_direct_set_fused(self.comm, 1 if fused else 0), "xt_direct_set_fused")
        return self

    def all_reduce_(self, flat, stream_ptr=None):
        """in-place SUM of a float32 device tensor over the ranks, enqueued on the given (default: current) stream"""
        sp = stream_ptr if stream_ptr is not None else self._L.stream_ptr()
        self._L.check(self.lib.xt_allreduce_direct(self.comm, self._L.ptr(flat), int(flat.numel()), sp), "xt_allreduce_direct")
        return flat

    def attach(self, net, overlap=False):
        """install as ``net``'s gradient exchange (``xt_net_set_grad_exchange_ex`` with the library's own adapter: no
        Python frame on the enqueue path).  One bucket per step: two concurrent exchanges would share the sequence."""
        if overlap:
            raise ValueError("DirectComm: the two-bucket overlap mode needs one comm per bucket; not supported")
        if int(net.params.numel()) > self.max_count:
            raise ValueError("DirectComm: the net has {} parameters, the comm was sized for {}".format(net.params.numel(), self.max_count))
        fn = self._ct.cast(self.lib.xt_direct_exchange_hook, self._ct.c_void_p)
        self._L.check(net.lib.xt_net_set_grad_exchange_ex(net.handle, fn, self.comm, 0), "xt_net_set_grad_exchange_ex")

    def attach_fused(self, net):
        """the exchange FUSED into the SGD step (C ABI ``xt_net_set_direct``; needs ``xt_net_set_dp`` on the net and a comm
        sized for ``net.grads_xchg``): the gradient reduction writes straight into the owners' inboxes, one small launch
        reduces this rank's slice and leaves the squared-norm partials, the optimiser reads the exchange block -- three
        launches behind the backward pass, no scatter / gather copies."""
        if int(net.grads_xchg.numel()) > self.max_count:
            raise ValueError("DirectComm: the exchanged buffer holds {} floats, the comm was sized for {}".format(
                net.grads_xchg.numel(), self.max_count))
        self._L.check(net.lib.xt_net_set_direct(net.handle, self.comm), "xt_net_set_direct")

    def detach(self, net):
        self._L.check(net.lib.xt_net_set_direct(net.handle, None), "xt_net_set_direct")

    def info(self):
        """dict(ranks_on_device, block_cap): how many ranks of the group share THIS rank's device (1 on a multi-GPU node) and
        the workgroup cap its spinning launches get (resident workgroups / ranks on the device)"""
        c = self._ct
        a, b = c.c_int32(0), c.c_int32(0)
        self._L.check(self.lib.xt_direct_info(self.comm, c.byref(a), c.byref(b)), "xt_direct_info")
        return dict(ranks_on_device=int(a.value), block_cap=int(b.value))

    def read_result(self, count):
        """the reduced buffer of the most recent exchange as a host array (tests; synchronises the device)"""
        out = np.empty(int(count), np.float32)
        self._L.check(self.lib.xt_direct_read_result(self.comm, self._ct.c_void_p(out.ctypes.data), int(count)),
                      "xt_direct_read_result")
        return out

    def reset(self, group=None):
        """COLLECTIVE: clear the sticky error word, tickets, sequence number and this rank's flags between two barriers
        (every rank calls it, no exchange in flight) -- the way back from a time-out without rebuilding the group"""
        if self.world > 1:
            dist.barrier(group=group)
        self._L.check(self.lib.xt_direct_reset(self.comm), "xt_direct_reset")
        if self.world > 1:
            dist.barrier(group=group)

    def status(self):
        """dict(calls, seq, error_bits): ``error_bits`` != 0 -> a bounded wait ran out (1: a peer's scatter data, 2: a peer's
        reduced slice) or the ranks held different numbers of rows (4).  Synchronises the device."""
        c = self._ct
        v = [c.c_int32(0) for _ in range(3)]
        self._L.check(self.lib.xt_direct_status(self.comm, *[c.byref(x) for x in v]), "xt_direct_status")
        return dict(zip(("calls", "seq", "error_bits"), (int(x.value) for x in v)))

    def destroy(self):
        if self.comm:
            self.lib.xt_direct_destroy(self.comm)
            self.comm = self._ct.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:       # noqa: BLE001
            pass


class LearnerDP(object):
    """Data parallelism of ONE learner rank behind the plugin classes (``PPO`` / ``IMPALAOpt`` models): what
    ``model_config`` / the launcher environment ask for, the gradient exchange attached to the network, who publishes.

    The reference builds one algorithm per learner (xt/framework/learner.py:518-525) and its only multi-device analogue
    averages ``get_grad()`` results between processes on the host (xt/framework/trainer.py:32-136, dead code).  Here N
    learner processes (``torchrun --nproc-per-node N``, one per GPU) each build the SAME Algorithm / Model pair; the model
    finds ``WORLD_SIZE > 1`` (or ``model_config.DP``) and becomes one replica:

    ``model_config`` keys (all optional)
      DP           "strict" (default when WORLD_SIZE > 1) | "weak" | "off"
                   strict: the reference's GLOBAL minibatch / chunk is split over the ranks (PPO: BATCH_SIZE rows, loss
                   means over the global rows; IMPALA: whole-trajectory shards of every BATCH_SIZE chunk, sum-form loss):
                   the single-GPU update up to fp32 summation order.  weak: every rank trains full BATCH_SIZE minibatches
                   of its own trajectories, gradients averaged (PPO) / summed (IMPALA): global batch = N x BATCH_SIZE, a
                   flagged semantic change.
      DP_FEED      how rollout messages reach the ranks
                   "replicated" (strict default): every rank is handed every message in the same order and keeps all of
                   them; shared epoch permutations -> exactly the single-GPU minibatches.
                   "round_robin": every rank is handed every message and keeps message k iff k % N == rank ("shard whole
                   trajectories over ranks at ingest"); "sharded" (weak default): every rank is handed only its own.
                   With these two a strict PPO minibatch is BATCH_SIZE/N rows of EACH rank's local permutation (stratified
                   over the ranks instead of one global shuffle: documented deviation, SURVEY 8(e) "permute within shards").
      DP_EXCHANGE  "rccl" (default: raw ncclAllReduce enqueued by the library, captured into the update's hipGraph) |
                   "direct" (the 2-phase exchange over peer-mapped memory FUSED into the step, xt_net_set_direct, also
                   in-graph.  EXPERIMENTAL across devices: every run so far had all ranks on ONE GPU -- up to eight
                   processes, tests/test_gpu_dp_ranks.py; its cross-device ordering (uncached peer writes + s_waitcnt
                   before the flag store) has not met real xGMI links.  ``tools/multi_gpu_preflight.py`` checks it against
                   torch.distributed in under a minute on any box with >= 2 GPUs) | "torch" (torch.distributed from a
                   host callback: any backend, not capturable -- tests on one GPU go through gloo)
      DP_GRAPH     capture the data-parallel update into a hipGraph?  Default: yes for "direct" (kernels only), NO for
                   "rccl": RCCL collectives inside a replayed hipGraph have only ever been validated with a 1-rank
                   communicator here, and the eager form (ONE C call per update enqueues every kernel and every
                   ncclAllReduce) measured the same on one rank (7.73 vs 7.77 ms); `DP_GRAPH: true` opts in.
      DP_BACKEND   torch.distributed backend when the process group does not exist yet ("nccl")
      DP_DEVICE    device index (default LOCAL_RANK)
      DP_PUBLISH   "rank0" | "all": which ranks answer ``checkpoint_ready`` / ``if_save`` with True, i.e. hand weights to
                   explorers and write checkpoints.  Default: rank 0 only when the ranks share one message stream
                   (replicated / round_robin); every rank when each has its own explorers (sharded) -- the replicas are
                   bit-identical, so any of them may serve.
    """

    def __init__(self, rank, world, mode, feed, exchange, publish=None, graph=None):
        self.rank, self.world, self.mode, self.feed, self.exchange = rank, world, mode, feed, exchange
        self.graph = (exchange == "direct") if graph is None else bool(graph)
        self.publish = publish or ("all" if feed == "sharded" else "rank0")
        if self.publish not in ("rank0", "all"):
            raise ValueError("model_config.DP_PUBLISH must be rank0 | all, got {!r}".format(self.publish))
        self.comm = None
        self._msg = 0

    @staticmethod
    def from_config(model_config, is_learner=True):
        """``is_learner``: the model is the learner's (``model_info["type"] == "learner"``, xt/framework/learner.py:544).  Only
        such a model becomes a replica IMPLICITLY (``WORLD_SIZE > 1`` without a ``DP`` key): attaching is a collective
        (weight broadcast, communicator set-up), and an evaluator-side model that happens to be built on a GPU inside a
        torchrun-launched process must not wait for peers that never build one.  An explicit ``DP`` key always counts."""
        import os
        cfg = model_config or {}
        mode = cfg.get("DP")
        env_world = int(os.environ.get("WORLD_SIZE", "1"))
        if mode in (None, "auto"):
            mode = "strict" if (env_world > 1 and is_learner) else "off"
        if mode not in ("strict", "weak", "off"):
            raise ValueError("model_config.DP must be 'strict', 'weak' or 'off', got {!r}".format(mode))
        if mode == "off":
            return None
        if not (dist.is_available() and dist.is_initialized()):
            if env_world <= 1:
                return None                    # DP asked for, but this is a single process: nothing to exchange with
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29544")
            dist.init_process_group(cfg.get("DP_BACKEND", "nccl"), rank=int(os.environ["RANK"]), world_size=env_world)
        rank, world = dist.get_rank(), dist.get_world_size()
        if world == 1:
            return None
        feed = cfg.get("DP_FEED", "replicated" if mode == "strict" else "sharded")
        if feed not in ("replicated", "round_robin", "sharded"):
            raise ValueError("model_config.DP_FEED must be replicated | round_robin | sharded, got {!r}".format(feed))
        if mode == "weak" and feed == "replicated":
            raise ValueError("DP 'weak' trains every rank on its own trajectories: DP_FEED must be round_robin or sharded")
        exchange = cfg.get("DP_EXCHANGE", "rccl")
        if exchange not in ("rccl", "direct", "torch"):
            raise ValueError("model_config.DP_EXCHANGE must be rccl | direct | torch, got {!r}".format(exchange))
        return LearnerDP(rank, world, mode, feed, exchange, cfg.get("DP_PUBLISH"), cfg.get("DP_GRAPH"))

    @staticmethod
    def device_index(model_config):
        import os
        cfg = model_config or {}
        if cfg.get("DP_DEVICE") is not None:
            return int(cfg["DP_DEVICE"])
        return int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)

    # ---- set-up
    def attach(self, net, loss_scale=1.0):
        """replicas start identical (rank 0's parameters and optimiser slots), the data-parallel tail is switched on
        (``xt_net_set_dp``: rows + loss shares travel behind the gradient, so a train needs no host collective) and the
        exchange is installed on ``net``.  ``loss_scale``: 1 when the ranks' losses are shares of one sum (PPO strict, IMPALA),
        1 / world for the mean of the ranks' means (PPO weak)."""
        from xingtian_amd import lib as L
        for t in (net.params, net.adam_m, net.adam_v, net.adam_state):
            dist.broadcast(t, src=0)
        torch.cuda.synchronize()
        net.touch()
        net.set_dp(self.rank, self.world, loss_scale)
        if self.exchange == "rccl":
            self.comm = RcclComm(self.rank, self.world)
            warm = torch.zeros(1024, dtype=torch.float32, device=net.device)
            self.comm.all_reduce_(warm, L.stream_ptr())       # RCCL's lazy allocations must not happen under capture
            torch.cuda.synchronize()
            self.comm.attach(net)
        elif self.exchange == "direct":
            self.comm = DirectComm(self.rank, self.world, int(net.grads_xchg.numel())).connect()
            self._direct_self_test(net.device)
            self.comm.attach_fused(net)
        else:
            self.comm = TorchDistExchange(net)
            self.comm.attach()
        return self

    def _direct_self_test(self, device):
        """one value-checked all-reduce through the freshly connected exchange blocks BEFORE the learner depends on them:
        integer-valued data whose sum is exact in float32, so any stale / torn / misdirected slice shows (the cross-device
        ordering of the direct exchange has only ever been exercised with all ranks on one GPU)"""
        n = 4099
        base = torch.arange(n, dtype=torch.float32, device=device) % 977.0
        buf = base * float(self.rank + 1)
        self.comm.all_reduce_(buf)
        torch.cuda.synchronize()
        want = base * float(self.world * (self.world + 1) // 2)
        st = self.comm.status()
        if st["error_bits"] or not torch.equal(buf, want):
            bad = int((buf != want).sum().item())
            raise RuntimeError("DP_EXCHANGE direct: the start-up self-test all-reduce failed on rank {} (error bits {}, {} of {} "
                               "elements wrong) -- use DP_EXCHANGE rccl on this machine and run tools/multi_gpu_preflight.py".format(
                                   self.rank, st["error_bits"], bad, n))

    def exchanged_gradient(self, net):
        """the gradient the last SGD step applied (after the exchange) as a host array (tests)"""
        torch.cuda.synchronize()
        if isinstance(self.comm, DirectComm):
            return self.comm.read_result(int(net.grads.numel()))
        return net.grads.detach().cpu().numpy().copy()

    @property
    def graph_capable(self):
        """may the data-parallel update be captured into a hipGraph?  The torch.distributed callback is host-synchronous
        (never); kernels-only exchanges yes; RCCL calls only when DP_GRAPH asks for it (see the class docstring)"""
        return self.exchange in ("rccl", "direct") and self.graph

    @property
    def is_publisher(self):
        return self.rank == 0 or self.publish == "all"

    def shared_seed(self, seed):
        """one seed for the generators that must agree on every rank (strict + replicated: the epoch permutations)"""
        t = torch.tensor([int(seed) if seed is not None else int(np.random.SeedSequence().entropy % (1 << 62))],
                         dtype=torch.int64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, src=0)
        return int(t.item())

    # ---- ingest
    def new_rollout(self):
        self._msg = 0

    def takes(self, _train_data=None):
        """does THIS rank keep the next arriving message?  (round_robin: message k belongs to rank k % N)"""
        k = self._msg
        self._msg += 1
        return self.feed != "round_robin" or k % self.world == self.rank

    # ---- PPO
    def ppo_cfg(self, net, cfg):
        """(xt_ppo_cfg, rows of a local minibatch) for this rank"""
        bsz = int(cfg["BATCH_SIZE"])
        if self.mode == "weak":
            return net.make_ppo_cfg(cfg, grad_scale=1.0 / self.world, global_batch=0), bsz
        if self.feed == "replicated":      # shared permutation, this rank's rows of every global minibatch (in the library)
            return net.make_ppo_cfg(cfg, grad_scale=1.0, global_batch=0, shard_rank=self.rank, shard_world=self.world), bsz
        if bsz % self.world:
            raise ValueError("strict data parallelism over sharded trajectories needs BATCH_SIZE ({}) divisible by the "
                             "number of ranks ({})".format(bsz, self.world))
        local = bsz // self.world
        return net.make_ppo_cfg(dict(cfg, BATCH_SIZE=local), grad_scale=1.0, global_batch=bsz), local

    def status(self):
        """exchange health (synchronises the device; the trains themselves learn of an error from ``loss_acc[2]``, which
        travels with the loss): raises if a bounded wait of the direct exchange ran out or the rows disagreed"""
        from xingtian_amd import lib as L
        if isinstance(self.comm, DirectComm):
            st = self.comm.status()
            if st["error_bits"]:
                raise RuntimeError("direct all-reduce: {} (error bits {})".format(L.dp_error_text(st["error_bits"]),
                                                                                st["error_bits"]))
            return st
        return {}
