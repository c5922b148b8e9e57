"""Data-parallel learner plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" == RCCL over
xGMI on ROCm; "gloo" on CPU for tests).

The learner update shards naturally: samples of a minibatch are independent through
forward/backward and GAE / v-trace recur only along time, so every rank owns whole trajectories and a
slice of every minibatch; the ONLY exchange is one all-reduce (SUM) of the flat fp32 gradient buffer per SGD
step, after which every rank applies the identical clip + Adam update (replicas stay bit-identical because
the reduced buffer is identical on all ranks).  PPO's loss is a mean over the global minibatch -> scale the
summed gradient by 1/world (``grad_scale``); IMPALA's loss is a sum -> no scaling.

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
    """Contiguous, balanced [begin, end) shard of n_items (trajectories) for ``rank``."""
    base, rem = divmod(n_items, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def split_minibatch(perm_row, start, batch_size, rank, world):
    """Rows of the GLOBAL minibatch perm_row[start:start+batch_size] owned by ``rank`` (equal shards; the global
    permutation is drawn once with a shared seed so that every rank partitions it identically)."""
    mb = np.asarray(perm_row[start:start + batch_size])
    b, e = shard_range(len(mb), rank, world)
    return mb[b:e]


def allreduce_sum_(flat_grad):
    """In-place SUM all-reduce of the flat gradient buffer (no-op for a single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


def grad_scale(loss_reduction, world):
    """Factor applied to the summed gradient: 'mean' losses (PPO) -> 1/world, 'sum' losses (IMPALA) -> 1."""
    if loss_reduction == "mean":
        return 1.0 / world
    if loss_reduction == "sum":
        return 1.0
    raise ValueError(loss_reduction)


def broadcast_weights_(flat_params, src=0):
    """Make replicas identical at start-up (random init differs per process unless seeded identically)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params, src=src)
    return flat_params


def dp_ppo_step(net, cfg_struct, lr, max_grad_norm, obs, idx, action, old_logp, adv, old_v, target_v, world,
                overlap=False):
    """One data-parallel PPO SGD step on a HipActorCritic: local fwd/bwd -> RCCL all-reduce (SUM) of net.grads ->
    identical clip+Adam on every rank.

    ``overlap``: two buckets.  The tail of the flat gradient (the Dense layer feeding the heads + the heads, 95 %
    of PpoCnn's 3.39 MB) is final after the first backward launch, so its all-reduce is issued asynchronously
    (RCCL's own stream; xGMI rings are per-link bound, >= 40 us for 3.2 MB at 8 GPUs) and overlaps the conv
    backward (~85 us); only the 170 KB head of the buffer is reduced after it.  Each bucket is summed by RCCL in
    a fixed order, so replicas stay bit-identical.  Costs two more c10d calls per step (~60 us of host time, measured
    with a 1-rank RCCL group: the eager step becomes host-bound), hence opt-in: worth it when the all-reduce
    itself is slower than that (more ranks / slower links)."""
    grouped = dist.is_available() and dist.is_initialized()
    if not overlap or not grouped:
        net.ppo_step(cfg_struct, obs, idx, action, old_logp, adv, old_v, target_v, apply=False)
        allreduce_sum_(net.grads)
    else:
        tail = net.ppo_step_begin(cfg_struct, obs, idx, action, old_logp, adv, old_v, target_v)
        w_tail = dist.all_reduce(net.grads[tail:], op=dist.ReduceOp.SUM, async_op=True)
        net.ppo_step_end(cfg_struct, obs, idx)
        w_head = dist.all_reduce(net.grads[:tail], op=dist.ReduceOp.SUM, async_op=True) if tail > 0 else None
        w_tail.wait()
        if w_head is not None:
            w_head.wait()
    net.apply(lr, max_grad_norm, grad_scale=grad_scale("mean", world))


class DpGraphStepper(object):
    """Data-parallel PPO SGD steps with the compute segments replayed from hipGraphs.

    The eager step enqueues ~12 kernels through three ctypes calls plus one or two c10d calls per SGD step; at
    ~155 us of GPU work per step that leaves the host little slack, and the two-bucket overlap (``dp_ppo_step``,
    ``overlap=True``) made the eager path host-bound.  Here each compute segment (forward + heads + first backward
    launch | rest of the backward | clip + Adam) is captured ONCE per minibatch size with ``torch.cuda.graph`` -- the
    library launches on torch's current stream, so its kernels are captured like torch's own -- and a step is
    ``copy the minibatch indices into a fixed buffer -> replay -> all-reduce -> replay -> all-reduce -> replay``.
    The all-reduces stay ordinary c10d calls on RCCL's stream (tied to the replays by c10d's stream events), exactly
    as in the eager form, so the arithmetic, its order and therefore the replicas' bits are unchanged.

    Any failure while capturing (an unsupported call under capture, a driver refusing it) permanently falls back to
    the eager ``dp_ppo_step`` -- a benchmark must never die on an optimisation.
    """

    def __init__(self, net, cfg_struct, lr, max_grad_norm, obs, action, old_logp, adv, old_v, target_v, world,
                 overlap=True, warm_steps=2):
        self.net, self.cfg, self.lr, self.clip = net, cfg_struct, lr, max_grad_norm
        self.data = (obs, action, old_logp, adv, old_v, target_v)
        self.world, self.overlap = world, bool(overlap)
        self.grouped = dist.is_available() and dist.is_initialized()
        self.idx_buf = torch.empty((net.max_batch,), dtype=torch.int32, device=net.params.device)
        self.graphs = {}
        self.eager_left = int(warm_steps)     # the first steps run eagerly: one-time host work stays outside captures
        self.failed = False
        self.scale = grad_scale("mean", world)

    def _eager(self, idx):
        obs, action, old_logp, adv, old_v, target_v = self.data
        dp_ppo_step(self.net, self.cfg, self.lr, self.clip, obs, idx, action, old_logp, adv, old_v, target_v,
                    self.world, overlap=False)

    def _capture(self, b):
        net = self.net
        obs, action, old_logp, adv, old_v, target_v = self.data
        idx = self.idx_buf[:b]
        seg = {}
        g1 = torch.cuda.CUDAGraph()
        if self.overlap and self.grouped:
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                seg["tail"] = net.ppo_step_begin(self.cfg, obs, idx, action, old_logp, adv, old_v, target_v)
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, capture_error_mode="thread_local"):
                net.ppo_step_end(self.cfg, obs, idx)
            seg["end"] = g2
        else:
            with torch.cuda.graph(g1, capture_error_mode="thread_local"):
                net.ppo_step(self.cfg, obs, idx, action, old_logp, adv, old_v, target_v, apply=False)
        seg["begin"] = g1
        g3 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g3, capture_error_mode="thread_local"):
            net.apply(self.lr, self.clip, grad_scale=self.scale)
        seg["apply"] = g3
        return seg

    def step(self, idx):
        """One SGD step on the rows ``idx`` (int32 device tensor) of the resident rollout."""
        if self.failed or self.eager_left > 0:
            self.eager_left -= 1
            return self._eager(idx)
        b = int(idx.numel())
        seg = self.graphs.get(b)
        if seg is None:
            try:
                seg = self._capture(b)
            except Exception as exc:       # noqa: BLE001 -- fall back, never fail the run
                import sys
                print("[xingtian_amd.parallel] hipGraph capture of the data-parallel step failed (%r): eager path"
                      % (exc,), file=sys.stderr)
                self.failed = True
                return self._eager(idx)
            self.graphs[b] = seg
        net = self.net
        self.idx_buf[:b].copy_(idx, non_blocking=True)
        seg["begin"].replay()
        if "end" in seg:
            tail = seg["tail"]
            w_tail = dist.all_reduce(net.grads[tail:], op=dist.ReduceOp.SUM, async_op=True)
            seg["end"].replay()
            w_head = dist.all_reduce(net.grads[:tail], op=dist.ReduceOp.SUM, async_op=True) if tail > 0 else None
            w_tail.wait()
            if w_head is not None:
                w_head.wait()
        else:
            allreduce_sum_(net.grads)
        seg["apply"].replay()


class RcclComm(object):
    """A raw RCCL communicator for the gradient-exchange hook of ``xt_net_ppo_train`` (C ABI >= 4).

    ``torch.distributed`` owns its communicator and issues collectives on its own stream from Python, one call per
    SGD step.  The hook form instead enqueues ``ncclAllReduce`` on the stream the learner kernels run on, from inside
    ``xt_net_ppo_train`` -- so the 52 all-reduces of an update are captured into the update's hipGraph and replayed
    without any host involvement.  The communicator is created with ctypes on the ``librccl.so`` that torch ships
    (already loaded in the process); the 128-byte unique id travels through the existing ``torch.distributed`` group.

    Opt-in (``bench.py --dp-mode ingraph``): validated with a 1-rank communicator on one GPU (bit-identical to the
    step-wise path); multi-rank runs were not possible in the round that added it.
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

        def _exchange(grads, count, user, stream):
            rc = self.lib.ncclAllReduce(grads, grads, count, self.NCCL_FLOAT32, self.NCCL_SUM, self.comm, stream)
            if rc != 0:
                self.errors.append(rc)
            return rc

        self._cb_type = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p)
        self._cb = self._cb_type(_exchange)          # keep alive: the library stores the raw pointer

    @staticmethod
    def _check(rc, what):
        if rc != 0:
            raise RuntimeError("%s failed with ncclResult %d" % (what, rc))

    def all_reduce_(self, flat, stream_ptr):
        """Eager in-place SUM all-reduce of a float32 device tensor on the given stream (also warms RCCL up before
        the first capture: its lazy allocations must not happen under stream capture)."""
        self._check(self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), self.NCCL_FLOAT32,
                                           self.NCCL_SUM, self.comm, stream_ptr), "ncclAllReduce")

    def attach(self, net):
        from xingtian_amd import lib as L
        L.check(net.lib.xt_net_set_grad_exchange(net.handle, self._ct.cast(self._cb, self._ct.c_void_p), None),
                "xt_net_set_grad_exchange")

    def detach(self, net):
        from xingtian_amd import lib as L
        L.check(net.lib.xt_net_set_grad_exchange(net.handle, None, None), "xt_net_set_grad_exchange")

    def destroy(self):
        if self.comm:
            self.lib.ncclCommDestroy.argtypes = [self._ct.c_void_p]
            self.lib.ncclCommDestroy(self.comm)
            self.comm = self._ct.c_void_p()
