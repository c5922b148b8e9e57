"""Rollout ingest for the learner (SURVEY.md section 8 "next" row f1): trajectories are copied into PINNED host
staging buffers as they arrive (``Algorithm.prepare_data``) and shipped to HBM with asynchronous copies on a
dedicated HIP stream, so that by the time ``train()`` is called the rollout is (mostly) resident and the
reference's ``np.concatenate`` of the whole rollout (xt/algorithm/ppo/ppo.py:66-71) plus the pageable
host->device upload are gone from the critical path.  Two buffer sets alternate so that the next rollout can
stream in while the previous update still reads the other set.

Plumbing only (PyTorch-ROCm tensors / streams); no arithmetic happens here.
"""
import numpy as np
import torch

# label fields of a PPO trajectory (xt/algorithm/ppo/ppo.py:79-85): name, device dtype, per-row width (0 = scalar)
PPO_FIELDS = (("action", torch.int32, 0), ("old_logp", torch.float32, 0), ("adv", torch.float64, 0),
              ("old_v", torch.float32, 0), ("target_v", torch.float64, 0))


def impala_fields(action_dim):
    """label fields of an IMPALAOpt rollout message (xt/algorithm/impala/impala_opt.py:116-147): behaviour logits
    [n, A] f32, actions i32, dones (bool -> u8), rewards (float64 on the wire, float32 at the placeholder)."""
    return (("logit", torch.float32, int(action_dim)), ("action", torch.int32, 0), ("done", torch.uint8, 0),
            ("reward", torch.float32, 0))




class _BufferSet(object):
    def __init__(self, cap, obs_tail, obs_u8, n_epochs, device, sig, fields=PPO_FIELDS):
        self.cap = cap
        self.sig = sig
        tdt = torch.uint8 if obs_u8 else torch.float32
        self.host = {"obs": torch.empty((cap,) + obs_tail, dtype=tdt, pin_memory=True)}
        self.dev = {"obs": torch.empty((cap,) + obs_tail, dtype=tdt, device=device)}
        for name, tdt2, width in fields:
            shape = (cap, width) if width else (cap,)
            self.host[name] = torch.empty(shape, dtype=tdt2, pin_memory=True)
            self.dev[name] = torch.empty(shape, dtype=tdt2, device=device)
        if n_epochs > 0:
            self.dev["perm"] = torch.empty((n_epochs, cap), dtype=torch.int32, device=device)
        self.host_np = {k: v.numpy() for k, v in self.host.items()}
        self.done = torch.cuda.Event()
        self.free = None          # recorded on the compute stream after the update that read this set


class RolloutIngest(object):
    def __init__(self, device, n_epochs, initial_capacity=4096, obs_u8=None, fields=PPO_FIELDS):
        """``obs_u8``: the observation type the NETWORK reads (``spec.input_xform[0]``: uint8 frames vs float32);
        arriving arrays of another dtype are cast into the staging buffer like the upload path casts them.  None
        = take the dtype of the first array (stand-alone use).  ``fields``: the label arrays that travel with the
        observations (``PPO_FIELDS`` / ``impala_fields(A)``); ``n_epochs`` = 0: no permutation buffer."""
        self.obs_u8 = obs_u8
        self.fields = tuple(fields)
        self.device = torch.device(device)
        self.n_epochs = n_epochs
        self.initial_capacity = initial_capacity
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.sets = [None, None]
        self.cur = 0
        self.n = 0

    # ------------------------------------------------------------------
    def _ensure(self, need, obs):
        u8 = (obs.dtype == np.uint8) if self.obs_u8 is None else bool(self.obs_u8)
        sig = (tuple(obs.shape[1:]), u8)
        s = self.sets[self.cur]
        same = s is not None and s.sig == sig          # every buffer set remembers the layout it was built for
        if same and s.cap >= need:
            return s
        cap = max(self.initial_capacity, need)
        if same:
            cap = max(cap, 2 * s.cap)
        new = _BufferSet(cap, tuple(obs.shape[1:]), u8, self.n_epochs, self.device, sig, self.fields)
        if same and self.n > 0:                        # grow: keep what was already ingested
            self.copy_stream.synchronize()
            for k in new.host:
                new.host[k][:self.n].copy_(s.host[k][:self.n])
            with torch.cuda.stream(self.copy_stream):
                for k in new.host:
                    new.dev[k][:self.n].copy_(new.host[k][:self.n], non_blocking=True)
        self.sets[self.cur] = new
        return new

    def put(self, obs, *labels, pinned=False):
        """Append one trajectory / rollout message ([T,...] arrays as the explorer ships them, labels in the order
        of ``fields``) and start its H2D copy.  ``pinned``: the arrays are views into page-locked memory (a pinned
        transport ring): fields whose dtype already is the device dtype are DMA-copied straight from the source, and
        the call returns only when those copies have landed (the caller recycles the slot right after)."""
        obs = np.asarray(obs)
        t = obs.shape[0]
        s = self._ensure(self.n + t, obs)
        if self.n == 0 and s.free is not None:      # do not overwrite a set an enqueued update still reads
            self.copy_stream.wait_event(s.free)
        lo, hi = self.n, self.n + t
        if len(labels) != len(self.fields):
            raise ValueError("RolloutIngest.put: {} label arrays for fields {}".format(
                len(labels), [f[0] for f in self.fields]))
        direct = {}
        named = [("obs", obs)] + [(f[0], np.asarray(a)) for f, a in zip(self.fields, labels)]
        for name, arr in named:
            dst = s.host_np[name][lo:hi]
            if pinned and isinstance(arr, np.ndarray) and arr.dtype == dst.dtype and arr.flags.c_contiguous \
                    and arr.flags.writeable and arr.size == dst.size:
                direct[name] = torch.from_numpy(arr.reshape(dst.shape))     # DMA source = the pinned transport slot
            else:
                # bool -> uint8 and float64 -> float32 are the casts the reference's placeholders apply
                np.copyto(dst, arr.reshape(dst.shape), casting="unsafe" if (name == "obs" or arr.dtype == np.bool_)
                          else "same_kind")
        with torch.cuda.stream(self.copy_stream):
            for k in s.host:
                s.dev[k][lo:hi].copy_(direct[k] if k in direct else s.host[k][lo:hi], non_blocking=True)
        if direct:
            self.copy_stream.synchronize()
        self.n = hi

    def finish(self):
        """All trajectories are in: make the compute stream wait for the copies, return (n, device buffers) and
        switch to the other buffer set for the next rollout."""
        s = self.sets[self.cur]
        n = self.n
        if s is None or n == 0:
            raise RuntimeError("RolloutIngest.finish(): nothing was ingested")
        s.done.record(self.copy_stream)
        torch.cuda.current_stream(self.device).wait_event(s.done)
        self.cur ^= 1
        self.n = 0
        self.last = s
        return n, s.dev

    def mark_consumed(self):
        """call after the update that reads the last finished set has been enqueued on the compute stream"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.last.free = ev

    def reset(self):
        self.n = 0
