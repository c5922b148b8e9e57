"""Rollout ingest for the learner (SURVEY.md section 8 "next" row f1): trajectories are copied into PINNED host
staging buffers as they arrive (``Algorithm.prepare_data``) and shipped to HBM with asynchronous copies on a
dedicated HIP stream, so that by the time ``train()`` is called the rollout is (mostly) resident and the
reference's ``np.concatenate`` of the whole rollout (xt/algorithm/ppo/ppo.py:66-71) plus the pageable
host->device upload are gone from the critical path.  Two buffer sets alternate so that the next rollout can
stream in while the previous update still reads the other set.

The frames of a trajectory (the only large field: 3.6 MB per 128-step Atari trajectory) are staged by the library's
native worker pool (``xt_stage_rows``: chunked, the H2D of chunk k enqueued while chunk k+1 is being staged, GIL
released); the label fields (a few hundred bytes each) are written into their pinned arrays and shipped with ONE copy
per field when the rollout is complete (``finish``) instead of one per field and trajectory.

Plumbing only (PyTorch-ROCm tensors / streams); no arithmetic happens here.
"""
import ctypes

import numpy as np
import torch

from xingtian_amd import lib as L

_TUNED = {}


def staging_report():
    """What ``xt_stage_tune`` measured on this host (GB/s per variant) and picked; tuned once per process."""
    if not _TUNED:
        h = L.load()
        g = (ctypes.c_float * 10)()
        L.check(h.xt_stage_tune(32 << 20, g), "xt_stage_tune")
        t, nt = ctypes.c_int32(), ctypes.c_int32()
        h.xt_stage_get(ctypes.byref(t), ctypes.byref(nt))
        names = ["inline", "1", "2", "4", "8"]
        _TUNED.update(threads=t.value, non_temporal=bool(nt.value),
                      gbps={("nt_" if i >= 5 else "memcpy_") + names[i % 5]: round(float(g[i]), 2) for i in range(10)})
    return dict(_TUNED)

# label fields of a PPO trajectory (xt/algorithm/ppo/ppo.py:79-85): name, device dtype, per-row width (0 = scalar)
PPO_FIELDS = (("action", torch.int32, 0), ("old_logp", torch.float32, 0), ("adv", torch.float64, 0),
              ("old_v", torch.float32, 0), ("target_v", torch.float64, 0))


def impala_fields(action_dim):
    """label fields of an IMPALAOpt rollout message (xt/algorithm/impala/impala_opt.py:116-147): behaviour logits
    [n, A] f32, actions i32, dones (bool -> u8), rewards (float64 on the wire, float32 at the placeholder)."""
    return (("logit", torch.float32, int(action_dim)), ("action", torch.int32, 0), ("done", torch.uint8, 0),
            ("reward", torch.float32, 0))




class _BufferSet(object):
    """Pinned host staging + device buffers of one rollout.  The label fields share ONE pinned block and ONE device
    block (256-byte aligned sub-arrays), so the labels of a whole rollout travel with a single H2D copy."""

    def __init__(self, cap, obs_tail, obs_u8, n_epochs, device, sig, fields=PPO_FIELDS):
        self.cap = cap
        self.sig = sig
        tdt = torch.uint8 if obs_u8 else torch.float32
        self.host = {"obs": torch.empty((cap,) + obs_tail, dtype=tdt, pin_memory=True)}
        self.dev = {"obs": torch.empty((cap,) + obs_tail, dtype=tdt, device=device)}
        off, lay = 0, []
        for name, tdt2, width in fields:
            nbytes = cap * max(width, 1) * torch.empty((), dtype=tdt2).element_size()
            lay.append((name, tdt2, (cap, width) if width else (cap,), off, nbytes))
            off += (nbytes + 255) // 256 * 256
        self.lab_host = torch.empty((max(off, 256),), dtype=torch.uint8, pin_memory=True)
        self.lab_dev = torch.empty((max(off, 256),), dtype=torch.uint8, device=device)
        for name, tdt2, shape, o, nbytes in lay:
            self.host[name] = self.lab_host[o:o + nbytes].view(tdt2).view(shape)
            self.dev[name] = self.lab_dev[o:o + nbytes].view(tdt2).view(shape)
        if n_epochs > 0:
            self.dev["perm"] = torch.empty((n_epochs, cap), dtype=torch.int32, device=device)
        if any(f[0] == "boot" for f in fields):      # ragged-GAE row offsets: [n_traj + 1] <= cap + 1 entries
            self.host["offsets"] = torch.empty((cap + 1,), dtype=torch.int32, pin_memory=True)
            self.dev["offsets"] = torch.empty((cap + 1,), dtype=torch.int32, device=device)
        self.dev_padded = None      # [cap, H, W, pad4(C)] copy of the frames (RolloutIngest.pad_channels)
        self.host_np = {k: v.numpy() for k, v in self.host.items()}
        self.done = torch.cuda.Event()
        self.free = None          # recorded on the compute stream after the update that read this set
        self.dma_ticket = 0       # newest xt_dma_h2d_async copy into this set (RolloutIngest.dma_h2d); 0 = none
        self.used_stream = False  # a copy of the current rollout went through a HIP copy stream (then `done` matters)


# extra label fields of a trajectory that arrives WITHOUT advantages (value / reward / done instead): the inputs of the
# learner-side GAE (xt/agent/ppo/ppo.py:77-106 moved to the learner GPU, one xt_gae_f64_ragged per rollout)
PPO_RAW_FIELDS = (("reward", torch.float64, 0), ("done", torch.uint8, 0), ("boot", torch.float32, 0))


class _DmaDone(object):
    """completion handle of an ``xt_dma_h2d_async`` copy with the two methods a ``transport.SlotGuard`` calls on an event"""
    __slots__ = ("lib", "ticket")

    def __init__(self, lib, ticket):
        self.lib, self.ticket = lib, ticket

    def query(self):
        return self.lib.xt_dma_wait_upto(self.ticket, 0) == 0

    def synchronize(self):
        if self.lib.xt_dma_wait_upto(self.ticket, 30000) != 0:
            raise RuntimeError("xingtian_amd: a rollout copy (ticket {}) did not land within 30 s".format(self.ticket))


class RolloutIngest(object):
    def __init__(self, device, n_epochs, initial_capacity=4096, obs_u8=None, fields=PPO_FIELDS, pad_channels=None,
                 copy_streams=2):
        """``obs_u8``: the observation type the NETWORK reads (``spec.input_xform[0]``: uint8 frames vs float32);
        arriving arrays of another dtype are cast into the staging buffer like the upload path casts them.  None
        = take the dtype of the first array (stand-alone use).  ``fields``: the label arrays that travel with the
        observations (``PPO_FIELDS`` / ``impala_fields(A)``); ``n_epochs`` = 0: no permutation buffer."""
        self.obs_u8 = obs_u8
        self.fields = tuple(fields)
        # pad_channels = (c_dst, fill byte): image observations are staged with their own channel count and expanded on
        # the device to the multiple of 4 the first layer reads (xt_pad_channels), see netspec._conv
        self.pad_channels = pad_channels
        self._n_raw = len(PPO_RAW_FIELDS) if self.fields[-len(PPO_RAW_FIELDS):] == PPO_RAW_FIELDS else 0
        self.raw_traj = 0               # trajectories of the current rollout that came without advantages
        self.adv_traj = 0               # ... and with them
        self.device = torch.device(device)
        self.n_epochs = n_epochs
        self.initial_capacity = initial_capacity
        # the frames of consecutive trajectories alternate between `copy_streams` HIP streams: one stream issues its
        # copies back to back with a gap after every piece (a 3.6 MB trajectory reaches ~49 GB/s, one 115 MB copy 57);
        # two streams keep a second DMA queued while the first one's completion is processed
        self.copy_streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(copy_streams)))]
        self.copy_stream = self.copy_streams[0]         # labels, padding, growth copies; joins the others in finish()
        self._rr = 0
        self.sets = [None, None]
        self.cur = 0
        self.n = 0
        # True: the label block is NOT copied to HBM -- the kernels read the page-locked staging block directly over PCIe
        # (``mapped_labels``).  For per-message trains (IMPALA: a few KB of labels read once, by the v-trace kernel) the label
        # H2D was a runtime blit kernel that queued behind the weights publish and held the train's first kernel up by ~30 us
        # (rocprofv3 trace of the loop, round 6)
        self.zero_copy_labels = False
        self.generation = 0             # rollouts handed to the learner so far (finish() calls)
        # dma_h2d (set by the model: ImpalaCnnOpt with zero-copy labels): frames that sit in page-locked transport slots go
        # to HBM through xt_dma_h2d_async -- the SDMA engines through the HSA runtime, tickets instead of streams and events
        # (hipMemcpyAsync costs the staging thread ~20 us per message, the event records around it ~12 more)
        self.dma_h2d = False
        self.on_finish = None           # callable: a transport.Prefetcher staging one train ahead is woken here
        self._lib = L.load()            # (the staging-copy variant for this host is picked on the first host copy)

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
            # frames travel device-to-device in copy-stream order (a frame that was DMA-copied straight out of a pinned
            # transport slot never existed in the host staging buffer); labels are still host-side only
            self._join_copy_streams()
            with torch.cuda.stream(self.copy_stream):
                new.dev["obs"][:self.n].copy_(s.dev["obs"][:self.n], non_blocking=True)
            for k in new.host:
                if k != "obs":
                    m = self.n + 1 if k == "offsets" else self.n
                    new.host[k][:m].copy_(s.host[k][:m])
            self.copy_stream.synchronize()             # the old set is released when this function returns
        self.sets[self.cur] = new
        return new

    def put(self, obs, *labels, pinned=False, slot_guard=None, _raw=False):
        """Append one trajectory / rollout message ([T,...] arrays as the explorer ships them, labels in the order
        of ``fields``) and start the H2D copy of its frames.  ``pinned``: the arrays are views into page-locked memory
        (a pinned transport ring): frames whose dtype already is the device dtype are DMA-copied straight from the
        source; the call returns when that copy has landed (the caller recycles the slot right after) -- or at once, if
        the transport handed over a ``slot_guard`` (``transport.SlotGuard``): it then gets the copy's event and keeps
        the slot until the event has fired.  Otherwise the arriving arrays are never referenced after the call returns."""
        obs = np.asarray(obs)
        t = obs.shape[0]
        s = self._ensure(self.n + t, obs)
        dma = bool(self.dma_h2d and pinned and self.zero_copy_labels and not _raw)
        if self.n == 0:
            s.used_stream = False
            if s.free is not None:      # do not overwrite a set an enqueued update still reads
                if dma:
                    s.free.synchronize()        # (no stream to order against; with the in-graph tail `free` is None here)
                for cs in self.copy_streams:
                    cs.wait_event(s.free)
        cstream = self.copy_streams[self._rr % len(self.copy_streams)]
        self._rr += 1
        lo, hi = self.n, self.n + t
        n_raw = self._n_raw
        if n_raw and len(labels) == len(self.fields) - n_raw:
            labels = tuple(labels) + (None,) * n_raw        # a trajectory that brings its advantages: no GAE inputs
        if len(labels) != len(self.fields):
            raise ValueError("RolloutIngest.put: {} label arrays for fields {}".format(
                len(labels), [f[0] for f in self.fields]))
        # a rollout is either all-raw (GAE on the learner) or all with advantages: reject the odd message AS IT ARRIVES
        # (before any copy is enqueued for it), so that the rollout gathered so far stays usable
        if (self.adv_traj if _raw else self.raw_traj):
            raise ValueError("RolloutIngest: a rollout must not mix trajectories with and without advantages "
                             "(this one {} them)".format("lacks" if _raw else "brings"))
        if not _raw:
            self.adv_traj += 1
        obs_dst = s.host_np["obs"][lo:hi]
        row_bytes = obs_dst.dtype.itemsize * int(np.prod(obs_dst.shape[1:], dtype=np.int64))
        dev_ptr = s.dev["obs"].data_ptr() + lo * row_bytes
        plain = obs.dtype == obs_dst.dtype and obs.flags.c_contiguous and obs.size == obs_dst.size
        if pinned and plain and dma:
            # ... and no stream either: one asynchronous SDMA copy with a ticket
            tk = ctypes.c_uint64()
            if self._lib.xt_dma_h2d_async(ctypes.c_void_p(dev_ptr), ctypes.c_void_p(obs.ctypes.data), obs.nbytes,
                                          ctypes.byref(tk)) != 0:
                # this process cannot (no HSA runtime to be had, memory it does not know): the stream path from now on
                self.dma_h2d = dma = False
        if pinned and plain and dma:
            s.dma_ticket = int(tk.value)
            if slot_guard is not None:      # the ring keeps the slot until the copy has landed: no wait here
                slot_guard.hold(_DmaDone(self._lib, s.dma_ticket))
            else:
                _DmaDone(self._lib, s.dma_ticket).synchronize()
        elif pinned and plain:
            # DMA source = the pinned transport slot: no host copy at all; the caller recycles the slot right after.  One raw
            # hipMemcpyAsync (a torch copy_ under a stream context costs ~40 us of Python per message) and pooled events
            s.used_stream = True
            L.memcpy_async(dev_ptr, obs.ctypes.data, obs.nbytes, L.H2D, cstream)
            if slot_guard is not None:      # the ring keeps the slot until this event has fired: no wait here
                pool = getattr(self, "_guard_events", None)
                if pool is None:
                    pool = self._guard_events = []
                ev = None
                for k, cand in enumerate(pool):      # an event whose copy has landed (and whose slot was released) is free again
                    if cand.query():
                        ev = pool.pop(k)
                        break
                if ev is None:
                    ev = torch.cuda.Event()
                ev.record(cstream)
                pool.append(ev)
                slot_guard.hold(ev)
            else:
                cstream.synchronize()
        elif plain:
            s.used_stream = True
            if not _TUNED:
                staging_report()        # measure the host's copy variants once per process, on first use
            # an IMPALA message of a few MB (one 128-step Atari trajectory: 3.6 MB) ships in 1 MiB pieces, so that its H2D runs
            # under its own staging instead of behind it; everything else keeps the 4 MiB default (fewer DMA set-ups)
            # (IMPALA ingests only, n_epochs == 0: a PPO rollout's 32+ trajectories already overlap each other's H2D, and three
            # more DMA set-ups per trajectory cost that path 0.7 ms per update, measured round 6)
            ship = (1 << 20) if (self.n_epochs == 0 and obs.nbytes <= (8 << 20)) else 0
            L.check(self._lib.xt_stage_rows(ctypes.c_void_p(s.host["obs"].data_ptr() + lo * row_bytes),
                                            ctypes.c_void_p(obs.ctypes.data), obs.nbytes, ctypes.c_void_p(dev_ptr), 0, ship, -1,
                                            ctypes.c_void_p(cstream.cuda_stream)), "xt_stage_rows")
        else:       # a cast on the way in (float frames for a uint8 network, ...): as the upload path casts them
            s.used_stream = True
            np.copyto(obs_dst, obs.reshape(obs_dst.shape), casting="unsafe")
            with torch.cuda.stream(cstream):
                s.dev["obs"][lo:hi].copy_(s.host["obs"][lo:hi], non_blocking=True)
        for f, a in zip(self.fields, labels):
            if a is None:                    # filled by the caller (put_raw) or on the device (GAE outputs)
                continue
            arr = np.asarray(a)
            dst = s.host_np[f[0]][lo:hi]
            # bool -> uint8 and float64 -> float32 are the casts the reference's placeholders apply
            np.copyto(dst, arr.reshape(dst.shape), casting="unsafe" if arr.dtype == np.bool_ else "same_kind")
        self.n = hi
        return s, lo, hi

    def put_raw(self, obs, action, logp, value, reward, done, pinned=False, slot_guard=None):
        """A PPO trajectory as the explorer holds it BEFORE ``data_proc`` (xt/agent/ppo/ppo.py:77-106): value [T+1],
        reward [T], done [T] instead of adv / old_value / target_value.  The value column is staged as ``old_v`` (it is
        that column), the bootstrap value and the row range are kept per trajectory; ``gae_on_device`` then fills adv /
        target_v for the whole rollout with one kernel launch."""
        names = [f[0] for f in self.fields]
        if "boot" not in names:
            raise RuntimeError("RolloutIngest.put_raw: built without PPO_RAW_FIELDS")
        value = np.asarray(value, np.float32).reshape(-1)
        t = int(np.asarray(obs).shape[0])
        if value.shape[0] != t + 1:
            raise ValueError("put_raw: value must have T+1 = {} entries, got {}".format(t + 1, value.shape[0]))
        by_name = dict(action=action, old_logp=logp, adv=None, old_v=value[:t], target_v=None,
                       reward=np.asarray(reward, np.float64), done=np.asarray(done, bool), boot=None)
        s, lo, hi = self.put(obs, *[by_name[n] for n in names], pinned=pinned, slot_guard=slot_guard, _raw=True)
        k = self.raw_traj
        s.host_np["boot"][k] = value[t]
        s.host_np["offsets"][k] = lo
        s.host_np["offsets"][k + 1] = hi
        self.raw_traj = k + 1

    def gae_on_device(self, dev, n, gamma, lam, stream_ptr):
        """adv / target_v of the rollout just finished, on the device (C ABI xt_gae_f64_ragged; float64, bit-exact with
        the reference's numpy loop).  Call after ``finish`` on the compute stream."""
        k = self.last_raw_traj
        L.check(self._lib.xt_gae_f64_ragged(L.ptr(dev["old_v"]), L.ptr(dev["boot"]), L.ptr(dev["reward"]), L.ptr(dev["done"]),
                                            L.ptr(dev["offsets"]), L.ptr(dev["adv"]), L.ptr(dev["target_v"]), k,
                                            float(gamma), float(lam), stream_ptr), "xt_gae_f64_ragged")

    def ship_labels(self):
        """The rollout is complete: enqueue the ONE copy of its label block (and join the frame copies) now.  ``finish``
        does it itself; a ``transport.Prefetcher`` calls it from its staging thread as soon as the last message of a train
        has been staged, so that the DMA's latency is not paid between ``train()`` and the GPU's first kernel."""
        s = self.sets[self.cur]
        if s is None or self.n == 0 or getattr(s, "shipped_n", -1) == self.n:
            return
        if self.zero_copy_labels and not self.raw_traj:
            self._join_copy_streams(wait=False)
            s.shipped_n = self.n
            return
        self._join_copy_streams(wait=False)
        # the labels of the whole rollout: ONE copy (a few 10 KB)
        L.memcpy_async(s.lab_dev.data_ptr(), s.lab_host.data_ptr(), s.lab_host.numel(), L.H2D, self.copy_stream)
        if self.raw_traj:
            L.memcpy_async(s.dev["offsets"].data_ptr(), s.host["offsets"].data_ptr(), 4 * (self.raw_traj + 1), L.H2D,
                           self.copy_stream)
        s.shipped_n = self.n

    def seal(self):
        """(a ``transport.Prefetcher``'s staging thread) the last message of a rollout has been staged: ship its labels and
        record the set's copies-done event NOW, so that ``finish`` on the learner thread only switches buffer sets.  A ``put``
        after this un-seals the set (``finish`` then does both again)."""
        s = self.sets[self.cur]
        if s is None or self.n == 0:
            return
        if self.pad_channels is not None and s.dev["obs"].shape[-1] != self.pad_channels[0]:
            return                              # (the channel padding runs in finish, in front of the event)
        self.ship_labels()
        if s.used_stream or not self.zero_copy_labels:
            s.done.record(self.copy_stream)
        s.sealed_n = self.n

    def finish(self, wait_on_stream=True):
        """All trajectories are in: make the compute stream wait for the copies, return (n, device buffers) and
        switch to the other buffer set for the next rollout.  ``wait_on_stream=False``: the caller makes its stream wait for
        ``self.last.done`` itself (``xt_net_impala_train_io`` does it inside the train's C call)."""
        s = self.sets[self.cur]
        n = self.n
        if s is None or n == 0:
            raise RuntimeError("RolloutIngest.finish(): nothing was ingested")
        if self.raw_traj and self.adv_traj:       # (unreachable through put / put_raw, which reject the odd message)
            self.reset()
            raise RuntimeError("RolloutIngest.finish(): a rollout must not mix trajectories with and without advantages")
        dev = s.dev
        sealed = getattr(s, "sealed_n", -1) == n      # (seal(): labels shipped and the event recorded by the staging thread)
        s.sealed_n = -1
        if not sealed:
            self.ship_labels()
        s.shipped_n = -1
        if self.pad_channels is not None and s.dev["obs"].shape[-1] != self.pad_channels[0]:
            c_dst, fill = self.pad_channels
            src = s.dev["obs"]
            if s.dev_padded is None:
                s.dev_padded = torch.empty(tuple(src.shape[:-1]) + (c_dst,), dtype=src.dtype, device=src.device)
            rows = n * int(np.prod(src.shape[1:-1], dtype=np.int64))
            if s.dma_ticket:                    # (ticketed copies are ordered against no stream: they have to be over first)
                _DmaDone(self._lib, s.dma_ticket).synchronize()
            L.check(self._lib.xt_pad_channels(L.ptr(src), L.ptr(s.dev_padded), rows, int(src.shape[-1]), c_dst,
                                              src.element_size(), int(fill),
                                              ctypes.c_void_p(self.copy_stream.cuda_stream)), "xt_pad_channels")
            dev = dict(s.dev, obs=s.dev_padded)
        stream_copies = s.used_stream or not self.zero_copy_labels or self.raw_traj > 0 or dev is not s.dev
        if not sealed and stream_copies:
            s.done.record(self.copy_stream)
        if wait_on_stream:
            if stream_copies:
                L.current_stream(self.device).wait_event(s.done)
            if s.dma_ticket:
                _DmaDone(self._lib, s.dma_ticket).synchronize()
        s.wait_stream = stream_copies        # (wait_on_stream=False: the caller waits for `done` iff this, and for dma_ticket)
        self.cur ^= 1
        self.n = 0
        self.last = s
        self.last_raw_traj, self.raw_traj, self.adv_traj = self.raw_traj, 0, 0
        self.generation += 1
        if self.on_finish is not None:
            self.on_finish()
        return n, dev

    def _join_copy_streams(self, wait=True):
        """stream 0 waits for the other copy streams (device-side); ``wait``: the host waits for all of them too"""
        if len(self.copy_streams) > 1 and not hasattr(self, "_join_ev"):
            self._join_ev = [torch.cuda.Event() for _ in self.copy_streams[1:]]
        for i, cs in enumerate(self.copy_streams[1:]):
            self._join_ev[i].record(cs)
            self.copy_stream.wait_event(self._join_ev[i])
        if wait:
            self.copy_stream.synchronize()

    def mapped_labels(self, n):
        """device-visible addresses of the LAST finished set's label arrays inside its page-locked staging block
        (``zero_copy_labels``): name -> int, or None when the block is not mapped into the device's address space"""
        s = self.last
        base = getattr(s, "_lab_mapped", None)
        if base is None:
            got = L.host_device_ptr(s.lab_host.data_ptr())
            base = s._lab_mapped = got if got is not None else 0
        if not base:
            return None
        h0 = s.lab_host.data_ptr()
        return {name: base + (s.host[name].data_ptr() - h0) for name, _dt, _w in self.fields}

    def consumed_event(self):
        """the event that marks the last finished set as consumed (raw-handle users record it themselves: the event exists,
        i.e. has been recorded once, when this returns)"""
        if self.last.free is None:
            self.last.free = torch.cuda.Event()
            self.last.free.record(L.current_stream(self.device))
        return self.last.free

    def mark_consumed(self):
        """call after the update that reads the last finished set has been enqueued on the compute stream"""
        if self.last.free is None:
            self.last.free = torch.cuda.Event()
        self.last.free.record(L.current_stream(self.device))

    def reset(self):
        self.n = 0
        self.raw_traj = self.adv_traj = 0
