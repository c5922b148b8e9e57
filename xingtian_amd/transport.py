"""Payload codec + shared-memory channel for rollout messages (transport half).

The reference moves every message as ``pyarrow.serialize(obj).to_buffer()`` (+ ``lz4.frame`` above 1 MB) through a
plasma object store between processes of one node (zeus/common/ipc/share_by_plasma.py:49-95) and through zmq
multipart frames between nodes (zeus/common/ipc/comm_by_zmq.py:69-97; broker.py:97-119 unpacks them).  Neither
survives a modern stack: pyarrow >= 2 has no ``serialize`` and no plasma (this image ships pyarrow 25), and ``lz4``
is not installed.  This module is the plasma-free replacement for the two message kinds on the learner path --
rollout batches to the learner (``cmd: train``) and weight dicts back to the explorers:

* ``encode(ctr_info, data)`` -> one contiguous buffer: a msgpack header (control dict, python-object fields such as
  the ``reward`` / ``done`` / ``info`` lists, and for every ndarray its dtype / shape / byte range) followed by the
  raw array bytes, each 64-byte aligned.  No compression: a uint8 Atari rollout is incompressible enough that the
  reference's lz4 pass costs more than the memcpy it saves on an intra-node hop.
* ``decode(buf)`` -> ``(ctr_info, data)`` with the arrays as ZERO-COPY views into ``buf``.
* ``decode_into(buf, sink)`` -> hands the views straight to an ingest callback (``Algorithm.prepare_data``), so that a
  trajectory goes wire -> pinned staging -> HBM with exactly one host copy (``RolloutIngest.put``).
* ``RingSet`` -- fan-in: one ``ShmRing`` per explorer, drained round robin by the learner (the broker's receive loop);
  ``WeightsRing`` -- fan-out: the learner's weights to N explorers through a sequence-locked slot ring (``ShareBuf``).
* ``ShmRing`` -- a single-producer / single-consumer ring of fixed-size slots in ``multiprocessing.shared_memory``
  carrying encoded messages between an explorer-side process and the learner process: the role of plasma's
  ``put_raw_buffer`` / ``get_buffers`` pair plus its control queue, without a server process.  ``send`` / ``recv``
  keep the reference channel's ``(ctr_info, data)`` contract.  ``pin()`` (learner side) page-locks the ring with
  ``hipHostRegister``: the ingest then DMA-copies the uint8 frames to HBM straight out of the slot -- wire -> HBM with
  NO host copy on the learner side.

* ``Prefetcher`` -- learner side, asynchronous algorithms (IMPALA): a thread that owns the consumer end of a ring / ring set
  and hands every message to the algorithm's ingest AS IT ARRIVES, so the H2D of train k+1's messages runs under the GPU's
  train k; the learner loop (``recv_into(alg.prepare_data)`` x prepare_data_times -> ``train()``) is unchanged.
  ``WeightsRing.start_committer()`` -- the other end: the D2H an update enqueued into a ring slot is committed (made visible
  to the explorers) by a helper thread when it lands; the learner thread does not wait (as the reference's learner hands the
  weights object to its send queue, xt/framework/learner.py:361-374).

Plumbing only: no arithmetic; the only GPU-runtime call is the optional ``hipHostRegister`` of ``pin()``.
"""
import struct
import threading
import time
from multiprocessing import shared_memory

import msgpack
import numpy as np

MAGIC = b"XTM1"
_ALIGN = 64


def _pad(n):
    return (n + _ALIGN - 1) // _ALIGN * _ALIGN


def _plain(obj):
    """numpy scalars / bool_ inside python containers -> msgpack-able python objects"""
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def _as_typed_array(val):
    """a python list / tuple of >= 8 bools, or of ints / floats, as the array ``np.asarray`` makes of it (bool, int64, float64);
    None for anything else (nested, mixed with other types, short)"""
    if not isinstance(val, (list, tuple)) or len(val) < 8:
        return None
    kinds = set(map(type, val))
    if kinds == {bool} or kinds == {np.bool_} or kinds <= {bool, np.bool_}:
        return np.asarray(val, dtype=np.bool_)
    if kinds <= {int, float, np.float64, np.float32, np.int64, np.int32} and kinds:
        arr = np.asarray(val)
        return arr if arr.dtype != object and arr.ndim == 1 else None
    return None


def _layout(ctr_info, data, pack_lists=False):
    """-> (msgpack header bytes, byte offset of the array section, [(offset, nbytes, contiguous array)], total bytes)"""
    arrays, objects, metas = [], {}, []
    off = 0
    for key, val in data.items():
        if pack_lists:
            # SENDER-side option: per-step python lists (the agents ship done / reward that way, xt/agent/*) travel as typed
            # arrays -- the learner's np.asarray(list) of every message (hundreds of boxed scalars through msgpack, ~20 us of
            # its serial staging time per message) becomes a zero-copy view; 512 explorers pay 6 us each in parallel instead
            packed = _as_typed_array(val)
            if packed is not None:
                val = packed
        if isinstance(val, np.ndarray) and val.dtype != object:
            arr = np.ascontiguousarray(val)            # (promotes 0-d to 1-d: the shape on the wire is the original one)
            metas.append([key, arr.dtype.str, list(val.shape), off, arr.nbytes])
            arrays.append(arr)
            off += _pad(arr.nbytes)
        else:
            objects[key] = _plain(val)
    header = msgpack.packb({"ctr": _plain(ctr_info), "obj": objects, "arr": metas, "order": list(data.keys())},
                           use_bin_type=True)
    base = _pad(8 + len(header))
    return header, base, [(m[3], m[4], a) for m, a in zip(metas, arrays)], base + off


def _write(view, header, base, arrays):
    view[0:4] = MAGIC
    struct.pack_into("<I", view, 4, len(header))
    view[8:8 + len(header)] = header
    for aoff, nbytes, arr in arrays:
        if nbytes:
            view[base + aoff:base + aoff + nbytes] = arr.reshape(-1).view(np.uint8)


def encode(ctr_info, data, pack_lists=False):
    """``data``: dict field -> ndarray | python object (the reference's train_data / weights dict).  Returns a
    ``bytearray``: MAGIC | u32 header length | msgpack header | padding | array bytes (64-byte aligned each).
    ``pack_lists``: homogeneous numeric / bool lists travel as arrays (the receiver gets an ndarray where a list was sent)."""
    header, base, arrays, total = _layout(ctr_info, data, pack_lists)
    buf = bytearray(total)
    _write(memoryview(buf), header, base, arrays)
    return buf


def encode_into(view, ctr_info, data, pack_lists=False):
    """Encode straight into a writable buffer (a shared-memory slot): one copy per array, no intermediate message.
    Returns the encoded length; ValueError if the buffer is too small."""
    header, base, arrays, total = _layout(ctr_info, data, pack_lists)
    if total > len(view):
        raise ValueError("message of {} bytes exceeds the {}-byte buffer".format(total, len(view)))
    _write(view, header, base, arrays)
    return total


def _header(buf):
    view = memoryview(buf)
    if bytes(view[0:4]) != MAGIC:
        raise ValueError("transport.decode: not an XTM1 message")
    hlen = struct.unpack_from("<I", view, 4)[0]
    head = msgpack.unpackb(bytes(view[8:8 + hlen]), raw=False, strict_map_key=False)
    return view, head, _pad(8 + hlen)


def decode(buf):
    """-> (ctr_info, data); ndarray fields are zero-copy (read-only when ``buf`` is) views into ``buf``."""
    view, head, base = _header(buf)
    fields = dict(head["obj"])
    for key, dt, shape, off, nbytes in head["arr"]:
        fields[key] = np.frombuffer(view[base + off:base + off + nbytes], dtype=np.dtype(dt)).reshape(shape)
    return head["ctr"], {k: fields[k] for k in head["order"]}


def decode_into(buf, sink):
    """Decode and hand the message to ``sink(data, ctr_info=...)`` (``Algorithm.prepare_data``'s signature,
    xt/framework/learner.py:313) while the views are still backed by ``buf``: the sink copies what it keeps (the
    streaming ingest copies into pinned staging), after which the slot can be released."""
    ctr, data = decode(buf)
    sink(data, ctr_info=ctr)
    return ctr


class SlotGuard(object):
    """Handed to the sink of a PINNED ring in ``ctr_info["_slot_guard"]``: a sink that starts an asynchronous copy out of
    the slot calls ``hold(event)`` with an object offering ``query()`` / ``synchronize()`` (a ``torch.cuda.Event``)
    instead of waiting for the copy; the ring releases the slot once the event has completed."""

    def __init__(self):
        self.event = None

    def hold(self, event):
        self.event = event


class ShmRing(object):
    """Single-producer / single-consumer ring of ``slots`` x ``slot_bytes`` in POSIX shared memory.

    Layout: 64-byte control block {head u64 (next slot to write), tail u64 (next slot to read)} + per-slot u64
    payload length + the slots.  Only the producer writes ``head`` / lengths / slot bytes, only the consumer writes
    ``tail``; 8-byte aligned stores are atomic on x86-64 and the payload is complete before ``head`` advances."""

    def __init__(self, name=None, slots=8, slot_bytes=8 << 20, create=True):
        self.slots, self.slot_bytes = int(slots), int(_pad(slot_bytes))
        size = _ALIGN + 8 * self.slots + self.slots * self.slot_bytes
        if create:
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.shm.buf[:_ALIGN + 8 * self.slots] = bytes(_ALIGN + 8 * self.slots)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.owner = bool(create)
        self.name = self.shm.name
        self._ctl = np.ndarray((2,), dtype=np.uint64, buffer=self.shm.buf, offset=0)
        self._len = np.ndarray((self.slots,), dtype=np.uint64, buffer=self.shm.buf, offset=_ALIGN)
        self._base = _ALIGN + 8 * self.slots
        self.pinned = False
        self._held = []            # consumer side: guards of delivered messages whose slot is not released yet
        self._taken = 0            # ... and their number (tail has not advanced past them)

    def pin(self):
        """Learner side, optional: page-lock the ring with the HIP runtime (``hipHostRegister``) so that the arrays
        of a received message can be DMA-copied to HBM straight out of the slot -- no staging copy at all
        (``recv_into`` then tells the sink that its views are pinned).  Needs a GPU process; returns True on success."""
        import ctypes
        if self.pinned:
            return True
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            self._hip = hip
            probe = np.frombuffer(self.shm.buf, dtype=np.uint8)
            self._pin_addr = int(probe.ctypes.data)
            del probe
            hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
            hip.hipHostRegister.restype = ctypes.c_int
            rc = hip.hipHostRegister(ctypes.c_void_p(self._pin_addr), ctypes.c_size_t(self.shm.size), 0)
        except OSError:
            return False
        self.pinned = (rc == 0)
        return self.pinned

    # ---- producer side
    def send(self, ctr_info, data, block=True, timeout=None, pack_lists=False):
        """Encode straight into the next free slot (ShareByPlasma.send / CommByZmq.send contract)."""
        msg = encode(ctr_info, data, pack_lists)
        return self.send_bytes(msg, block=block, timeout=timeout)

    def send_bytes(self, msg, block=True, timeout=None):
        n = len(msg)
        if n > self.slot_bytes:
            raise ValueError("message of {} bytes exceeds the slot size {}".format(n, self.slot_bytes))
        t0 = time.monotonic()
        while int(self._ctl[0]) - int(self._ctl[1]) >= self.slots:           # ring full
            if not block or (timeout is not None and time.monotonic() - t0 > timeout):
                return False
            time.sleep(0.0002)
        slot = int(self._ctl[0]) % self.slots
        off = self._base + slot * self.slot_bytes
        self.shm.buf[off:off + n] = msg
        self._len[slot] = n
        self._ctl[0] = int(self._ctl[0]) + 1
        return True

    # ---- consumer side
    def recv_view(self, block=True, timeout=None):
        """-> memoryview of the oldest unread message (valid until ``release``) or None."""
        t0 = time.monotonic()
        while int(self._ctl[0]) == int(self._ctl[1]) + self._taken:
            self._reap()                  # slots held for a deferred copy may be what the producer is waiting for
            if int(self._ctl[0]) != int(self._ctl[1]) + self._taken:
                break
            if not block or (timeout is not None and time.monotonic() - t0 > timeout):
                return None
            time.sleep(0.0002)
        slot = (int(self._ctl[1]) + self._taken) % self.slots
        off = self._base + slot * self.slot_bytes
        return self.shm.buf[off:off + int(self._len[slot])]

    def release(self):
        self._ctl[1] = int(self._ctl[1]) + 1

    def _done_with_slot(self, guard=None):
        """the message just delivered needs its slot no longer (or only until ``guard.event``): slots are released in
        arrival order, so it queues behind earlier messages whose deferred copies are still pending"""
        if (guard is None or guard.event is None) and not self._held:
            self.release()
        else:
            self._held.append(guard if guard is not None else SlotGuard())
            self._taken += 1

    def recv(self, block=True, timeout=None):
        """-> (ctr_info, data) with the arrays COPIED out of the slot (the reference channel's contract)."""
        view = self.recv_view(block, timeout)
        if view is None:
            return None
        ctr, data = decode(view)
        data = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in data.items()}
        del view
        self._done_with_slot()
        return ctr, data

    def recv_into(self, sink, block=True, timeout=None):
        """Zero-copy receive: ``sink(data, ctr_info=...)`` sees views into the slot; the slot is released after it
        returns (``decode_into``).  On a pinned ring the sink may start a DMA straight out of the slot and hand the
        ring an event instead of waiting for it (``ctr_info["_slot_guard"].hold(event)``, see ``SlotGuard``): the slot
        is then released, in order, once that copy has landed -- up to ``slots - 1`` messages' copies stay in flight
        while the learner already decodes the next message."""
        self._reap()
        view = self.recv_view(block, timeout)
        if view is None:
            return None
        if self.pinned:
            if self._taken >= self.slots - 1:     # at most slots - 1 deferred copies in flight: wait for the oldest
                g = self._held[0]
                if g.event is not None:
                    g.event.synchronize()
                self._reap()
            ctr, data = decode(view)
            guard = SlotGuard()
            sink(data, ctr_info=dict(ctr, _pinned_views=True, _slot_guard=guard))   # the sink may DMA straight out of the slot
            del data
            del view
            self._done_with_slot(guard)           # (released in order by _reap() when a copy is still pending)
            return ctr
        ctr = decode_into(view, sink)
        del view
        self._done_with_slot()
        return ctr

    def _reap(self, wait=False):
        """release the slots whose deferred copies have completed (oldest first)"""
        while self._held:
            g = self._held[0]
            if g.event is not None:
                if wait:
                    g.event.synchronize()
                elif not g.event.query():
                    break
            self._held.pop(0)
            self._taken -= 1
            self.release()

    def pending(self):
        """messages sent and not yet delivered to the consumer"""
        return int(self._ctl[0]) - int(self._ctl[1]) - self._taken

    def reap(self):
        """release the slots whose deferred copies have completed; returns the number still held"""
        self._reap()
        return len(self._held)

    def drain(self):
        """wait for every deferred copy and release its slot"""
        self._reap(wait=True)

    def close(self):
        if self._held:
            self._reap(wait=True)
        self._ctl = self._len = None
        if self.pinned:
            self._hip.hipHostUnregister.argtypes = [__import__("ctypes").c_void_p]
            self._hip.hipHostUnregister(__import__("ctypes").c_void_p(self._pin_addr))
            self.pinned = False
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass


class RingSet(object):
    """Fan-in of many explorers into one learner: ONE single-producer ring per explorer (so producers never contend and
    ``ShmRing``'s lock-free single-writer protocol holds) and a learner-side poller that drains them round robin -- the
    role of the broker's receive loop, which forwards every explorer's ``cmd: train`` message into the one queue the
    learner's ``prepare_data`` loop reads (xt/framework/broker.py:97-119, xt/framework/learner.py:306-313).

    * at most one message per ring per sweep and the sweep resumes behind the ring served last: a fast explorer cannot
      starve the others;
    * back pressure is per explorer: a full ring blocks (or times out) only its own producer;
    * the sink sees ``ctr_info["explorer_id"]`` = ring index, the tag the broker derives from the socket identity
      and ``FIFODistPolicy`` uses to route the new weights back (xt/algorithm/alg_utils.py:FIFODistPolicy).

    Learner side: ``RingSet(n)`` creates the rings; explorer i attaches with ``RingSet.attach(names[i], ...)``."""

    def __init__(self, n_rings, slots=4, slot_bytes=8 << 20):
        self.rings = [ShmRing(slots=slots, slot_bytes=slot_bytes) for _ in range(int(n_rings))]
        self.names = [r.name for r in self.rings]
        self.slots, self.slot_bytes = int(slots), self.rings[0].slot_bytes if self.rings else int(slot_bytes)
        self._next = 0
        self.served = [0] * len(self.rings)

    @staticmethod
    def attach(name, slots=4, slot_bytes=8 << 20):
        """Explorer side: open the ring the learner created for this explorer."""
        return ShmRing(name=name, create=False, slots=slots, slot_bytes=slot_bytes)

    def pin(self):
        """Page-lock every ring (``ShmRing.pin``): frames are then DMA-copied to HBM straight out of the slots."""
        return all([r.pin() for r in self.rings])

    def pending(self):
        return sum(r.pending() for r in self.rings)

    def poll_into(self, sink, max_msgs=None):
        """One non-blocking round-robin sweep: deliver at most one waiting message per ring to
        ``sink(data, ctr_info=...)`` (zero-copy views, released after the sink returns).  Returns the number
        delivered (0: nothing was waiting)."""
        n = len(self.rings)
        got = 0
        for k in range(n):
            if max_msgs is not None and got >= max_msgs:
                break
            i = (self._next + k) % n
            ring = self.rings[i]
            # slots held behind deferred copies (pinned rings) are released HERE as their events fire: pending() does not
            # count them, so a ring whose slots are all held would otherwise never be visited again and its producer
            # would wait on a full ring forever
            ring.reap()
            if ring.pending() == 0:
                continue
            tag = lambda data, ctr_info=None, _i=i: sink(data, ctr_info=dict(ctr_info or {}, explorer_id=_i))
            if ring.recv_into(tag, block=False) is not None:
                got += 1
                self.served[i] += 1
                last = i
        if got:
            self._next = (last + 1) % n
        return got

    def recv_many_into(self, sink, count, timeout=None):
        """Block until ``count`` messages have been delivered (the learner's ``prepare_data_times`` loop,
        learner.py:306-313) or ``timeout`` seconds passed; returns the number delivered."""
        t0 = time.monotonic()
        got = 0
        while got < count:
            k = self.poll_into(sink, max_msgs=count - got)
            got += k
            if k == 0:
                if timeout is not None and time.monotonic() - t0 > timeout:
                    break
                for r in self.rings:          # idle: give back the slots whose copies have landed meanwhile
                    r.reap()
                time.sleep(0.0002)
        return got

    def close(self):
        for r in self.rings:
            r.close()


class WeightsRing(object):
    """Fan-out of the learner's weights to any number of explorers: one writer, N readers, no server process, no
    per-reader queue -- the role of the reference's ``ShareBuf`` (one plasma object per publish that every explorer
    fetches by id and the learner reference-counts, zeus/common/ipc/share_buffer.py:131-168) for the
    ``get_weights()`` hand-over (xt/framework/learner.py:361-363).

    Layout: 64-byte control block {latest u64 = sequence number of the newest COMPLETE publish} + ``slots`` x
    (64-byte slot header {seq u64, nbytes u64} + payload).  Publish k goes to slot k % slots: the writer first sets
    the slot's seq to 0 (invalid), writes the payload (``encode_into``: the packed parameter arrays + their name
    table), then seq = k, then latest = k.  A reader copies the payload of slot latest % slots and accepts it only if
    the slot's seq was k before AND after the copy (a sequence lock: a writer that laps the reader is detected, the
    reader retries on the newer publish).  x86-64 total store order + one C call per store keep the order."""

    def __init__(self, name=None, slot_bytes=8 << 20, slots=3, create=True):
        self.slots, self.slot_bytes = int(slots), int(_pad(slot_bytes))
        size = _ALIGN + self.slots * (_ALIGN + self.slot_bytes)
        if create:
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.shm.buf[:_ALIGN] = bytes(_ALIGN)
            for i in range(self.slots):
                o = _ALIGN + i * (_ALIGN + self.slot_bytes)
                self.shm.buf[o:o + _ALIGN] = bytes(_ALIGN)
        else:
            self.shm = shared_memory.SharedMemory(name=name)
        self.owner = bool(create)
        self.name = self.shm.name
        self._latest = np.ndarray((1,), dtype=np.uint64, buffer=self.shm.buf, offset=0)
        self._hdr = [np.ndarray((2,), dtype=np.uint64, buffer=self.shm.buf, offset=_ALIGN + i * (_ALIGN + self.slot_bytes))
                     for i in range(self.slots)]
        self._seen = 0
        self.pinned = False
        self._pending = []          # begun, uncommitted packed publishes, oldest first: (seq, slot, bytes, event)
        self._wlock = threading.Condition(threading.RLock())     # writer state (_pending, headers, latest)
        self.async_commit = False   # start_committer(): begun publishes are committed by a helper thread as their copies land
        self._committer = None

    def _payload_offset(self, i):
        return _ALIGN + i * (_ALIGN + self.slot_bytes) + _ALIGN

    def _payload(self, i):
        o = self._payload_offset(i)
        return self.shm.buf[o:o + self.slot_bytes]

    # ---- writer (the learner)
    def publish(self, weights, ctr_info=None):
        """Write the name -> ndarray dict (``get_weights()``: views into the pinned D2H block are fine, every array
        is copied exactly once, into the slot).  Returns the sequence number of this publish."""
        if self._pending:
            # a packed publish is begun and not committed: its DMA targets slot (latest + 1) % slots, the one a dict publish
            # would write now
            raise RuntimeError("WeightsRing.publish: {} packed publish(es) begun and not committed; commit_flat_publish() "
                               "or retarget_flat_publish() first".format(len(self._pending)))
        k = int(self._latest[0]) + 1
        i = k % self.slots
        self._hdr[i][0] = 0
        view = self._payload(i)
        n = encode_into(view, dict(ctr_info or {}, cmd="weights", seq=k), weights)
        del view
        self._hdr[i][1] = n
        self._hdr[i][0] = k
        self._latest[0] = k
        return k

    def pin(self):
        """Learner side, optional: page-lock the ring (``hipHostRegister``) so that ``publish_flat_from_device`` can DMA
        the parameter block from HBM straight into the slot.  Needs a GPU process; returns True on success."""
        import ctypes
        if getattr(self, "pinned", False):
            return True
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            self._hip = hip
            probe = np.frombuffer(self.shm.buf, dtype=np.uint8)
            self._pin_addr = int(probe.ctypes.data)
            del probe
            hip.hipHostRegister.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
            hip.hipHostRegister.restype = ctypes.c_int
            rc = hip.hipHostRegister(ctypes.c_void_p(self._pin_addr), ctypes.c_size_t(self.shm.size), 0)
        except OSError:
            return False
        self.pinned = (rc == 0)
        return self.pinned

    def _reserve_flat(self, spec, nbytes, ctr_info):
        """Claim the next slot for a packed publish of ``nbytes`` bytes of ``spec``'s flat parameter buffer and write the
        message header (the name -> (offset, shape, storage shape) table is packed once per spec; only the control dict
        changes per publish).  -> (sequence number, slot, byte offset of the array section inside the slot's payload)."""
        lay = getattr(self, "_flat_layout", None)
        if lay is None or lay[0] is not spec:
            table = [[name, int(off), [int(d) for d in shape],
                      [int(d) for d in spec.store_shape[name]] if name in getattr(spec, "store_shape", {}) else None]
                     for name, (off, shape) in spec.names.items()]
            rest = msgpack.packb({"obj": {"__layout__": table}, "arr": [["__flat__", "<f4", [nbytes // 4], 0, nbytes]],
                                  "order": ["__layout__", "__flat__"]}, use_bin_type=True)
            assert rest[0] == 0x83              # fixmap of three entries: spliced behind the per-publish "ctr" entry below
            lay = self._flat_layout = (spec, rest[1:], nbytes)
        if lay[2] != nbytes:
            raise ValueError("WeightsRing: the flat buffer of this spec has {} bytes, not {}".format(lay[2], nbytes))
        if len(self._pending) >= self.slots - 1:
            if not self.async_commit:
                raise RuntimeError("WeightsRing: {} publishes begun and not committed (slots = {})".format(len(self._pending), self.slots))
            with self._wlock:           # the committer thread drains them as their copies land: wait for a free slot
                while len(self._pending) >= self.slots - 1:
                    self._wlock.wait(0.001)
        k = int(self._latest[0]) + 1 + len(self._pending)
        i = k % self.slots
        self._hdr[i][0] = 0
        # the header differs from publish to publish in the sequence number only: packed once per control dict with the number
        # as a fixed-width uint64 (0xcf + 8 bytes, big endian), patched per publish (msgpack.packb of the dict: ~6 us of the
        # learner thread's time between two trains)
        tmpl = getattr(self, "_hdr_tmpl", None)
        if tmpl is None or tmpl[0] is not lay or tmpl[1] != (ctr_info or {}):
            mark = 0x7E5A3C1F2D4B6978
            packed = b"\x84" + msgpack.packb("ctr") + msgpack.packb(_plain(dict(ctr_info or {}, cmd="weights", seq=mark)),
                                                                    use_bin_type=True) + lay[1]
            pos = packed.find(b"\xcf" + struct.pack(">Q", mark))
            assert pos >= 0 and packed.find(b"\xcf" + struct.pack(">Q", mark), pos + 1) < 0
            tmpl = self._hdr_tmpl = (lay, dict(ctr_info or {}), packed[:pos + 1], packed[pos + 9:])
        header = tmpl[2] + struct.pack(">Q", k) + tmpl[3]
        base = _pad(8 + len(header))
        if base + nbytes > self.slot_bytes:
            raise ValueError("weights of {} bytes exceed the {}-byte slot".format(base + nbytes, self.slot_bytes))
        view = self._payload(i)
        view[0:4] = MAGIC
        struct.pack_into("<I", view, 4, len(header))
        view[8:8 + len(header)] = header
        del view
        return k, i, base

    def publish_flat_host(self, flat, spec, ctr_info=None):
        """Packed publish from a HOST array (the flat float32 parameter buffer of ``spec``, e.g. a CPU replica's): one
        copy into the slot, same message format as ``publish_flat_from_device``."""
        flat = np.ascontiguousarray(flat, np.float32).reshape(-1)
        k, i, base = self._reserve_flat(spec, flat.nbytes, ctr_info)
        view = self._payload(i)
        view[base:base + flat.nbytes] = flat.view(np.uint8)
        del view
        self._hdr[i][1] = base + flat.nbytes
        self._hdr[i][0] = k
        self._latest[0] = k
        return k

    def begin_flat_publish(self, net, ctr_info=None):
        """First half of a packed publish: write the message header into the next slot and ENQUEUE the device-to-host
        copy of the learner network's flat parameter block straight into the (pinned) slot, ordered behind everything
        already enqueued on the current stream (the update whose result it publishes).  Returns at once; the slot stays
        invalid (seq 0) until ``commit_flat_publish``.  Up to ``slots - 2`` publishes may be begun ahead of their commit
        (``publish_weights(lag=1)``: the weights handed out are one update old, nothing waits)."""
        import torch
        nbytes = int(net.params.numel()) * 4
        with self._wlock:
            if getattr(self, "_commit_error", None) is not None:
                raise RuntimeError("WeightsRing: the committer thread failed") from self._commit_error
            return self._begin_flat_publish_locked(net, ctr_info, nbytes, in_stream=self.async_commit)

    def _begin_flat_publish_locked(self, net, ctr_info, nbytes, in_stream=False):
        import torch
        k, i, base = self._reserve_flat(net.spec, nbytes, ctr_info)
        st = getattr(self, "_d2h", None)
        if st is None:
            st = self._d2h = (torch.cuda.Stream(device=net.device), torch.cuda.Event(),
                              [torch.cuda.Event() for _ in range(self.slots)])
        side, ready, dones = st
        done = dones[i]
        from xingtian_amd import lib as L
        if in_stream:
            # asynchronous commit: the D2H goes on the CURRENT stream, in order behind the update -- the next update cannot
            # tear it, and nothing sits in a DMA queue blocked on a dependency (a D2H enqueued up front on a side stream held
            # up the next message's H2D; a device-side snapshot + a D2H enqueued later by the committer thread serialised with
            # the learner thread's next launch: both measured, round 6).  Nobody on the host waits for it.
            cur = L.current_stream(net.device)
            L.memcpy_async(self._pin_addr + self._payload_offset(i) + base, net.params.data_ptr(), nbytes, L.D2H, cur)
            done.record(cur)
        else:
            ready.record(L.current_stream(net.device))
            side.wait_event(ready)
            L.memcpy_async(self._pin_addr + self._payload_offset(i) + base, net.params.data_ptr(), nbytes, L.D2H, side)
            done.record(side)
        self._pending.append((k, i, base + nbytes, done))
        self._last_begun = k
        self._wlock.notify_all()
        return k

    def publish_reserve(self, net, ctr_info=None):
        """asynchronous commit, split form (``xt_net_impala_train_io``): claim the next slot and write its header; -> ticket
        (seq, slot, total bytes, HOST address of the slot's array section, raw handle of the slot's event, event).  The caller
        enqueues the D2H of the parameter block to that address IN STREAM ORDER behind its update, records the event, then
        calls ``publish_enqueued(ticket)``."""
        import torch
        from xingtian_amd import lib as L
        spare = getattr(self, "_spare_ticket", None)
        if spare is not None:
            # reserved ahead (publish_prereserve, while the device ran the previous train): still the next slot in line?
            self._spare_ticket = None
            with self._wlock:
                fresh = spare[0] == int(self._latest[0]) + 1 + len(self._pending) and getattr(self, "_commit_error", None) is None
            if fresh and spare[6] is net and spare[7] == (ctr_info or {}):
                return spare[:6]
        nbytes = int(net.params.numel()) * 4
        with self._wlock:
            if getattr(self, "_commit_error", None) is not None:
                raise RuntimeError("WeightsRing: the committer thread failed") from self._commit_error
            k, i, base = self._reserve_flat(net.spec, nbytes, ctr_info)
            st = getattr(self, "_d2h", None)
            if st is None:
                st = self._d2h = (torch.cuda.Stream(device=net.device), torch.cuda.Event(),
                                  [torch.cuda.Event() for _ in range(self.slots)])
            if not getattr(self, "_d2h_primed", False):
                cur = L.current_stream(net.device)
                for ev in st[2]:
                    ev.record(cur)                   # (a torch event gets its handle at its first record)
                self._d2h_primed = True
            done = st[2][i]
            return (k, i, base + nbytes, self._pin_addr + self._payload_offset(i) + base, done.cuda_event, done)

    def publish_prereserve(self, net, ctr_info=None):
        """reserve the NEXT publish's slot and write its header now (the learner thread calls this while the device runs the
        train it has just launched); the next ``publish_reserve`` with the same net and control dict returns this ticket.  An
        unused reservation costs nothing: the slot it marked is the oldest one, and the next reserve claims it again."""
        if len(self._pending) >= self.slots - 1:
            return None                         # (would have to wait for the committer: not on this thread's time)
        self._spare_ticket = None
        t = self.publish_reserve(net, ctr_info)
        self._spare_ticket = t + (net, dict(ctr_info or {}))
        return t

    def publish_enqueued(self, ticket):
        k, i, total, _addr, _raw, done = ticket[:6]
        with self._wlock:
            self._pending.append((k, i, total, done))
            self._last_begun = k
            self._wlock.notify_all()
        return k

    def commit_flat_publish(self):
        """Second half: wait for the OLDEST begun copy and make that publish visible to the readers.  Returns its
        sequence number."""
        with self._wlock:
            k, i, total, done = self._pending[0]
        done.synchronize()
        with self._wlock:
            if self._pending and self._pending[0][0] == k:
                self._pending.pop(0)
                self._hdr[i][1] = total
                self._hdr[i][0] = k
                self._latest[0] = k
                self._wlock.notify_all()
        return k

    def start_committer(self):
        """ASYNCHRONOUS commit (asynchronous algorithms): from now on a begun packed publish -- the D2H the update itself
        enqueued into the ring slot IN STREAM ORDER (``HipActorCritic.attach_weights_ring`` + ``snapshot_weights_async`` /
        ``xt_net_impala_train_io``) -- is made visible to the readers by a helper thread as soon as its copy has landed; ``publish_weights(ring)`` returns the sequence
        number it WILL carry without waiting.  The weights handed out are exactly those of the train that published them (no
        lag in content); only the learner thread does not sit through the D2H -- as the reference's learner hands the
        weights object to its send queue and goes on (xt/framework/learner.py:361-374; zeus/common/ipc/share_buffer.py).
        ``drain()`` waits until everything begun is visible."""
        if self._committer is not None:
            return self
        if not self.pinned:
            raise RuntimeError("WeightsRing.start_committer: the ring must be page-locked first (pin()): the committer DMA-copies "
                               "the device-side snapshot straight into the slot")
        self.async_commit = True
        self._stop_committer = False
        self._commit_error = None

        def run():
            while True:
                with self._wlock:
                    while not self._pending and not self._stop_committer:
                        self._wlock.wait(0.005)
                    if self._stop_committer and not self._pending:
                        return
                try:
                    self.commit_flat_publish()
                except Exception as exc:       # noqa: BLE001 -- surfaces in the learner thread's next publish
                    with self._wlock:
                        self._commit_error = exc
                        self._pending.clear()
                        self._wlock.notify_all()
                    return

        self._committer = threading.Thread(target=run, name="xt-weights-commit", daemon=True)
        self._committer.start()
        return self

    def drain(self, timeout=5.0):
        """wait until every begun publish is visible to the readers; -> latest sequence number"""
        t0 = time.monotonic()
        with self._wlock:
            while self._pending and time.monotonic() - t0 < timeout:
                if self._committer is None:
                    break
                self._wlock.wait(0.001)
        while self._pending and self._committer is None:
            self.commit_flat_publish()
        return int(self._latest[0])

    def retarget_flat_publish(self):
        """Drop the newest begun, uncommitted publish (an update whose weights are never handed out: its slot is reused)."""
        if self._pending:
            self._pending.pop()

    def publish_flat_from_device(self, net, ctr_info=None):
        """Publish the learner network's packed parameter block with ONE device-to-host copy straight into the (pinned)
        slot: the message carries the flat float32 buffer plus the name -> (offset, shape) table, ``fetch`` rebuilds
        the name-keyed dict on the reader side.  ``net``: a ``HipActorCritic`` (``params``, ``spec``)."""
        self.begin_flat_publish(net, ctr_info)
        return self.commit_flat_publish()

    # ---- readers (explorers)
    def latest(self):
        return int(self._latest[0])

    def fetch(self, newer_than=None, retries=64):
        """-> (seq, ctr_info, weights) of the newest complete publish with private copies of the arrays, or None when
        nothing newer than ``newer_than`` (default: the last one this reader fetched) has been published."""
        floor = self._seen if newer_than is None else int(newer_than)
        for _ in range(retries):
            k = int(self._latest[0])
            if k == 0 or k <= floor:
                return None
            i = k % self.slots
            if int(self._hdr[i][0]) != k:
                continue                              # the writer already recycles this slot: a newer publish is coming
            n = int(self._hdr[i][1])
            view = self._payload(i)
            blob = bytes(view[:n])                    # the one copy out of shared memory
            del view
            if int(self._hdr[i][0]) != k:
                continue                              # torn: the writer lapped us during the copy
            ctr, data = decode(blob)
            if "__flat__" in data:                    # packed form (publish_flat_from_device): rebuild the name-keyed dict
                flat, out = data["__flat__"], {}
                for name, off, shape, st in data["__layout__"]:
                    if st:
                        v = flat[off:off + int(np.prod(st))].reshape(st)[tuple(slice(0, d) for d in shape)]
                    else:
                        v = flat[off:off + int(np.prod(shape, dtype=np.int64))].reshape(shape)
                    out[name] = v.copy()
                data = out
            else:
                data = {name: (v.copy() if isinstance(v, np.ndarray) else v) for name, v in data.items()}
            self._seen = k
            return k, ctr, data
        raise RuntimeError("WeightsRing.fetch: no stable publish after {} attempts".format(retries))

    def close(self):
        if getattr(self, "_committer", None) is not None:
            self._stop_committer = True
            with self._wlock:
                self._wlock.notify_all()
            self._committer.join(timeout=5.0)
            self._committer = None
        # begun-but-uncommitted D2H copies may still be in flight INTO the slots: wait for them before the pages are
        # un-registered and the segment unlinked
        for pend in getattr(self, "_pending", None) or []:
            try:
                pend[3].synchronize()
            except Exception:       # noqa: BLE001 -- closing must not raise over a dead context
                pass
        self._pending = []
        self._latest = None
        self._hdr = None
        if getattr(self, "pinned", False):
            import ctypes
            self._hip.hipHostUnregister.argtypes = [ctypes.c_void_p]
            self._hip.hipHostUnregister(ctypes.c_void_p(self._pin_addr))
            self.pinned = False
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except (BufferError, FileNotFoundError):
            pass


class _IdleGate(object):
    """``threading.Event`` of the learner thread's idleness (``set`` while it is inside the train's C calls or waits for a
    message) with STRICT ALTERNATION (``strict``): ``clear()`` -- the learner is about to run Python again -- first lets a
    message that is being staged finish, instead of fighting it for the interpreter: two Python threads that both issue a
    dozen short runtime calls hand the GIL back and forth at every one of them (round 6: the same loop ran 3.2-4.0 M
    env-frames/s when the two did not collide and 2.0-2.2 M when they did).  OFF by default: measured same-box (4 x 2 runs)
    2.81-3.57 M strict against 3.12-3.73 M without -- the wait it adds to the learner thread costs what the collisions do."""

    def __init__(self, strict=False):
        self._idle = threading.Event()
        self._quiet = threading.Event()       # SET while no message is being staged
        self._quiet.set()
        self.strict = bool(strict)

    def set(self):
        self._idle.set()

    def is_set(self):
        return self._idle.is_set()

    def wait(self, timeout=None):
        return self._idle.wait(timeout)

    def clear(self, wait_s=0.002):
        self._idle.clear()
        if self.strict and not self._quiet.is_set():
            self._quiet.wait(wait_s)          # (bounded: a stuck staging thread surfaces through its own error path)

    # ---- staging thread
    def begin(self):
        """-> may a message be staged now?  (announce first, then look again: either the learner's clear() sees the
        announcement and waits, or this sees the cleared flag and backs off)"""
        self._quiet.clear()
        if self._idle.is_set():
            return True
        self._quiet.set()
        return False

    def end(self):
        self._quiet.set()


class Prefetcher(object):
    """Learner side of an ASYNCHRONOUS algorithm (IMPALA: explorers never wait for weights, the next rollout message is
    usually in the ring while the GPU still runs the current train).  The reference's learner thread receives and
    ``prepare_data``-s the messages of train k+1 only after ``train()`` k has returned (xt/framework/learner.py:306-348): the
    H2D of 3.6 MB of frames then sits between two trains.  Here a thread owns the consumer end of ``source`` (a ``ShmRing`` or
    ``RingSet``) and hands every message to ``alg.stage_message`` AS IT ARRIVES -- decode, pinned staging / DMA straight out
    of a pinned slot, asynchronous H2D into the ingest's OTHER buffer set -- at most one train ahead: before it stages the
    first message of train k+1 it waits until the learner has taken over train k's buffer set (``RolloutIngest.finish``).

    The learner loop keeps its shape and its semantics: ``pf.recv_into(alg.prepare_data)`` x ``prepare_data_times`` hands
    ``prepare_data`` a token (the control dict + ``{"_prefetched": n_rows}``) for a message whose data is already on its
    way to HBM, in arrival order; ``alg.train()`` trains exactly the messages it would have been handed.  What changes is
    WHEN the copy happens, not what is trained or published."""

    def __init__(self, source, alg, group=None, poll_s=0.0002, gate=True, strict=False, inline=None):
        if not hasattr(alg, "stage_message"):
            raise TypeError("Prefetcher: {} has no stage_message (only streaming-ingest algorithms can be prefetched)".format(
                type(alg).__name__))
        self.source, self.alg = source, alg
        self.group = int(group or alg.prepare_data_times)
        self._poll = float(poll_s)
        self._tokens = []                   # staged, not yet handed to prepare_data: control dicts in arrival order
        self._cv = threading.Condition()
        # SET while the learner thread is blocked (waiting for the GPU inside train(), or for a message here): the staging
        # thread does its Python work then.  Two Python threads that both issue dozens of short runtime calls per train hand
        # the GIL back and forth at every one of them: measured round 6, every call of the learner's train() 2-3x slower
        # (graph launch 30 -> 78 us, weight snapshot 30 -> 86 us) and the prefetched loop no faster than the blocking one
        self.learner_idle = _IdleGate(strict)
        self.learner_idle.set()
        self.gate = bool(gate)              # False: the staging thread runs whenever a message is there (same-box A/B,
                                            # round 6: breakout_impala 1.3-1.6 M ungated vs 1.9-2.0 M gated, pong 10-11 vs 12-13 M)
        self._staged = 0                    # messages staged so far
        self._error = None
        self._stop = False
        self._ingest_gen = alg.staged_generation
        self._ingest_gen()                  # (creates the ingest on THIS thread, before the staging thread touches it)
        # INLINE: no thread at all -- the learner thread stages the next train's messages itself, between two looks at the
        # loss while the device trains (``HipActorCritic.impala_wait_loss`` calls ``pump_once``) and whenever it waits for a
        # message.  One interpreter thread: what two threads lose to each other on the interpreter lock (round 6: the same
        # loop at 2.0-2.2 M or 3.1-4.0 M env-frames/s, run by run) cannot happen; the price is that the loss of a train is
        # noticed up to one message's staging time late.
        # Default (None): inline where the algorithm's model offers the hook AND calls it (``stage_inline_capable``: the
        # deferred in-graph tail), else the thread.  Same-box A/B, 4 x 2 runs (round 6): breakout_impala 3.55 / 3.65 / 4.01 /
        # 3.84 M inline against 3.06 / 2.79 / 3.20 / 2.35 M with the thread; pong_impala_speedup 13.8 / 14.2 against 13.0 / 11.5 M.
        if inline is None:
            inline = bool(getattr(alg, "stage_inline_capable", lambda: False)())
        self.inline = bool(inline)
        if self.inline:
            if not hasattr(alg, "stage_inline"):
                raise TypeError("Prefetcher(inline=True): {} has no stage_inline".format(type(alg).__name__))
            self._thread = None
            self._multi = hasattr(self.source, "poll_into")
            alg.stage_inline(self.pump_once)
            return
        self._thread = threading.Thread(target=self._run, name="xt-prefetch", daemon=True)
        self._thread.start()

    # ---- pump thread
    def _stage(self, data, ctr_info=None):
        ctr = dict(ctr_info or {})
        rows = self.alg.stage_message(data, ctr_info=ctr)
        ctr.pop("_slot_guard", None)
        ctr.pop("_pinned_views", None)
        if (self._staged + 1) % self.group == 0 and hasattr(self.alg, "stage_group_complete"):
            self.alg.stage_group_complete()         # the train's label block goes to HBM now, not inside train()
        with self._cv:
            self._tokens.append((ctr, rows))
            self._staged += 1
            self._cv.notify_all()

    def pump_once(self):
        """(inline mode, learner thread) stage ONE waiting message if the one-train-ahead rule allows it; -> did it?"""
        if self._staged and self._staged % self.group == 0 and self._ingest_gen() < self._staged // self.group:
            return False
        if self._multi:
            got = self.source.poll_into(self._stage, max_msgs=1)
        else:
            got = 1 if self.source.recv_into(self._stage, block=False) is not None else 0
        if not got and hasattr(self.source, "reap"):
            self.source.reap()
        return bool(got)

    def _run(self):
        try:
            if hasattr(self.alg, "stage_thread_init"):
                self.alg.stage_thread_init(wake=self.notify, idle=self.learner_idle if self.gate else None)
            multi = hasattr(self.source, "poll_into")
            while not self._stop:
                # one train ahead at most: the first message of the NEXT train goes into the buffer set the learner is about
                # to release -- wait until train (staged / group - 1) has been taken over by the learner (finish())
                if self._staged and self._staged % self.group == 0:
                    with self._cv:
                        while not self._stop and self._ingest_gen() < self._staged // self.group:
                            self._cv.wait(0.001)
                    if self._stop:
                        break
                announced = False
                if self.gate:
                    # (bounded: a learner that never blocks must not starve the ingest -- after 2 ms the message is staged anyway)
                    if self.learner_idle.wait(0.002):
                        announced = self.learner_idle.begin()
                        if not announced:
                            continue
                try:
                    if multi:
                        room = 1 if self.gate else self.group - self._staged % self.group     # (gated: one message per turn)
                        got = self.source.poll_into(self._stage, max_msgs=room)
                    else:
                        got = 1 if self.source.recv_into(self._stage, block=False) is not None else 0
                finally:
                    if announced:
                        self.learner_idle.end()
                if not got:
                    if hasattr(self.source, "reap"):
                        self.source.reap()
                    time.sleep(self._poll)
        except BaseException as exc:        # noqa: BLE001 -- surfaces in the learner thread's next recv_into
            with self._cv:
                # (without its traceback: the frames hold zero-copy views into the ring's shared memory)
                self._error = exc.with_traceback(None)
                self._cv.notify_all()

    # ---- learner thread: the ring's receive contract
    def recv_into(self, sink, block=True, timeout=None):
        """hand the oldest staged message's token to ``sink(data, ctr_info=...)`` (``Algorithm.prepare_data``); -> its control
        dict, or None (non-blocking / timed out)"""
        t0 = time.monotonic()
        if self.inline:
            while not self._tokens:
                if not self.pump_once():
                    if not block or (timeout is not None and time.monotonic() - t0 > timeout):
                        return None
                    time.sleep(self._poll)
            ctr, rows = self._tokens.pop(0)
            sink({"_prefetched": rows}, ctr_info=ctr)
            return ctr
        with self._cv:
            while not self._tokens:
                if self._error is not None:
                    raise RuntimeError("Prefetcher: the staging thread failed") from self._error
                if not block or (timeout is not None and time.monotonic() - t0 > timeout):
                    return None
                self.learner_idle.set()
                self._cv.wait(0.0005)
            ctr, rows = self._tokens.pop(0)
        self.learner_idle.clear()
        sink({"_prefetched": rows}, ctr_info=ctr)
        return ctr

    def recv_many_into(self, sink, count, timeout=None):
        got = 0
        t0 = time.monotonic()
        while got < count:
            left = None if timeout is None else max(0.0, timeout - (time.monotonic() - t0))
            if self.recv_into(sink, timeout=left) is None:
                break
            got += 1
        return got

    def notify(self):
        """the learner has taken over a buffer set (``train()`` called ``finish()``): wake the staging thread"""
        with self._cv:
            self._cv.notify_all()

    def close(self):
        self._stop = True
        if self.inline:
            try:
                self.alg.stage_inline(None)
            except Exception:       # noqa: BLE001
                pass
            return
        self.learner_idle.set()
        with self._cv:
            self._cv.notify_all()
        self._thread.join(timeout=5.0)
        if hasattr(self.alg, "stage_thread_init"):
            try:
                self.alg.stage_thread_init(wake=None, idle=None, bind_device=False)
            except Exception:       # noqa: BLE001
                pass
