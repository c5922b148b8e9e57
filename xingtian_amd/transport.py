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

* ``FrameSocket`` -- the inter-node hop: the two-frame multipart message of ``CommByZmq`` (control dict | payload,
  zeus/common/ipc/comm_by_zmq.py:69-97) over a plain stream socket with length-prefixed frames (zmq is not installed
  here; a maintainer who keeps zmq sends the same two buffers with ``send_multipart``).  ``recv_bytes`` / ``send_bytes``
  forward a message without decoding its payload (what the broker does, broker.py:97-119); ``recv_into(sink)`` receives
  into a reusable buffer and hands zero-copy views to ``Algorithm.prepare_data``.

Plumbing only: no arithmetic; the only GPU-runtime call is the optional ``hipHostRegister`` of ``pin()``.
"""
import socket as _socket
import struct
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


def _layout(ctr_info, data):
    """-> (msgpack header bytes, byte offset of the array section, [(offset, nbytes, contiguous array)], total bytes)"""
    arrays, objects, metas = [], {}, []
    off = 0
    for key, val in data.items():
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


def encode(ctr_info, data):
    """``data``: dict field -> ndarray | python object (the reference's train_data / weights dict).  Returns a
    ``bytearray``: MAGIC | u32 header length | msgpack header | padding | array bytes (64-byte aligned each)."""
    header, base, arrays, total = _layout(ctr_info, data)
    buf = bytearray(total)
    _write(memoryview(buf), header, base, arrays)
    return buf


def encode_into(view, ctr_info, data):
    """Encode straight into a writable buffer (a shared-memory slot): one copy per array, no intermediate message.
    Returns the encoded length; ValueError if the buffer is too small."""
    header, base, arrays, total = _layout(ctr_info, data)
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
    def send(self, ctr_info, data, block=True, timeout=None):
        """Encode straight into the next free slot (ShareByPlasma.send / CommByZmq.send contract)."""
        msg = encode(ctr_info, data)
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
            raise RuntimeError("WeightsRing: {} publishes begun and not committed (slots = {})".format(len(self._pending), self.slots))
        k = int(self._latest[0]) + 1 + len(self._pending)
        i = k % self.slots
        self._hdr[i][0] = 0
        header = b"\x84" + msgpack.packb("ctr") + msgpack.packb(_plain(dict(ctr_info or {}, cmd="weights", seq=k)),
                                                                use_bin_type=True) + lay[1]
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
        k, i, base = self._reserve_flat(net.spec, nbytes, ctr_info)
        st = getattr(self, "_d2h", None)
        if st is None:
            st = self._d2h = (torch.cuda.Stream(device=net.device), torch.cuda.Event(),
                              [torch.cuda.Event() for _ in range(self.slots)])
        side, ready, dones = st
        done = dones[i]
        ready.record(torch.cuda.current_stream(net.device))
        side.wait_event(ready)
        from xingtian_amd import lib as L
        L.memcpy_async(self._pin_addr + self._payload_offset(i) + base, net.params.data_ptr(), nbytes, L.D2H, side)
        done.record(side)
        self._pending.append((k, i, base + nbytes, done))
        return k

    def commit_flat_publish(self):
        """Second half: wait for the OLDEST begun copy and make that publish visible to the readers.  Returns its
        sequence number."""
        k, i, total, done = self._pending.pop(0)
        done.synchronize()
        self._hdr[i][1] = total
        self._hdr[i][0] = k
        self._latest[0] = k
        return k

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


class FrameSocket(object):
    """Two-frame messages (control dict | payload) over a stream socket: the inter-node counterpart of ``ShmRing`` with
    the contract of ``CommByZmq`` (zeus/common/ipc/comm_by_zmq.py:69-97: ``send`` / ``recv`` of ``(ctr_info, data)``,
    ``send_bytes`` / ``recv_bytes`` of the two raw frames for a forwarding broker).  Wire format per message:
    ``b"XTF1" | u32 n_frames | n_frames x u64 length | the frames``; frame 0 = msgpack of the control dict, frame 1 =
    ``encode({}, data)`` (msgpack header + raw 64-byte-aligned arrays).  One connection = one producer and one consumer."""

    MAGIC = b"XTF1"

    def __init__(self, sock):
        self.sock = sock
        self.sock.setsockopt(_socket.IPPROTO_TCP, _socket.TCP_NODELAY, 1) if sock.family in (_socket.AF_INET, _socket.AF_INET6) else None
        self._buf = self._aligned(1 << 20)    # reusable receive buffer of recv_into (grows to the largest message)

    @staticmethod
    def _aligned(nbytes):
        """writable memoryview of ``nbytes`` bytes whose first byte sits on a 64-byte boundary (the payload's arrays are
        64-byte aligned relative to the frame start: aligned views for the staging copy / a later hipHostRegister)"""
        raw = np.empty(nbytes + _ALIGN, np.uint8)
        off = (-raw.ctypes.data) % _ALIGN
        return memoryview(raw[off:off + nbytes])

    # ---- connection set-up (the reference binds the learner side and connects the explorers, comm_by_zmq.py:45-66)
    @staticmethod
    def listen(addr="127.0.0.1", port=0, backlog=8):
        """-> (listening socket, bound port); accept connections with ``FrameSocket.accept``."""
        srv = _socket.socket(_socket.AF_INET, _socket.SOCK_STREAM)
        srv.setsockopt(_socket.SOL_SOCKET, _socket.SO_REUSEADDR, 1)
        srv.bind((addr, port))
        srv.listen(backlog)
        return srv, srv.getsockname()[1]

    @staticmethod
    def accept(srv, timeout=None):
        srv.settimeout(timeout)
        conn, _ = srv.accept()
        conn.settimeout(None)
        return FrameSocket(conn)

    @staticmethod
    def connect(addr, port, timeout=10.0):
        t0 = time.monotonic()
        while True:
            try:
                return FrameSocket(_socket.create_connection((addr, port), timeout=timeout))
            except (ConnectionRefusedError, OSError):
                if time.monotonic() - t0 > timeout:
                    raise
                time.sleep(0.02)

    # ---- frames
    def send_bytes(self, ctr_frame, data_frame):
        """Send the two frames as they are (a broker forwards what ``recv_bytes`` gave it without decoding the payload)."""
        head = self.MAGIC + struct.pack("<IQQ", 2, len(ctr_frame), len(data_frame))
        self.sock.sendall(head)
        self.sock.sendall(ctr_frame)
        self.sock.sendall(data_frame)

    def _recv_exact(self, view):
        got = 0
        while got < len(view):
            k = self.sock.recv_into(view[got:])
            if k == 0:
                raise ConnectionError("FrameSocket: peer closed the connection mid-message" if got else "FrameSocket: connection closed")
            got += k

    def _recv_header(self):
        head = bytearray(4 + 4 + 16)
        self._recv_exact(memoryview(head))
        if bytes(head[:4]) != self.MAGIC:
            raise ValueError("FrameSocket: not an XTF1 message")
        n, l0, l1 = struct.unpack_from("<IQQ", head, 4)
        if n != 2:
            raise ValueError("FrameSocket: {} frames (expected 2)".format(n))
        return l0, l1

    def recv_bytes(self):
        """-> (control frame, payload frame) as private ``bytes`` / ``bytearray`` objects."""
        l0, l1 = self._recv_header()
        f0, f1 = bytearray(l0), bytearray(l1)
        self._recv_exact(memoryview(f0))
        self._recv_exact(memoryview(f1))
        return bytes(f0), f1

    # ---- (ctr_info, data) contract
    def send(self, ctr_info, data):
        self.send_bytes(msgpack.packb(_plain(ctr_info), use_bin_type=True), encode({}, data))

    def recv(self):
        """-> (ctr_info, data) with private arrays (views into a buffer this call allocated)."""
        f0, f1 = self.recv_bytes()
        return msgpack.unpackb(f0, raw=False, strict_map_key=False), decode(f1)[1]

    def recv_into(self, sink):
        """Receive the next message into the reusable buffer and hand it to ``sink(data, ctr_info=...)`` as zero-copy
        views (valid until the next ``recv_into``): the streaming ingest copies what it keeps (one host copy: socket
        buffer -> pinned staging).  Returns the control dict."""
        l0, l1 = self._recv_header()
        base = _pad(l0)                        # payload frame 64-byte aligned inside the buffer: aligned array views
        if base + l1 > len(self._buf):
            self._buf = self._aligned(_pad(base + l1))
        view = self._buf
        self._recv_exact(view[:l0])
        self._recv_exact(view[base:base + l1])
        ctr = msgpack.unpackb(bytes(view[:l0]), raw=False, strict_map_key=False)
        data = decode(view[base:base + l1])[1]
        sink(data, ctr_info=ctr)
        del data, view
        return ctr

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass
