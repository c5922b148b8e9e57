#!/usr/bin/env python
"""Where the host time of one IMPALAOpt train goes on the ring-fed, prefetched path (bench.impala_ring_loop): per-call wall
time of the pieces of train() on the learner thread, breakout_impala shape by default.  GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from xingtian_amd import ingest, transport  # noqa: E402
from xingtian_amd.model import hip_net  # noqa: E402

T = {}


def wrap(cls, name, key=None):
    fn = getattr(cls, name)
    key = key or "{}.{}".format(cls.__name__, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = T.setdefault(key, [0.0, 0])
            d[0] += time.perf_counter() - t0
            d[1] += 1
    setattr(cls, name, timed)


wrap(ingest.RolloutIngest, "finish")
wrap(ingest.RolloutIngest, "put")
wrap(ingest.RolloutIngest, "ship_labels")
wrap(hip_net.HipActorCritic, "impala_train_io")
NETS = []
_init = hip_net.HipActorCritic.__init__


def _init_keep(self, *a, **k):
    _init(self, *a, **k)
    self.gate_at_launch = "gatewait" not in sys.argv[2:]
    NETS.append(self)


hip_net.HipActorCritic.__init__ = _init_keep
wrap(hip_net.HipActorCritic, "snapshot_weights_async")
wrap(hip_net.HipActorCritic, "read_loss")
wrap(hip_net.HipActorCritic, "publish_weights")
wrap(transport.WeightsRing, "_reserve_flat")
wrap(transport.Prefetcher, "recv_into")
wrap(transport.WeightsRing, "publish_reserve")
wrap(transport.WeightsRing, "publish_enqueued")
wrap(ingest.RolloutIngest, "mapped_labels")
wrap(ingest.RolloutIngest, "consumed_event")
from xingtian_amd.model.impala import impala_cnn_opt  # noqa: E402
from xingtian_amd.algorithm.impala import impala_opt  # noqa: E402
wrap(impala_cnn_opt.ImpalaCnnOpt, "train_ingested")
wrap(impala_cnn_opt.ImpalaCnnOpt, "_lr_steps")
wrap(impala_opt.IMPALAOpt, "train")
wrap(impala_opt.IMPALAOpt, "stage_message")
wrap(ingest.RolloutIngest, "_ensure")
wrap(ingest.RolloutIngest, "seal")
wrap(ingest.RolloutIngest, "_join_copy_streams")
import numpy as _np  # noqa: E402
from xingtian_amd import lib as _L  # noqa: E402
for _mod, _name in ((_L, "memcpy_async"), (ingest.np, "copyto"), (ingest.np, "asarray")):
    _fn = getattr(_mod, _name)

    def _timed(*a, _fn=_fn, _key="fn." + _name, **k):
        t0 = time.perf_counter()
        try:
            return _fn(*a, **k)
        finally:
            d = T.setdefault(_key, [0.0, 0])
            d[0] += time.perf_counter() - t0
            d[1] += 1
    if _mod is _L:
        ingest.L.memcpy_async = _timed
    else:
        pass    # (numpy functions are shared module attributes: timed through their callers only)
wrap(torch.cuda.Event, "record", "Event.record")
wrap(torch.cuda.Event, "query", "Event.query")
wrap(torch.cuda.Stream, "wait_event", "Stream.wait_event")
wrap(impala_opt.IMPALAOpt, "stage_group_complete")
wrap(transport.Prefetcher, "_stage")
wrap(transport.RingSet, "poll_into")
wrap(transport.ShmRing, "recv_into", "ShmRing.recv_into")
wrap(transport.ShmRing, "recv_view")
wrap(transport.ShmRing, "_done_with_slot")
wrap(transport.ShmRing, "_reap")
wrap(transport.SlotGuard, "hold")
wrap(hip_net.HipActorCritic, "impala_wait_loss")
_dec = transport.decode


def _decode_timed(*a, **k):
    t0 = time.perf_counter()
    try:
        return _dec(*a, **k)
    finally:
        d = T.setdefault("transport.decode", [0.0, 0])
        d[0] += time.perf_counter() - t0
        d[1] += 1


transport.decode = _decode_timed
wrap(impala_opt.IMPALAOpt, "prepare_data")
wrap(impala_opt.IMPALAOpt, "publish_weights")
wrap(impala_opt.IMPALAOpt, "checkpoint_ready")
key = sys.argv[1] if len(sys.argv) > 1 else "breakout_impala"
w = bench.IMPALA[key]
mpt = w.get("msgs_per_train", 1 if key == "breakout_impala" else 4)
res = bench.impala_ring_loop(w, w["frames_per_train"] // mpt, mpt, w.get("train_per_checkpoint", 1), n_prod=2, seconds=1.0,
                             prefetch=(len(sys.argv) < 3 or sys.argv[2] != "blocking"),
                             async_commit=(len(sys.argv) < 3 or sys.argv[2] != "blocking"),
                             gate=(len(sys.argv) < 3 or sys.argv[2] != "nogate"), pack_lists="pylists" not in sys.argv[2:], strict="strict" in sys.argv[2:], inline=(False if "thread" in sys.argv[2:] else None),
                             model_config={"IO_TAIL_IN_GRAPH": 0 if "notail" in sys.argv[2:] else 1 if "ingraph" in sys.argv[2:] else 2, "USE_HIP_GRAPH": "nograph" not in sys.argv[2:],
                                           "INGEST_COPY_STREAMS": 2 if "cs2" in sys.argv[2:] else 1})
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()})
for k, (tot, n) in sorted(T.items(), key=lambda kv: -kv[1][0]):
    print("%-42s %7d calls  %8.1f us/call" % (k, n, 1e6 * tot / max(n, 1)))
for net in NETS:
    t = net.io_times()
    if t["calls"]:
        print("xt_net_impala_train_io phases (us/call):", {k: round(v, 1) if isinstance(v, float) else v for k, v in t.items()})
