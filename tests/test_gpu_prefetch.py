"""The asynchronous host path of the IMPALA learner: ``transport.Prefetcher`` (every rollout message is
staged to HBM when it ARRIVES, one train ahead at most) + ``WeightsRing.start_committer`` (the D2H the update enqueued into a
ring slot is committed by a helper thread) must not change WHAT is trained or published: against the reference-shaped
blocking loop (recv + prepare_data x k -> train() -> publish, xt/framework/learner.py:298-380) on the same message stream the
losses, the final parameters and the weights a reader fetches are bit for bit the same."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

T_LEN, ENVS, A_DIM, DIM, MSGS_PER_TRAIN, TRAINS = 10, 2, 6, 42, 2, 9


def _alg(tpc):
    from xingtian_amd.algorithm import alg_builder
    mi = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [DIM, DIM, 4], "input_dtype": "uint8", "state_mean": 128.0,
                    "state_std": 128.0, "action_dim": A_DIM, "type": "learner",
                    "model_config": {"LR": 1e-3, "sample_batch_step": T_LEN, "grad_norm_clip": 40.0, "SEED": 4}}}
    return alg_builder("IMPALAOpt", mi, {"instance_num": 4, "agent_num": 1, "prepare_times_per_train": MSGS_PER_TRAIN,
                                        "train_per_checkpoint": tpc, "BATCH_SIZE": ENVS * T_LEN * MSGS_PER_TRAIN})


def _msg(k):
    rng = np.random.default_rng(4000 + k)
    n = ENVS * T_LEN
    return {"cur_state": rng.integers(0, 256, (n, DIM, DIM, 4)).astype(np.uint8),
            "logit": rng.standard_normal((n, A_DIM)).astype(np.float32), "action": rng.integers(0, A_DIM, n).astype(np.int32),
            "done": list(rng.random(n) < 0.1), "reward": list(rng.choice([-1.0, 0.0, 1.0], n))}


def _run(prefetch, tpc, pinned):
    from xingtian_amd import transport
    alg = _alg(tpc)
    ring = transport.ShmRing(slots=4, slot_bytes=1 << 20)
    wring = transport.WeightsRing(slot_bytes=8 << 20, slots=4)
    reader = transport.WeightsRing(name=wring.name, create=False, slot_bytes=8 << 20, slots=4)
    if pinned:
        assert ring.pin()
    assert wring.pin()
    if prefetch:
        wring.start_committer()
        alg.actor.net.attach_weights_ring(wring)

    def produce():
        for k in range(TRAINS * MSGS_PER_TRAIN):
            assert ring.send({"cmd": "train", "k": k}, _msg(k), timeout=30)

    prod = threading.Thread(target=produce, daemon=True)
    prod.start()
    src = transport.Prefetcher(ring, alg) if prefetch else ring
    losses, seqs = [], []
    try:
        for t in range(TRAINS):
            for _ in range(MSGS_PER_TRAIN):
                assert src.recv_into(alg.prepare_data, timeout=30) is not None
            losses.append(float(alg.train(episode_num=t)))
            if alg.checkpoint_ready(t):
                seqs.append(int(alg.publish_weights(wring)))
        torch.cuda.synchronize()
        latest = wring.drain()
        got = reader.fetch()
        params = alg.actor.net.params.cpu().numpy().copy()
        weights = {k: v.copy() for k, v in alg.get_weights().items()}
    finally:
        if prefetch:
            src.close()
        prod.join(timeout=10)
        alg.actor.net.attach_weights_ring(None)
        reader.close()
        wring.close()
        ring.close()
    return losses, params, weights, seqs, latest, got


@pytest.mark.parametrize("tpc", [1, 3])
@pytest.mark.parametrize("pinned", [True, False])
def test_prefetch_and_asynchronous_commit_train_and_publish_exactly_what_the_blocking_loop_does(tpc, pinned):
    ref = _run(False, tpc, pinned)
    got = _run(True, tpc, pinned)
    assert ref[0] == got[0], (ref[0], got[0])                       # every reported loss, bit for bit
    assert np.array_equal(ref[1], got[1])                           # final parameters
    assert ref[3] == got[3] and got[4] == len(got[3]) == (TRAINS + tpc - 1) // tpc       # publish sequence numbers, all visible
    seq, ctr, w = got[5]
    if (TRAINS - 1) % tpc == 0:           # the last train published: the reader holds the learner's final weights
        assert seq == got[4]
        for name, v in got[2].items():
            assert np.array_equal(w[name], v), name
    seq_r, _ctr, w_r = ref[5]
    assert seq_r == seq
    for name in w:
        assert np.array_equal(w[name], w_r[name]), name
