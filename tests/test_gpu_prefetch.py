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


def _alg(tpc, tail=True, dim=DIM, t_len=T_LEN, envs=ENVS):
    from xingtian_amd.algorithm import alg_builder
    mi = {"actor": {"model_name": "ImpalaCnnOpt", "state_dim": [dim, dim, 4], "input_dtype": "uint8", "state_mean": 128.0,
                    "state_std": 128.0, "action_dim": A_DIM, "type": "learner",
                    "model_config": {"LR": 1e-3, "sample_batch_step": t_len, "grad_norm_clip": 40.0, "SEED": 4,
                                     "IO_TAIL_IN_GRAPH": tail}}}
    return alg_builder("IMPALAOpt", mi, {"instance_num": 4, "agent_num": 1, "prepare_times_per_train": MSGS_PER_TRAIN,
                                        "train_per_checkpoint": tpc, "BATCH_SIZE": envs * t_len * MSGS_PER_TRAIN})


def _msg(k, dim=DIM, t_len=T_LEN, envs=ENVS):
    rng = np.random.default_rng(4000 + k)
    n = envs * t_len
    return {"cur_state": rng.integers(0, 256, (n, dim, dim, 4)).astype(np.uint8),
            "logit": rng.standard_normal((n, A_DIM)).astype(np.float32), "action": rng.integers(0, A_DIM, n).astype(np.int32),
            "done": list(rng.random(n) < 0.1), "reward": list(rng.choice([-1.0, 0.0, 1.0], n))}


def _run(prefetch, tpc, pinned, tail=True, dim=DIM, t_len=T_LEN, envs=ENVS, trains=TRAINS):
    from xingtian_amd import transport
    alg = _alg(tpc, tail, dim, t_len, envs)
    ring = transport.ShmRing(slots=4, slot_bytes=max(1 << 20, envs * t_len * dim * dim * 4 + (1 << 16)))
    wring = transport.WeightsRing(slot_bytes=8 << 20, slots=4)
    reader = transport.WeightsRing(name=wring.name, create=False, slot_bytes=8 << 20, slots=4)
    if pinned:
        assert ring.pin()
    assert wring.pin()
    if prefetch:
        wring.start_committer()
        alg.actor.net.attach_weights_ring(wring)

    def produce():
        for k in range(trains * MSGS_PER_TRAIN):
            assert ring.send({"cmd": "train", "k": k}, _msg(k, dim, t_len, envs), timeout=30)

    prod = threading.Thread(target=produce, daemon=True)
    prod.start()
    # prefetch: True = the default form (inline: the learner thread stages while the device trains), "thread" = a staging thread
    src = transport.Prefetcher(ring, alg, inline=(False if prefetch == "thread" else None)) if prefetch else ring
    losses, seqs = [], []
    try:
        for t in range(trains):
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
    return losses, params, weights, seqs, latest, got, bool(getattr(src, "inline", False))


@pytest.mark.parametrize("tpc", [1, 3])
@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("how", [True, "thread"])
def test_prefetch_and_asynchronous_commit_train_and_publish_exactly_what_the_blocking_loop_does(tpc, pinned, how):
    # the blocking loop with the loss read-back and the weights copy as separate launches behind the train (events), the
    # asynchronous path with both as the train's own last kernels inside its replayed hipGraph (xt_train_io.tail_in_graph)
    ref = _run(False, tpc, pinned, tail=False)
    got = _run(how, tpc, pinned, tail=True)
    if how is True:
        assert got[6]                  # (the default picked the inline form for this algorithm)
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


@pytest.mark.parametrize("tail", [1, 2])
@pytest.mark.parametrize("prefetch", [True, "thread", False])
def test_the_in_graph_tail_reports_and_publishes_what_the_separate_launches_do(prefetch, tail):
    """xt_train_io.tail_in_graph on / off, everything else equal (train_per_checkpoint 3: trains WITH and WITHOUT a publish
    alternate through the same replayed graphs -- the destination travels through the mailbox)"""
    a = _run(prefetch, 3, True, tail=False)
    b = _run(prefetch, 3, True, tail=tail)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and a[3] == b[3] and a[4] == b[4]
    assert a[5][0] == b[5][0]
    for name in a[5][2]:
        assert np.array_equal(a[5][2][name], b[5][2][name]), name


@pytest.mark.parametrize("use_graph", [False, True])
@pytest.mark.parametrize("defer", [False, True])
@pytest.mark.parametrize("mode", [1, 2])
def test_train_io_tail_in_graph_through_the_c_abi(use_graph, defer, mode):
    """xt_net_impala_train_io with tail_in_graph against the same trains with the separate launches (loss copy + event,
    parameter copy + event): 6 trains on two alternating input sets, a publish on every second one (the destination travels
    through the mailbox: the replayed graphs are the same with and without it); every loss block, the device-side loss_acc,
    every published parameter block and the final parameters bit for bit; ``defer``: the split form (xt_net_io_wait)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    from xingtian_amd.model.hip_net import HipActorCritic
    spec, data, tlen, ntraj = dp_worker.impala_case()
    n = tlen * ntraj
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(9)
    sets = []
    for k in range(2):
        obs = data["obs"] if k == 0 else rng.integers(0, 256, data["obs"].shape).astype(data["obs"].dtype)
        sets.append([d(obs), d(data["bp"] + 0.1 * k), d(data["act"]), d(data["done"]), d(data["rew"])])

    def run(tail):
        net = HipActorCritic(spec, max_batch=n, seed=5)
        c = net.make_impala_cfg(1e-3, 40.0, tlen)
        slots = [torch.zeros(spec.n_flat, dtype=torch.float32, pin_memory=True) for _ in range(3)]
        evs = [torch.cuda.Event() for _ in range(3)]
        for ev in evs:
            ev.record()
        losses, accs, pubs = [], [], []
        for t in range(6):
            obs, bp, act, done, rew = sets[t % 2]
            # (tail: no event behind the graph -- the copy kernel reports its own completion through the mailbox)
            # (mode 2: a snapshot in the graph; the SDMA copy is made by handle.synchronize(), i.e. by whoever waits)
            pub = (slots[t % 3].data_ptr(), None if tail else evs[t % 3].cuda_event) if t % 2 == 0 else None
            a = net.impala_train_io(c, obs, n, bp, act, done, rew, use_graph=use_graph, publish=pub, wait_loss=True,
                                    tail_in_graph=tail, defer=defer and tail)
            landed = None
            if tail and pub is not None:
                handle = net.io_publish_done()
                handle.synchronize()
                assert handle.query()
                landed = slots[t % 3].numpy().copy()      # what the host sees the moment the sequence number has arrived
            if a is None:
                a = net.impala_wait_loss()
            losses.append(a.copy())
            torch.cuda.synchronize()
            accs.append(net.loss_acc.cpu().numpy()[:4].copy())
            if pub is not None:
                assert np.array_equal(slots[t % 3].numpy(), net.params.cpu().numpy())
                assert landed is None or np.array_equal(landed, slots[t % 3].numpy())
                pubs.append(slots[t % 3].numpy().copy())
        return losses, accs, pubs, net.params.cpu().numpy().copy()

    ref, got = run(0), run(mode)
    for a, b in zip(ref[0], got[0]):
        assert np.array_equal(a, b), (a, b)
    for a, b, c_ in zip(got[0], got[1], ref[1]):
        assert np.array_equal(a, b) and np.array_equal(b, c_)       # the device-side loss_acc holds the same four floats
    assert len(ref[2]) == len(got[2]) == 3
    for a, b in zip(ref[2], got[2]):
        assert np.array_equal(a, b)
    assert np.array_equal(ref[3], got[3])


@pytest.mark.parametrize("mode", [1, 2])
def test_the_copy_kernels_sequence_number_is_only_seen_behind_the_whole_parameter_block(mode):
    """The in-graph parameter copy reports its own completion from INSIDE the kernel (system-scope write-through stores, every
    workgroup's stores acknowledged, a ticket, the last workgroup writes the sequence number -- no fence, no event): 150
    trains of the breakout_impala shape (4.2 MB of parameters over the bus per train), the page-locked destination read by
    the host the moment the number has arrived must already be the train's parameters, bit for bit."""
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    tlen = 64
    spec = netspec.impala_cnn_opt((84, 84, 4), 4, 0.0, 255.0, "uint8")
    net = HipActorCritic(spec, max_batch=tlen, seed=3)
    c = net.make_impala_cfg(1e-3, 40.0, tlen)
    rng = np.random.default_rng(1)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    obs = d(rng.integers(0, 256, (tlen, 84, 84, 4)).astype(np.uint8))
    bp, act = d(rng.standard_normal((tlen, 4)).astype(np.float32)), d(rng.integers(0, 4, tlen).astype(np.int32))
    done, rew = d((rng.random(tlen) < 0.05).astype(np.uint8)), d(rng.choice([-1.0, 0.0, 1.0], tlen).astype(np.float32))
    slots = [torch.zeros(spec.n_flat, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    landed = np.empty(spec.n_flat, np.float32)
    for t in range(150):
        a = net.impala_train_io(c, obs, tlen, bp, act, done, rew, use_graph=True, publish=(slots[t % 2].data_ptr(), None),
                                wait_loss=True, tail_in_graph=mode, defer=True)
        assert a is None
        net.io_publish_done().synchronize()       # (mode 2: waits for the snapshot's report, then the SDMA copy, synchronously)
        np.copyto(landed, slots[t % 2].numpy())
        a = net.impala_wait_loss()
        torch.cuda.synchronize()
        assert np.isfinite(a[0]) and a[1] == 1.0
        assert np.array_equal(landed, net.params.cpu().numpy()), t


def test_sdma_copies_through_the_hsa_runtime_of_the_process():
    """csrc/xt_sdma.hip through the C ABI: ticketed page-locked-host -> device copies (xt_dma_h2d_async / xt_dma_wait_upto:
    tickets count up, ``upto`` covers every earlier copy, the query form) out of a torch pinned tensor and out of a
    hipHostRegister'ed /dev/shm ring, and the synchronous device -> host copy (xt_sdma_copy_d2h) back into both; bytes exact."""
    import ctypes
    from xingtian_amd import lib as L, transport
    lib = L.load()
    n = 1_051_003
    rng = np.random.default_rng(2)
    ring = transport.WeightsRing(slot_bytes=8 << 20, slots=2)
    assert ring.pin()
    shm = np.frombuffer(ring.shm.buf, dtype=np.float32, count=n, offset=8192 + 328)
    shm_addr = ring._pin_addr + 8192 + 328
    pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
    try:
        tickets = []
        for k, (host_np, addr) in enumerate(((pinned.numpy(), pinned.data_ptr()), (shm, shm_addr)) * 2):
            host_np[:] = rng.standard_normal(n).astype(np.float32)
            dev = torch.zeros(n, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            tk = ctypes.c_uint64()
            L.check(lib.xt_dma_h2d_async(dev.data_ptr(), addr, n * 4, ctypes.byref(tk)), "xt_dma_h2d_async")
            tickets.append(int(tk.value))
            assert lib.xt_dma_wait_upto(tk.value, 5000) == 0 and lib.xt_dma_wait_upto(tk.value, 0) == 0
            assert np.array_equal(dev.cpu().numpy(), host_np)
            back = pinned if k % 2 else None
            dst_np, dst_addr = (pinned.numpy(), pinned.data_ptr()) if back is not None else (shm, shm_addr)
            dev.mul_(2.0)
            torch.cuda.synchronize()
            L.check(lib.xt_sdma_copy_d2h(dst_addr, dev.data_ptr(), n * 4), "xt_sdma_copy_d2h")
            assert np.array_equal(dst_np, dev.cpu().numpy())
        assert tickets == list(range(tickets[0], tickets[0] + 4))
        # several copies in flight, one wait for all of them
        devs = [torch.zeros(n, dtype=torch.float32, device="cuda") for _ in range(6)]
        torch.cuda.synchronize()
        last = ctypes.c_uint64()
        for d_ in devs:
            L.check(lib.xt_dma_h2d_async(d_.data_ptr(), pinned.data_ptr(), n * 4, ctypes.byref(last)), "xt_dma_h2d_async")
        assert lib.xt_dma_wait_upto(last.value, 5000) == 0
        for d_ in devs:
            assert np.array_equal(d_.cpu().numpy(), pinned.numpy())
    finally:
        host_np = dst_np = shm = None        # (no view into the ring's shared memory may outlive it)
        ring.close()


def test_ticketed_sdma_ingest_trains_on_the_same_bytes_as_the_stream_copies():
    """The frames of a message in a page-locked ring slot reach HBM through xt_dma_h2d_async (SDMA engine via HSA: ordered
    against no stream, invisible to the device's caches), the frames of a pageable ring through pinned staging + hipMemcpyAsync
    on a copy stream.  Same messages -> every loss and the final parameters bit for bit (a kernel that read a stale cached
    line of a reused buffer set would show here: 9 trains over 2 buffer sets, every message different)."""
    a = _run(True, 1, False, tail=2)          # pageable ring: staging copy + stream H2D
    b = _run(True, 1, True, tail=2)           # pinned ring: ticketed SDMA copies
    c = _run(False, 1, True, tail=0)          # pinned ring, blocking loop, separate launches (events, no mailbox)
    assert a[0] == b[0] == c[0], (a[0], b[0], c[0])
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[1], c[1])
    # the breakout_impala shape: 2 x 1.8 MB of frames per train through 2 buffer sets, 24 trains
    big = dict(dim=84, t_len=32, envs=2, trains=24)
    a = _run(True, 1, False, tail=2, **big)
    b = _run(True, 1, True, tail=2, **big)
    assert a[0] == b[0], (a[0], b[0])
    assert np.array_equal(a[1], b[1])
