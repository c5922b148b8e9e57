"""The direct 2-phase all-reduce (C ABI xt_allreduce_direct, csrc/xt_xgmi.hip; SURVEY.md 8(b) export list, 8(e)):
N logical ranks in one process and N PROCESSES on one GPU over hipIpc-mapped exchange blocks -- results bitwise equal to the
fixed-rank-order float32 host sum, hundreds of back-to-back iterations without a stale read; the exchange wired into
xt_net_ppo_train (captured into the update's hipGraph) against the step-wise data-parallel path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4, 8])
def test_in_process_group_is_bitwise_the_fixed_order_sum(world):
    import direct_worker as W
    from xingtian_amd.parallel import DirectComm
    ranks = DirectComm.local_group(world, max(W.COUNTS), timeout_ms=10000)
    streams = [torch.cuda.Stream() for _ in range(world)]
    try:
        n_it = 0
        for rep in range(3):
            for ci, count in enumerate(W.COUNTS):
                it = rep * len(W.COUNTS) + ci
                bufs = [torch.from_numpy(W.rank_input(it, r, count)).cuda() for r in range(world)]
                torch.cuda.synchronize()
                DirectComm.all_reduce_group_(ranks, bufs, streams)
                n_it += 1
                torch.cuda.synchronize()
                want = W.expected_sum(it, world, count)
                for r in range(world):
                    assert np.array_equal(bufs[r].cpu().numpy(), want), (world, count, r)
        for c in ranks:
            st = c.status()
            assert st["error_bits"] == 0 and st["seq"] == n_it, st
    finally:
        for c in ranks:
            c.destroy()


def test_single_rank_is_the_identity_and_bad_arguments_fail_loudly():
    from xingtian_amd.parallel import DirectComm
    c = DirectComm(0, 1, 1024)
    x = torch.arange(1000, dtype=torch.float32, device="cuda")
    y = x.clone()
    c.all_reduce_(y)
    torch.cuda.synchronize()
    assert torch.equal(x, y)
    with pytest.raises(RuntimeError, match="count"):
        c.all_reduce_(torch.zeros(2048, dtype=torch.float32, device="cuda"))
    c.destroy()
    with pytest.raises(RuntimeError, match="rank"):
        DirectComm(3, 2, 16)
    two = DirectComm(0, 2, 64)
    with pytest.raises(RuntimeError, match="connect"):
        two.all_reduce_(torch.zeros(64, dtype=torch.float32, device="cuda"))
    two.destroy()


def test_a_missing_peer_times_out_with_an_error_bit_instead_of_hanging():
    """rank 1 of an in-process pair never calls: rank 0's bounded waits run out (50 ms), the kernels finish, the error
    word says which wait failed"""
    from xingtian_amd.parallel import DirectComm
    ranks = DirectComm.local_group(2, 4096, timeout_ms=50)
    try:
        x = torch.ones(4096, dtype=torch.float32, device="cuda")
        import time
        t0 = time.perf_counter()
        ranks[0].all_reduce_(x)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        st = ranks[0].status()
        assert st["error_bits"] & 1, st
        # the error is sticky: the next all-reduces of this comm do not wait again
        t0 = time.perf_counter()
        for _ in range(10):
            ranks[0].all_reduce_(x)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < max(0.05, first), (first, time.perf_counter() - t0)
    finally:
        for c in ranks:
            c.destroy()


@pytest.mark.parametrize("world,mode", [(2, "fused"), (4, "fused"), (8, "fused"), (2, "chain"), (8, "chain")])
def test_n_processes_on_one_gpu_over_ipc_handles(tmp_path, world, mode):
    """200 all-reduces in batches of 50 back-to-back launches per rank, sizes cycling through the PpoCnn / ImpalaCnnOpt flat
    gradient sizes, odd tails and counts smaller than the number of ranks (empty slices): every rank's result bitwise = the
    fixed-order host sum, no timeout, sequence = 200 -- as ONE launch per all-reduce (fused, the default) and as the
    three-launch chain."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "direct_worker.py"), str(tmp_path),
           "200", "50", mode]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
    outs = []
    for r in range(world):
        p = os.path.join(str(tmp_path), "direct_r{}.txt".format(r))
        outs.append(open(p).read() if os.path.exists(p) else "missing")
    assert proc.returncode == 0 and all(o.startswith("OK") for o in outs), (outs, proc.stdout.decode()[-2000:])
    print("direct all-reduce,", world, "processes on one GPU:", outs[0].strip().split("mode=")[-1])
