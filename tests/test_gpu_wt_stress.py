"""Pins the write-through (sc1) store path.  Round 3 issued the 16-byte sc1 stores from inline asm
with a hand-placed `s_nop 1` behind them; without it the two-rank bitwise test failed 4 runs of 5 (a VMEM store of more
than 64 bits followed by a VALU write of its data registers needs two wait states on gfx940+, and the hazard recogniser
does not look inside asm).  Round 4 emits them through compiler-visible builtins (xt_common.h: store4_wt / store1_wt), and
this test holds the path against a twin library in which every such store is a PLAIN store (-DXT_NO_WT,
libxt_mi355x_nowt.so): hundreds of consecutive updates must end in bit-identical parameters and optimiser slots --
single process (PPO: 500 updates = 26 000 SGD steps through the replayed hipGraph; both IMPALA shapes) and two ranks."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOWT = os.path.join(ROOT, "xingtian_amd", "libxt_mi355x_nowt.so")


def _need_twin():
    if not os.path.exists(NOWT):
        pytest.fail("libxt_mi355x_nowt.so is missing: run `make -C xingtian_amd/csrc` (target `all` builds it)")


def test_write_through_stores_equal_plain_stores_over_500_updates(tmp_path):
    _need_twin()
    outs = []
    for libname in ("libxt_mi355x.so", "libxt_mi355x_nowt.so"):
        out = os.path.join(str(tmp_path), libname + ".npz")
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "wt_stress_worker.py"), libname, out, "500"],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert proc.returncode == 0, proc.stdout.decode()[-3000:]
        outs.append(dict(np.load(out).items()))
    assert sorted(outs[0]) == sorted(outs[1]) and len(outs[0]) == 5
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), "write-through build differs from the plain-store build: " + k
    assert np.abs(outs[0]["ppo_params"]).max() > 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_write_through_stores_equal_plain_stores_with_two_ranks(tmp_path):
    """the configuration in which round 3's missing wait states showed: two processes on the GPU at once"""
    _need_twin()
    res = {}
    for libname in ("libxt_mi355x.so", "libxt_mi355x_nowt.so"):
        sub = os.path.join(str(tmp_path), libname)
        os.makedirs(sub)
        for rep in range(3):
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                   "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py"), sub, "strict_hook"]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2", XT_TEST_LIB=libname)
            proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=240)
            assert proc.returncode == 0, proc.stdout.decode()[-3000:]
            p0 = np.load(os.path.join(sub, "params_strict_hook_r0.npy"))
            p1 = np.load(os.path.join(sub, "params_strict_hook_r1.npy"))
            assert np.array_equal(p0, p1), "replicas diverged (%s, run %d)" % (libname, rep)
            res.setdefault(libname, []).append(p0)
    ref = res["libxt_mi355x_nowt.so"][0]
    for libname, runs in res.items():
        for rep, p in enumerate(runs):
            assert np.array_equal(p, ref), "%s run %d differs from the plain-store build" % (libname, rep)
