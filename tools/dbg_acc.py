import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from xingtian_amd import lib as L
from xingtian_amd.model import netspec
from xingtian_amd.model.hip_net import HipActorCritic
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
rng = np.random.default_rng(6)
tlen, ntraj, a_dim, bs = 10, 5, 6, 20
m = tlen * ntraj
ispec = netspec.impala_cnn_opt((42, 42, 4), a_dim, 128.0, 128.0)
bufs = [d(rng.integers(0, 256, (m, 42, 42, 4)).astype(np.uint8)), d(rng.standard_normal((m, a_dim)).astype(np.float32)),
        d(rng.integers(0, a_dim, m).astype(np.int32)), d((rng.random(m) < 0.1).astype(np.uint8)),
        d(rng.choice([-2.0, 0.0, 1.0], m).astype(np.float32))]
for knob, g in ((0, True), (0, False), (1, False), (1, True)):
    old = L.set_tuning(tail_overlap=knob)
    net = HipActorCritic(ispec, max_batch=bs, seed=0)
    c = net.make_impala_cfg(7e-4, 40.0, tlen)
    for _ in range(3):
        acc = net.impala_train(c, bufs[0], bs, *bufs[1:], use_graph=g)
        torch.cuda.synchronize()
        print(knob, g, acc.cpu().numpy(), acc.cpu().numpy().view(np.uint32)[:4])
    L.set_tuning(**old)
