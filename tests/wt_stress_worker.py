"""Worker of tests/test_gpu_wt_stress.py: many consecutive learner updates with the library named on the command line
(the product library, or its `nowt` twin whose write-through stores are compiled to plain stores); writes the final
parameters + optimiser slots of the PPO and the IMPALA network to <out>.npz."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    libname, out, updates = sys.argv[1], sys.argv[2], int(sys.argv[3])
    from xingtian_amd import lib
    lib.LIB_PATH = os.path.join(ROOT, "xingtian_amd", libname)
    assert os.path.exists(lib.LIB_PATH), lib.LIB_PATH
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    rng = np.random.default_rng(123)
    # ---- BASELINE configs[1]: PpoCnn 84x84x4, B = 320, 4096 samples, 4 epochs (52 SGD steps per update), hipGraph replay
    cfg = dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=1.0, MAX_GRAD_NORM=5.0,
               BATCH_SIZE=320, NUM_SGD_ITER=4)
    n = 4096
    net = HipActorCritic(netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True), max_batch=320, seed=0)
    obs = d(rng.integers(0, 256, (n, 84, 84, 4), dtype=np.uint8))
    action = d(rng.integers(0, 4, n).astype(np.int32))
    logp = d((-np.abs(rng.standard_normal(n)) - 0.5).astype(np.float32))
    adv = d(rng.standard_normal(n))
    old_v = d(rng.standard_normal(n).astype(np.float32))
    tgt = d(rng.standard_normal(n))
    perm = torch.empty((4, n), dtype=torch.int32, device="cuda")
    c = net.make_ppo_cfg(cfg)
    for u in range(updates):
        perm.copy_(torch.from_numpy(np.stack([rng.permutation(n) for _ in range(4)]).astype(np.int32)))
        net.ppo_train(c, obs, perm, action, logp, adv, old_v, tgt, use_graph=True)
    torch.cuda.synchronize()
    res = {"ppo_params": net.params.cpu().numpy(), "ppo_m": net.adam_m.cpu().numpy(), "ppo_v": net.adam_v.cpu().numpy()}
    del net
    # ---- BASELINE configs[2] / [4] shapes: ImpalaCnnOpt 84x84 (128-frame trains) and 42x42 (1000-frame trains)
    for dim, a_dim, tlen, frames, tag in ((84, 4, 128, 128, "imp84"), (42, 6, 50, 1000, "imp42")):
        inet = HipActorCritic(netspec.impala_cnn_opt((dim, dim, 4), a_dim, 128.0, 128.0), max_batch=frames, seed=0)
        nn = frames * 8
        iobs = d(rng.integers(0, 256, (nn, dim, dim, 4), dtype=np.uint8))
        bp = d(rng.standard_normal((nn, a_dim)).astype(np.float32))
        act = d(rng.integers(0, a_dim, nn).astype(np.int32))
        done = d((rng.random(nn) < 0.01).astype(np.uint8))
        rew = d(rng.choice([-1.0, 0.0, 1.0], nn).astype(np.float32))
        ic = inet.make_impala_cfg(5e-4, 40.0, tlen)
        for u in range(max(updates // 4, 5)):
            inet.impala_train(ic, iobs, frames, bp, act, done, rew, use_graph=True)
        torch.cuda.synchronize()
        res[tag + "_params"] = inet.params.cpu().numpy()
        del inet
    assert all(np.isfinite(v).all() for v in res.values())
    np.savez(out, **res)


if __name__ == "__main__":
    main()
