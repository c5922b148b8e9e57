"""Stand-alone hot-path ops on the GPU (thin host wrappers over the C ABI)."""
import numpy as np
import torch

from xingtian_amd import lib as L


def gae(value, reward, done, gamma=0.99, lam=0.95, device="cuda:0"):
    """GAE for n_traj trajectories at once, float64 on the GPU, bit-exact with the
    reference's numpy loop (xt/agent/ppo/ppo.py:87-104).

    value [n,T+1] f32, reward [n,T] f64, done [n,T] bool -> (adv [n,T] f64,
    old_value [n,T] f32, target_value [n,T] f64) as numpy arrays.
    """
    L.require_gpu()
    lib = L.load()
    value = np.ascontiguousarray(value, np.float32)
    reward = np.ascontiguousarray(reward, np.float64)
    done = np.ascontiguousarray(np.asarray(done, bool).astype(np.uint8))
    n, t = reward.shape
    if value.shape != (n, t + 1) or done.shape != (n, t):
        raise ValueError("gae: value must be [n,T+1], reward/done [n,T]")
    dev = torch.device(device)
    v = torch.from_numpy(value).to(dev)
    r = torch.from_numpy(reward).to(dev)
    d = torch.from_numpy(done).to(dev)
    adv = torch.empty((n, t), dtype=torch.float64, device=dev)
    tgt = torch.empty((n, t), dtype=torch.float64, device=dev)
    ov = torch.empty((n, t), dtype=torch.float32, device=dev)
    L.check(lib.xt_gae_f64(L.ptr(v), L.ptr(r), L.ptr(d), L.ptr(adv), L.ptr(tgt), L.ptr(ov), n, t,
                           float(gamma), float(lam), L.stream_ptr()), "xt_gae_f64")
    return adv.cpu().numpy(), ov.cpu().numpy(), tgt.cpu().numpy()
