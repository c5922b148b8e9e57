#!/usr/bin/env python
"""Per-block timelines of the layer kernels (diagnostic; needs `make -C xingtian_amd/csrc tl`).

Every block's thread 0 stores the 100 MHz wall clock at up to 6 marks (XT_TL in the kernels) plus a role word
and HW_ID/XCC_ID.  This script runs each layer kernel of the PpoCnn B=320 step once with the buffer armed,
dumps the raw marks to gpurun_out/timeline.npz and prints a summary: launch-to-launch period (HIP events,
50 back-to-back launches) against the span in which blocks were actually alive, per-phase durations, how
the blocks spread over XCDs/CUs.

Usage: python tools/timeline.py [B]        (on the GPU box)
       python tools/timeline.py --report gpurun_out/timeline.npz   (anywhere)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PHASES = {
    10: ("fwd", ["decode+fetch0/1 issued", "first stash+barrier", "k loop", "epilogue issue", "store drain"]),
    20: ("wgrad", ["rowtab+fetch issued", "first stash+barrier", "m loop", "epilogue issue", "store drain"]),
    21: ("wgrad (staged rows)", ["offset table + first fetch issued", "first stash+barrier", "sample loop (MFMA)", "slab store issue", "store drain"]),
    31: ("dgrad (all classes)", ["decode+fetch issued", "first stash+barrier", "k loop", "epilogue (x load + store issue)", "store drain"]),
    30: ("dgrad", ["decode+fetch issued", "first stash+barrier", "k loop", "epilogue (x load + store issue)", "store drain"]),
    40: ("conv1 fwd", ["loads issued", "split+LDS+barrier", "mfma loop", "transpose+store issue", "store drain"]),
    50: ("conv1 wgrad", ["loads issued+LDS", "barrier", "mfma loop", "combine+store issue", "store drain"]),
    70: ("direct fwd", ["decode+2 stages issued", "", "reduction loop", "combine+store issue", "store drain"]),
    80: ("direct dgrad", ["decode+x+2 stages issued", "", "reduction loop", "combine+store issue", "store drain"]),
    60: ("grads_finish", ["slab loads+sum", "block reduce", "ticket", "finalize (last block)", ""]),
}


def report(path):
    z = np.load(path, allow_pickle=True)
    for key in z.files:
        if key.endswith("_ms"):
            continue
        raw = z[key].astype(np.int64)
        ms = float(z[key + "_ms"]) if key + "_ms" in z.files else float("nan")
        used = raw[raw[:, 0] > 0]
        if not len(used):
            print("%s: no marks" % key)
            continue
        t0 = used[:, 0].min()
        print("==== %s: period %.2f us/launch (events, back-to-back); %d blocks with marks" % (key, ms * 1e3, len(used)))
        for role in np.unique(used[:, 6]):
            u = used[used[:, 6] == role]
            name, ph = PHASES.get(int(role), ("role%d" % role, [""] * 5))
            last = np.where(u[:, 1:6] > 0, u[:, 1:6], 0).max(axis=1)
            span = (last.max() - u[:, 0].min()) / 100.0
            st = (u[:, 0] - t0) / 100.0
            print("  -- %s: %d blocks, alive span %.2f us (first start +%.2f us, last start +%.2f us, last end +%.2f us)"
                  % (name, len(u), span, st.min(), st.max(), (last.max() - t0) / 100.0))
            life = (last - u[:, 0]) / 100.0
            print("     block lifetime: mean %.2f  p10 %.2f  p50 %.2f  p90 %.2f  max %.2f us"
                  % (life.mean(), np.percentile(life, 10), np.percentile(life, 50), np.percentile(life, 90), life.max()))
            prev = u[:, 0]
            for s in range(1, 6):
                cur = u[:, s]
                ok = (cur > 0) & (prev > 0)
                if ok.any():
                    d = (cur[ok] - prev[ok]) / 100.0
                    print("     phase %d %-34s mean %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f us  (n=%d)"
                          % (s, ph[s - 1], d.mean(), np.percentile(d, 50), np.percentile(d, 90), d.max(), ok.sum()))
                prev = np.where(cur > 0, cur, prev)
            hw = u[:, 7]
            xcc = (hw >> 32) & 0xF
            cu = (hw >> 8) & 0xF
            se = (hw >> 13) & 0x7
            sh = (hw >> 12) & 0x1
            cuid = xcc * 1000 + se * 100 + sh * 10 + cu
            uniq, cnt = np.unique(cuid, return_counts=True)
            print("     %d distinct CUs used; blocks per CU: min %d max %d; per XCD: %s"
                  % (len(uniq), cnt.min(), cnt.max(), np.bincount(xcc.astype(int), minlength=8).tolist()))
            # start-time histogram in 1 us buckets
            hist = np.bincount(np.minimum(st.astype(int), 39), minlength=1)
            print("     block starts per us: %s" % hist.tolist())


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        return report(sys.argv[2])
    import torch
    from xingtian_amd import lib as L
    L.LIB_PATH = os.environ.get("XT_TL_LIB") or os.path.join(ROOT, "xingtian_amd", "libxt_mi355x_tl.so")
    if os.environ.get("XT_TL_KNOBS"):          # e.g. XT_TL_KNOBS='{"wgrad_rows": 3}' (experiment builds)
        import json
        L.set_tuning(**json.loads(os.environ["XT_TL_KNOBS"]))
    from xingtian_amd.model import netspec
    from xingtian_amd.model.hip_net import HipActorCritic

    impala = len(sys.argv) > 1 and sys.argv[1] in ("impala42", "impala84")     # ImpalaCnnOpt layers instead of PpoCnn
    if impala:
        dim = 42 if sys.argv[1] == "impala42" else 84
        B = int(sys.argv[2]) if len(sys.argv) > 2 else (1000 if dim == 42 else 128)
        spec = netspec.impala_cnn_opt((dim, dim, 4), 6 if dim == 42 else 4, 128.0 if dim == 42 else 0.0,
                                      128.0 if dim == 42 else 255.0)
        N = B
    else:
        dim = 84
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
        spec = netspec.ppo_cnn((84, 84, 4), 4, (256,), "relu", True)
        N = 4096
    net = HipActorCritic(spec, max_batch=B, seed=0)
    lib = net.lib
    rng = np.random.default_rng(0)
    obs = torch.from_numpy(rng.integers(0, 256, (N, dim, dim, 4), dtype=np.uint8)).cuda()
    idx = None if impala else torch.from_numpy(rng.permutation(N)[:B].astype(np.int32)).cuda()
    net.forward(obs[:B])
    cap = 8192
    buf = torch.zeros((cap, 8), dtype=torch.int64, device="cuda")
    setters = [getattr(lib, "xt_tl_set_" + n) for n in ("igemm", "conv1", "optim", "direct")]
    for s in setters:
        s.restype = ctypes.c_int
        s.argtypes = [ctypes.c_void_p]

    def arm(on):
        torch.cuda.synchronize()
        for s in setters:
            rc = s(ctypes.c_void_p(buf.data_ptr() if on else 0))
            assert rc == 0, rc

    lib.xt_tl_null_period.restype = ctypes.c_int
    lib.xt_tl_null_period.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                      ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
    scratch = torch.zeros(8192 * 32, dtype=torch.float32, device="cuda")
    for nb in (1, 256, 1024, 4096):
        for wr in (0, 1):
            ms = ctypes.c_float()
            lib.xt_tl_null_period(200, nb, wr, ctypes.c_void_p(scratch.data_ptr()), ctypes.byref(ms), L.stream_ptr())
            print("null kernel %5d blocks wr=%d: %.2f us launch-to-launch" % (nb, wr, ms.value * 1e3))

    lib.xt_tl_clock_probe.restype = ctypes.c_int
    lib.xt_tl_clock_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    cbuf = torch.zeros(2, dtype=torch.int64, device="cuda")

    def clock_mhz(tag):
        lib.xt_tl_clock_probe(ctypes.c_void_p(cbuf.data_ptr()), L.stream_ptr())
        torch.cuda.synchronize()
        r, c = cbuf.cpu().tolist()
        print("shader clock %-28s %.0f MHz" % (tag, c / (r / 100.0)))

    clock_mhz("(idle)")
    for _ in range(3):
        net.time_layer(1, 0, obs, idx, B, reps=200)
        clock_mhz("(after 200 conv2 fwd)")
    if os.environ.get("XT_TL_TRAIN"):
        rngp = np.random.default_rng(1)
        perm = torch.from_numpy(np.stack([rngp.permutation(N) for _ in range(4)]).astype(np.int32)).cuda()
        act0 = torch.from_numpy(rngp.integers(0, 4, N).astype(np.int32)).cuda()
        g0 = lambda: torch.from_numpy(rngp.standard_normal(N).astype(np.float32)).cuda()
        lp0, ad0, ov0, tg0 = g0() - 1.5, g0(), g0(), g0()
        cfg0 = net.make_ppo_cfg(dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=0.5,
                                     MAX_GRAD_NORM=5.0, BATCH_SIZE=B, NUM_SGD_ITER=4))
        for rep in range(4):
            for _ in range(10):
                net.ppo_train(cfg0, obs, perm, act0, lp0, ad0, ov0, tg0, use_graph=True)
            clock_mhz("(after 10 ppo_train, rep %d)" % rep)

    out = {}
    jobs = [("L0_fwd", 0, 0), ("L0_wgrad", 0, 1)]
    for li in range(1, len(spec.layers)):
        jobs += [("L%d_fwd" % li, li, 0), ("L%d_bwd" % li, li, 3), ("L%d_dgrad" % li, li, 2)]
    for name, li, which in jobs:
        arm(False)
        ms = net.time_layer(li, which, obs, idx, B, reps=50)
        buf.zero_()
        arm(True)
        net.time_layer(li, which, obs, idx, B, reps=1)
        arm(False)
        out[name] = buf.cpu().numpy().copy()
        out[name + "_ms"] = np.float64(ms)
    if impala:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        path = os.path.join(ROOT, "gpurun_out", "timeline_%s.npz" % sys.argv[1])
        np.savez_compressed(path, **out)
        return report(path)
    # one whole SGD step: later kernels overwrite earlier ones' rows; the grads_finish rows (role 60) survive
    act = torch.from_numpy(rng.integers(0, 4, N).astype(np.int32)).cuda()
    f = lambda: torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
    logp, adv, oldv, tgt = f() - 1.5, f(), f(), f()
    cfg = net.make_ppo_cfg(dict(LR=2.5e-4, LOSS_CLIPPING=0.1, ENTROPY_LOSS=0.003, VF_CLIP=5.0, CRITIC_LOSS_COEF=0.5,
                                MAX_GRAD_NORM=5.0, BATCH_SIZE=B, NUM_SGD_ITER=4))
    try:
        step = lambda: net.ppo_step(cfg, obs, idx, act, logp, adv, oldv, tgt)
        step()
        buf.zero_()
        arm(True)
        step()
        arm(False)
        raw = buf.cpu().numpy().copy()
        raw[raw[:, 6] != 60] = 0
        out["grads_finish"] = raw
    except Exception as e:  # signature drift: the per-layer data above is what matters
        print("ppo_step timeline skipped:", repr(e))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "timeline.npz")
    np.savez_compressed(path, **out)
    report(path)


if __name__ == "__main__":
    main()
