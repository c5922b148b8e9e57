#!/usr/bin/env python
"""20 s of the ring-fed IMPALA loop (bench.impala_ring_loop, final form): throughput, host RSS and device memory before / after
-- a leak or a slow-down over ~100 k trains would show here.  GPU box."""
import os
import resource
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "breakout_impala"
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
w = bench.IMPALA[key]
mpt = w.get("msgs_per_train", 1 if key == "breakout_impala" else 4)
out = []
for _ in range(2):
    r = bench.impala_ring_loop(w, w["frames_per_train"] // mpt, mpt, w.get("train_per_checkpoint", 1), n_prod=2, seconds=secs / 2)
    torch.cuda.synchronize()
    out.append((round(r["value"]), r["trains"], resource.getrusage(resource.RUSAGE_SELF).ru_maxrss // 1024,
                torch.cuda.memory_allocated() >> 20))
print("value, trains, max RSS MiB, torch device MiB per half:", out)
