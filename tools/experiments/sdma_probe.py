#!/usr/bin/env python
"""xt_sdma_copy_d2h inside the torch process: correctness into a registered /dev/shm ring slot and a torch pinned tensor, and
(under rocprofv3 --memory-copy-trace --stats) which engine ran it."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xingtian_amd import lib as L, transport  # noqa: E402

lib = L.load()
n = 1_051_003
src = torch.arange(n, dtype=torch.float32, device="cuda") * 0.5
torch.cuda.synchronize()
ring = transport.WeightsRing(slot_bytes=8 << 20, slots=4)
assert ring.pin()
pinned = torch.zeros(n, dtype=torch.float32, pin_memory=True)
dst_shm = ring._pin_addr + 8192 + 328
view = np.frombuffer(ring.shm.buf, dtype=np.float32, count=n, offset=8192 + 328)
for name, dst, chk in (("pinned", pinned.data_ptr(), pinned.numpy()), ("shm", dst_shm, view)):
    t0 = time.perf_counter()
    for _ in range(5):
        L.check(lib.xt_sdma_copy_d2h(dst, src.data_ptr(), n * 4), "xt_sdma_copy_d2h")
    dt = (time.perf_counter() - t0) / 5
    ok = np.array_equal(chk, src.cpu().numpy())
    print(name, "ok" if ok else "MISMATCH", "%.1f us per copy" % (dt * 1e6))
del view
ring.close()
