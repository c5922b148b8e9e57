#!/usr/bin/env python
"""Which engine does a 4 MB hipMemcpyAsync D2H take in THIS process (torch loaded)?  Run under
rocprofv3 --kernel-trace --memory-copy-trace --stats: MEMORY_COPY_DEVICE_TO_HOST = SDMA, __amd_rocclr_copyBuffer = blit kernel."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from xingtian_amd import lib as L, transport  # noqa: E402

n = 1_051_000 * 4
src = torch.ones(n // 4, dtype=torch.float32, device="cuda")
ring = transport.WeightsRing(slot_bytes=8 << 20, slots=4)
assert ring.pin()
pinned = torch.empty(n // 4, dtype=torch.float32, pin_memory=True)
side = torch.cuda.Stream()
hip = ctypes.CDLL("libamdhip64.so")
raw = ctypes.c_void_p()
hip.hipStreamCreateWithFlags(ctypes.byref(raw), 1)
hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
own = ctypes.c_void_p()
hip.hipMalloc(ctypes.byref(own), n)
which = sys.argv[1]
for rep in range(5):
    if which == "shm_torchstream":
        L.memcpy_async(ring._pin_addr + 8192 + 328, src.data_ptr(), n, L.D2H, side)
    elif which == "pinned_torchstream":
        L.memcpy_async(pinned.data_ptr(), src.data_ptr(), n, L.D2H, side)
    elif which == "shm_rawstream":
        hip.hipMemcpyAsync(ctypes.c_void_p(ring._pin_addr + 8192 + 328), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(n), 2, raw)
    elif which == "shm_rawstream_ownsrc":
        hip.hipMemcpyAsync(ctypes.c_void_p(ring._pin_addr + 8192 + 328), own, ctypes.c_size_t(n), 2, raw)
    elif which == "shm_nullstream":
        hip.hipMemcpyAsync(ctypes.c_void_p(ring._pin_addr + 8192 + 328), ctypes.c_void_p(src.data_ptr()), ctypes.c_size_t(n), 2, None)
    torch.cuda.synchronize()
ring.close()
print("done", which)
