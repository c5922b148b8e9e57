#!/usr/bin/env python
"""Kernel-side cost of the direct 2-phase all-reduce (xt_allreduce_direct, csrc/xt_xgmi.hip) on ONE device: N logical
ranks in this process (xt_direct_connect_local), one stream each, phase-ordered launches.  All ranks share the GPU, so this
is NOT an xGMI measurement: it prices the three-launch chain (scatter -> reduce -> gather: two flag hand-offs, three kernel
boundaries per rank) and the local memory traffic -- the part of the all-reduce that does not depend on the links.
Prints one JSON line: {world: {us_per_allreduce, ...}} for the PpoCnn (847 496) and ImpalaCnnOpt (1 005 109) gradient sizes."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from xingtian_amd.parallel import DirectComm
    out = {}
    for count in (847496, 1005109):
        res = {}
        for world in (2, 4, 8):
            ranks = DirectComm.local_group(world, count, timeout_ms=5000)
            streams = [torch.cuda.Stream() for _ in range(world)]
            bufs = [torch.full((count,), float(r + 1), dtype=torch.float32, device="cuda") for r in range(world)]
            torch.cuda.synchronize()
            for _ in range(5):
                DirectComm.all_reduce_group_(ranks, bufs, streams)
            torch.cuda.synchronize()
            reps = 100
            t0 = time.perf_counter()
            for _ in range(reps):
                DirectComm.all_reduce_group_(ranks, bufs, streams)
            torch.cuda.synchronize()
            us = 1e6 * (time.perf_counter() - t0) / reps
            st = ranks[0].status()
            errs = [c.status()["error_bits"] for c in ranks]
            # one rank alone on its stream inside a hipGraph-like back-to-back chain: the per-rank launch chain
            res[str(world)] = {"us_per_allreduce_all_ranks_on_one_gpu": round(us, 2),
                               "bytes_moved_on_this_gpu": int(world * (2 * count * 4 + 2 * count * 4)), "error_bits": errs}
            for c in ranks:
                c.destroy()
        out[str(count)] = res
    print(json.dumps({"direct_probe": out, "note": "N in-process ranks share ONE GPU: chain + local traffic, not xGMI"}))


if __name__ == "__main__":
    main()
