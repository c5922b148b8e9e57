#!/usr/bin/env python
"""Per-kernel summary of the separate rocprofv3 --pmc passes of tools/profile_round.sh.

Usage: python tools/pmc_summary.py <FETCH_SIZE pass dir> <WRITE_SIZE pass dir> <SQ pass dir> <out.json> [kernel_stats.csv ...]

Per kernel (mean per launch): HBM-side bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests
at 64 B for 16-byte-coalesced reads -- MI355X_MICROARCH.md, HBM section; counts L2 memory-side requests, Infinity-Cache
hits included), MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs ... reported raw and as
busy/(wave-resident time)), wait fractions of the wave cycles."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def collect(d):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                a = acc[row["Kernel_Name"]][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return {k: {c: v[0] / v[1] for c, v in cs.items() if v[1]} for k, cs in acc.items()}


def durations(paths):
    """kernel name -> in-graph average duration (ns) from rocprofv3 --kernel-trace --stats summaries"""
    d = {}
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if int(row["Calls"]) >= 20:
                    d.setdefault(row["Name"], float(row["AverageNs"]))
    return d


SIMDS, GHZ = 1024, 2.4      # 256 CUs x 4 SIMDs; shader clock of the measured runs (tools/timeline.py: 2.40-2.43 GHz)


def main():
    fdir, wdir, sdir, out = sys.argv[1:5]
    dur = durations(sys.argv[5:])
    fetch, write, sq = collect(fdir), collect(wdir), collect(sdir)
    rows = []
    for k in sorted(set(fetch) | set(write) | set(sq)):
        if "xt::" not in k:
            continue
        r = {"kernel": k, "FETCH_SIZE_KB": fetch.get(k, {}).get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get(k, {}).get("WRITE_SIZE")}
        if r["FETCH_SIZE_KB"] is not None and r["WRITE_SIZE_KB"] is not None:
            r["hbm_side_MB"] = (2.0 * r["FETCH_SIZE_KB"] + r["WRITE_SIZE_KB"]) / 1024.0
        s = sq.get(k, {})
        r.update({c: s[c] for c in s})
        if s.get("SQ_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES is summed over the shader engines' SQs; SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs (cycles).
            # 1024 SIMDs: utilisation of the chip's matrix pipes over the kernel's busy time
            r["mfma_busy_over_wave_cycles"] = s.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * s["SQ_WAVE_CYCLES"]) if s.get("SQ_WAVE_CYCLES") else None
        if k in dur:
            r["avg_duration_us"] = dur[k] / 1e3
            if s.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
                # matrix-pipe utilisation: busy cycles summed over the chip's 1024 SIMD matrix pipes / (duration x clock x 1024)
                r["mfma_pipe_util"] = s["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur[k] * GHZ * SIMDS)
            if r.get("hbm_side_MB") is not None:
                r["hbm_side_TBps"] = r["hbm_side_MB"] * 1.048576e6 / (dur[k] * 1e-9) / 1e12
        if s.get("SQ_WAVE_CYCLES"):
            r["wait_any_frac"] = s.get("SQ_WAIT_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
            r["wait_inst_frac"] = s.get("SQ_WAIT_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
            r["active_frac"] = s.get("SQ_ACTIVE_INST_ANY", 0.0) / s["SQ_WAVE_CYCLES"]
        rows.append(r)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from xingtian_amd.lib import built_sources_sha
    # the digest embedded in the library that RAN (xt_build_sources_sha), not the digest of the tree next to it
    doc = {"kernel_sources_sha": built_sources_sha(),
           "note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_* in separate runs) over tools/step_probe.py (eager updates of the three workloads); "
                   "mean per launch; hbm_side_MB = (2 x FETCH_SIZE_KB + WRITE_SIZE_KB) / 1024 (gfx950 correction for 16-byte "
                   "coalesced reads; L2 memory-side requests incl. Infinity-Cache hits).  SQ_VALU_MFMA_BUSY_CYCLES counts "
                   "cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md constants "
                   "table): mfma_busy_over_wave_cycles = MFMA-busy cycles / (4 x wave quad-cycles) = fraction of the time "
                   "waves were resident during which their SIMD's matrix pipe was busy, per resident wave.  mfma_pipe_util = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (in-graph average duration x 2.4 GHz x 1024 SIMDs) -- the fraction of the chip's "
                   "matrix-pipe time the kernel keeps busy (an fp32 32x32x2 MFMA holds its pipe 64 cycles, a bf16 32x32x16 MFMA "
                   "32); hbm_side_TBps = hbm_side_MB / duration",
           "kernels": rows}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    for r in rows:
        print("%7.1f MB %5s TB/s  %6s us  mfma-pipe %5s  wait %4s inst-wait %4s  %s" % (
            r.get("hbm_side_MB") or -1, ("%.2f" % r["hbm_side_TBps"]) if "hbm_side_TBps" in r else "-",
            ("%.2f" % r["avg_duration_us"]) if "avg_duration_us" in r else "-",
            ("%.3f" % r["mfma_pipe_util"]) if r.get("mfma_pipe_util") is not None else "-",
            ("%.2f" % r["wait_any_frac"]) if "wait_any_frac" in r else "-", ("%.2f" % r["wait_inst_frac"]) if "wait_inst_frac" in r else "-",
            r["kernel"][:100]))


if __name__ == "__main__":
    main()
