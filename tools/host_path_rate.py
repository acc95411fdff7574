#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (dryv_recon_submit + dryv_recon_wait): the number a
dryv process holding coefficients in ordinary host memory sees. Never bench.py's `value` (that one has its
inputs resident in HBM); recorded in profiles/rNN/host_path.json and quoted in DESIGN.md §4.

    python tools/host_path_rate.py [--workload C2_1080p_intra_4x4] [--frames 100] [--reps 5] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dryv_amd import abi, synth  # noqa: E402
from dryv_amd.frame import ReconContext  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C2_1080p_intra_4x4")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    fp, mbs, co, _ = synth.workload(a.workload, a.frames)
    n_mbs = a.frames * fp.pic_width_in_mbs * fp.pic_height_in_mbs
    ctx = ReconContext(0)
    times, kern = [], []
    for r in range(a.reps + 1):
        t0 = time.perf_counter()
        ctx.submit(fp, a.frames, mbs, co)
        yuv = ctx.wait()
        dt = time.perf_counter() - t0
        if r:  # first pass allocates the staging buffers
            times.append(dt)
            kern.append(ctx.last_kernel_ms())
    best = min(times)
    bytes_moved = mbs.nbytes + co.nbytes + yuv.nbytes
    rec = {
        "workload": a.workload, "frames": a.frames, "macroblocks": n_mbs, "memory": "pageable host buffers",
        "seconds_best": best, "seconds_all": times, "kernel_ms": kern,
        "mb_per_s_pcie_inclusive": n_mbs / best, "host_bytes_moved": int(bytes_moved),
        "effective_copy_GBs": bytes_moved / best / 1e9,
        "checksum": int(np.bitwise_xor.reduce(yuv.view(np.uint64))),
    }
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
