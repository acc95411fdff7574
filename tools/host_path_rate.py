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
    recs = []
    # (1) the simple entry points on pageable memory: dryv_recon_submit + dryv_recon_wait
    times = []
    for r in range(a.reps + 1):
        t0 = time.perf_counter()
        ctx.submit(fp, a.frames, mbs, co)
        yuv = ctx.wait()
        dt = time.perf_counter() - t0
        if r:  # first pass allocates the staging buffers
            times.append(dt)
    ref_sum = int(np.bitwise_xor.reduce(yuv.view(np.uint64)))
    bytes_moved = mbs.nbytes + co.nbytes + yuv.nbytes
    recs.append({"path": "dryv_recon_submit + dryv_recon_wait, pageable host buffers", "seconds_best": min(times),
                 "mb_per_s_pcie_inclusive": n_mbs / min(times), "effective_copy_GBs": bytes_moved / min(times) / 1e9})
    # (2) the pipelined entry point on page-locked memory: dryv_recon_submit_host + dryv_recon_sync
    pm = ctx.alloc_host(mbs.shape, mbs.dtype)
    pc = ctx.alloc_host(co.shape, co.dtype)
    po = ctx.alloc_host((n_mbs * 384,), np.uint8)
    pm[...] = mbs
    pc[...] = co
    for chunk in (None, 6, 12, 25, 50):
        if chunk is None:
            os.environ.pop("DRYV_RECON_CHUNK_FRAMES", None)
        else:
            os.environ["DRYV_RECON_CHUNK_FRAMES"] = str(chunk)
        times = []
        for r in range(a.reps + 1):
            po[:64] = 0
            t0 = time.perf_counter()
            ctx.submit_host(fp, a.frames, pm, pc, po)
            ctx.sync()
            dt = time.perf_counter() - t0
            if r:
                times.append(dt)
        ok = int(np.bitwise_xor.reduce(po.view(np.uint64))) == ref_sum
        recs.append({"path": "dryv_recon_submit_host + dryv_recon_sync, page-locked buffers",
                     "chunk_frames": chunk if chunk else "default (~128 MB of coefficients)", "seconds_best": min(times),
                     "mb_per_s_pcie_inclusive": n_mbs / min(times), "effective_copy_GBs": bytes_moved / min(times) / 1e9,
                     "same_output_as_path_1": ok})
    rec = {"workload": a.workload, "frames": a.frames, "macroblocks": n_mbs, "host_bytes_moved": int(bytes_moved),
           "paths": recs}
    line = json.dumps(rec)
    print(line)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
