#!/usr/bin/env python3
"""Tuning aid: pace of the band kernel's steps as a function of how many bands run and how many of them follow another
band of the same picture. Pictures of W x H macroblocks with H = 4 (one band: nobody to follow), 8, 16, 68 rows; frames
chosen so that every band task is resident at once (<= 2048 teams). Prints kernel ms and microseconds per step of the
critical path (W + 2 * 3 steps per band + 8 per band below the first).
Usage: tools/chain_pace.py [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dryv_amd
from dryv_amd import synth

W = int(sys.argv[1]) if len(sys.argv) > 1 else 120
dev = torch.device("cuda", 0)
ctx = dryv_amd.ReconContext(0)
cfg = synth.config(**synth.WORKLOADS["C2_1080p_intra_4x4"][5])


def run(h, frames, reps=12):
    fp = dryv_amd.make_frame_params(W, h)
    mbs, co = synth.generate(fp, cfg, 7, 0, frames)
    d_mbs = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).to(dev)
    d_co = torch.from_numpy(co).to(dev)
    d_out = torch.zeros(frames * W * h * 384, dtype=torch.uint8, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.04:
        ctx.submit_device(fp, frames, d_mbs.data_ptr(), d_co.data_ptr(), d_out.data_ptr())
        ctx.sync()
    ms = []
    for _ in range(reps):
        ctx.submit_device(fp, frames, d_mbs.data_ptr(), d_co.data_ptr(), d_out.data_ptr())
        ctx.sync()
        ms.append(ctx.last_kernel_ms())
    return float(np.median(ms))


print("W = %d" % W)
for h, frames_list in ((4, (1, 64, 256, 512, 1024, 1536, 2048)), (8, (1, 128, 512, 1024)), (16, (1, 64, 256, 512)),
                       (68, (1, 15, 60, 120))):
    bands = (h + 3) // 4
    steps = W + 6 + 8 * (bands - 1)
    for frames in frames_list:
        ms = run(h, frames)
        print("H %3d (%2d bands) frames %5d  teams busy %5d  kernel %.3f ms  = %.2f us per step of the critical path (%d steps)"
              % (h, bands, frames, frames * bands, ms, 1000 * ms / steps, steps), flush=True)
ctx.close()
