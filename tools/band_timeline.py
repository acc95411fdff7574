#!/usr/bin/env python3
"""Diagnostic: when does each band task run?

Builds libdryv_recon_btl.so with -DDRYV_BAND_TIMELINE (three 100 MHz timestamps per band task: claim, BACK's first and
last step), runs the C2 workload and prints how the launch's time is spent: tasks in flight over time, task duration
by claim round and band, start-up wait of a task behind the band above. Never used by tests, bench or the product.
usage: band_timeline.py [frames] [extra -D flags ...]
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dryv_amd import _build, abi, synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    flags = sys.argv[2:]
    so = os.path.join(_build.LIB, "libdryv_recon_btl.so")
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-DDRYV_BAND_TIMELINE", "-o", so] + flags + srcs)
    import torch
    lib = abi.load_library(so)
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    h = C.c_void_p()
    assert lib.dryv_recon_create(C.byref(h), 0) == 0
    for _ in range(4):
        assert lib.dryv_recon_submit_device(h, C.byref(fp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()),
                                            C.c_void_p(d_o.data_ptr())) == 0
        assert lib.dryv_recon_sync(h) == 0
    ms = C.c_float()
    lib.dryv_recon_last_kernel_ms(h, C.byref(ms))
    nb = 17
    tasks = n * nb
    out = np.zeros((tasks, 4), dtype=np.uint64)
    assert lib.dryv_recon_debug_band_timeline(h, C.c_int(tasks), out.ctypes.data_as(C.c_void_p)) == 0
    t0 = out[:, 0].min()
    claim = (out[:, 0] - t0).astype(np.float64) / 100.0   # us
    first = (out[:, 1] - t0).astype(np.float64) / 100.0
    last = (out[:, 2] - t0).astype(np.float64) / 100.0
    wave = out[:, 3].astype(np.int64)
    end = last.max()
    print("== %d frames %s: kernel %.3f ms; claim..last step spans %.1f us; %d tasks, %d teams" %
          (frames, " ".join(flags), ms.value, end, tasks, len(np.unique(wave))))
    dur = last - claim
    print("task duration (claim -> BACK's last step): mean %.1f us, p10 %.1f, p50 %.1f, p90 %.1f" %
          (dur.mean(), *np.percentile(dur, [10, 50, 90])))
    print("team occupancy: sum of task durations / (teams x span) = %.3f" % (dur.sum() / (len(np.unique(wave)) * end)))
    grp = next((int(f.split('=')[1]) for f in flags if f.startswith('-DDRYV_BAND_TASK_GROUP=')), 1)   # band_kernel.h: band_geo
    t_ = np.arange(tasks)
    g_, r_ = t_ // (grp * n), t_ % (grp * n)
    nb_ = np.minimum(grp, nb - g_ * grp)
    band = g_ * grp + r_ % nb_ if grp > 1 else t_ // n
    print("band: claim mean / first-step mean / last-step mean (us), duration")
    for b in range(nb):
        m = band == b
        print("  %2d  %8.1f %8.1f %8.1f   %7.1f" % (b, claim[m].mean(), first[m].mean(), last[m].mean(), dur[m].mean()))
    # tasks in flight over time
    edges = np.linspace(0, end, 21)
    print("time slice (us): tasks in flight (claimed, not finished)")
    for a, bb in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (a + bb)
        print("  %7.1f  %5d" % (mid, int(((claim <= mid) & (last > mid)).sum())))
    lib.dryv_recon_destroy(h)


if __name__ == "__main__":
    main()
