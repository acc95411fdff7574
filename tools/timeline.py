#!/usr/bin/env python3
"""Occupancy over time of one launch (diagnostic build): when does each macroblock row start and end?

    python tools/timeline.py [frames]
Builds lib/libdryv_recon_prof.so with -DDRYV_PHASE_PROFILE, runs the C2 workload and prints, for 40 equal time
slices of the kernel, how many rows (= waves) were being processed, plus the ramp and the tail.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dryv_amd import _build, abi, synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    so = os.path.join(_build.LIB, "libdryv_recon_prof.so")
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_kernel.hip", "recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDRYV_PHASE_PROFILE", "-o", so] + srcs)
    import torch
    lib = abi.load_library(so)
    lib.dryv_recon_debug_phases.restype = C.c_int
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    h = C.c_void_p()
    assert lib.dryv_recon_create(C.byref(h), 0) == 0
    for _ in range(3):
        assert lib.dryv_recon_submit_device(h, C.byref(fp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()),
                                            C.c_void_p(d_o.data_ptr())) == 0
        assert lib.dryv_recon_sync(h) == 0
    ms = C.c_float()
    lib.dryv_recon_last_kernel_ms(h, C.byref(ms))
    H = fp.pic_height_in_mbs
    rows = n * ((H + 3) // 4) * 4
    out = np.zeros((16384 + rows, 10), dtype=np.uint64)
    assert lib.dryv_recon_debug_phases(h, C.byref(fp), C.c_uint32(n), C.c_int(16384 + rows), out.ctypes.data_as(C.c_void_p)) == 0
    tl = out[16384:, :3].astype(np.int64)
    tl = tl[tl[:, 1] > 0]
    t0, t1 = tl[:, 0].min(), tl[:, 1].max()
    span = float(t1 - t0)
    print("kernel %.3f ms, %d rows, %d ticks (%.1f ticks/us)" % (ms.value, len(tl), span, span / (ms.value * 1e3)))
    bins = 40
    edges = np.linspace(t0, t1, bins + 1)
    act = np.zeros(bins)
    for b in range(bins):
        lo, hi = edges[b], edges[b + 1]
        ov = np.clip(np.minimum(tl[:, 1], hi) - np.maximum(tl[:, 0], lo), 0, None)
        act[b] = ov.sum() / (hi - lo)
    print("rows in flight per 1/40 of the kernel:")
    print(" ".join("%5d" % a for a in act[:20]))
    print(" ".join("%5d" % a for a in act[20:]))
    dur = (tl[:, 1] - tl[:, 0]) / (span / (ms.value * 1e3))
    print("row duration us: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f" % (dur.min(), np.percentile(dur, 10),
          np.median(dur), np.percentile(dur, 90), dur.max()))
    print("mean rows in flight %.0f of 8192 slots" % act.mean())
    lib.dryv_recon_destroy(h)


if __name__ == "__main__":
    main()
