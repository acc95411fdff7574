#!/usr/bin/env python3
"""Diagnostic: the mode pre-passes of a launch in time (-DDRYV_BAND_TIMELINE -DDRYV_BAND_TLMODES: per band task the 100 MHz
stamps of the pre-pass's start, of the end of its first wait for the band above, of its end, and of FRONT's arrival at the
wait for it). Never used by tests, bench or the product. usage: modes_timeline.py [workload] [frames]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dryv_amd import _build, abi, synth  # noqa: E402


def main():
    wl = "C2_1080p_intra_4x4"
    args = sys.argv[1:]
    if args and args[0].startswith("C"):
        wl, args = args[0], args[1:]
    frames = int(args[0]) if args else None
    so = os.path.join(_build.LIB, "libdryv_recon_btm.so")
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-DDRYV_BAND_TIMELINE", "-DDRYV_BAND_TLMODES", "-o", so] + srcs)
    import torch
    lib = abi.load_library(so)
    fp, mbs, co, n = synth.workload(wl, n_frames=frames)
    frames = n
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    h = C.c_void_p()
    assert lib.dryv_recon_create(C.byref(h), 0) == 0
    for _ in range(4):
        assert lib.dryv_recon_submit_device(h, C.byref(fp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()),
                                            C.c_void_p(d_o.data_ptr())) == 0
        assert lib.dryv_recon_sync(h) == 0
    nb = (fp.pic_height_in_mbs + 3) // 4
    tasks = n * nb
    out = np.zeros((tasks, 4), dtype=np.uint64)
    assert lib.dryv_recon_debug_band_timeline(h, C.c_int(tasks), out.ctypes.data_as(C.c_void_p)) == 0
    t = out.astype(np.float64) / 100.0
    t -= t[:, 0].min()
    start, polled, end, front = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    band = np.arange(tasks) // n
    print("== %d frames: %d tasks" % (frames, tasks))
    print("pre-pass: wait for the band above %.1f us (p50 %.1f, p90 %.1f) | own work %.1f us (p50 %.1f, p90 %.1f)" % (
        (polled - start).mean(), *np.percentile(polled - start, [50, 90]), (end - polled).mean(), *np.percentile(end - polled, [50, 90])))
    w = np.maximum(end - front, 0)
    print("FRONT waits for the pre-pass: %.1f us per task (p50 %.1f, p90 %.1f); it arrives %.1f us after the pre-pass started (p50 %.1f)" % (
        w.mean(), *np.percentile(w, [50, 90]), (front - start).mean(), np.median(front - start)))
    up = np.arange(tasks) - n
    ok = up >= 0
    lag = start[ok] - end[up[ok]]
    print("a pre-pass starts %.1f us after the band above's ended (p10 %.1f, p50 %.1f, p90 %.1f; negative = it has to wait)" % (
        lag.mean(), *np.percentile(lag, [10, 50, 90])))
    for b in range(0, nb, max(nb // 9, 1)):
        m = band == b
        print("  band %2d: pre-pass %6.1f .. %6.1f us (wait %5.1f, work %5.1f), FRONT arrives %6.1f, waits %5.1f" % (
            b, start[m].mean(), end[m].mean(), (polled - start)[m].mean(), (end - polled)[m].mean(), front[m].mean(), w[m].mean()))


if __name__ == "__main__":
    main()
