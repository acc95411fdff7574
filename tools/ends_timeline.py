#!/usr/bin/env python3
"""Diagnostic: which wave of a team finishes a band task when (-DDRYV_BAND_TIMELINE -DDRYV_BAND_TLENDS: per task the 100 MHz
stamps of FRONT's, BACK's and CHROMA's last step and of CHROMA's first). Never used by tests, bench or the product.
usage: ends_timeline.py [workload] [frames]"""
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
    so = os.path.join(_build.LIB, "libdryv_recon_bte.so")
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-DDRYV_BAND_TIMELINE", "-DDRYV_BAND_TLENDS", "-o", so] + srcs)
    import torch
    lib = abi.load_library(so)
    fp, mbs, co, n = synth.workload(wl, n_frames=int(args[0]) if args else None)
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
    nb = (fp.pic_height_in_mbs + 3) // 4
    tasks = n * nb
    out = np.zeros((tasks, 4), dtype=np.uint64)
    assert lib.dryv_recon_debug_band_timeline(h, C.c_int(tasks), out.ctypes.data_as(C.c_void_p)) == 0
    t = out.astype(np.float64) / 100.0
    t -= t[:, 3].min()
    front, back, chroma, cstart = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    band = np.arange(tasks) // n
    print("== %s, %d frames: kernel %.3f ms, %d tasks; last FRONT / BACK / CHROMA end at %.1f / %.1f / %.1f us" % (
        wl, n, ms.value, tasks, front.max(), back.max(), chroma.max()))
    print("per task: BACK ends %.1f us after FRONT (p50 %.1f); CHROMA ends %.1f us after BACK (p10 %.1f, p50 %.1f, p90 %.1f); CHROMA's task takes %.1f us, from its first step" % (
        (back - front).mean(), np.median(back - front), (chroma - back).mean(), *np.percentile(chroma - back, [10, 50, 90]), (chroma - cstart).mean()))
    for b in range(0, nb, max(nb // 9, 1)):
        m = band == b
        print("  band %2d: CHROMA starts %7.1f, FRONT / BACK / CHROMA end %7.1f / %7.1f / %7.1f us" % (
            b, cstart[m].mean(), front[m].mean(), back[m].mean(), chroma[m].mean()))


if __name__ == "__main__":
    main()
