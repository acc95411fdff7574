#!/usr/bin/env python3
"""Diagnostic: where does a band wave spend its cycles?

Builds libdryv_recon_bprof.so with -DDRYV_BAND_PROFILE (s_memtime stamps around each phase of the band kernel's step,
summed per wave into a buffer of their own), runs the C2 workload with the given numbers of frames and prints cycles
per step and phase. The stamps drain the LDS queue and fence overlaps the real kernel has: read shares, not run time
(cdna_hip_programming.md section 7). Never used by tests, bench or the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dryv_amd import _build, abi, synth  # noqa: E402

ROLES = ["FRONT", "BACK", "CHROMA", "BACK8"]   # wave w of a team: role w % 3 (w % 4 in builds with the 8x8 transform; recon_band.hip)
PHASES = {
    "FRONT": ["claim+prologue", "record decode", "luma residuals", "hand-off+prefetch", "record for BACK (incl. wait for a free buffer)",
              "-", "mode pre-pass (8x8 builds: the wait for CHROMA's) (per task, here per step)"],
    "CHROMA": ["task+prologue", "hand-off traffic", "chroma residuals+prefetch", "chroma prediction", "lines+copies+flush",
               "wait for FRONT's mode pre-pass (8x8 builds: the pre-pass) (here per step)"],
    "BACK": ["wait for record", "top border", "intra16x16", "(publish: gone)", "intra4x4 chain", "line+copies+flush",
             "wait for BACK8"],
    "BACK8": ["wait for record", "wait for BACK's write-out of the step before", "top border + four Intra8x8 blocks"],
}


def main():
    wl = "C2_1080p_intra_4x4"
    args = sys.argv[1:]
    if args and args[0].startswith("C"):
        wl, args = args[0], args[1:]
    frame_list = [int(a) for a in args] or [1, 300]
    so = os.path.join(_build.LIB, "libdryv_recon_bprof.so")
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DDRYV_BAND_PROFILE", "-o", so] + srcs)
    import torch
    lib = abi.load_library(so)
    lib.dryv_recon_debug_band_phases.restype = C.c_int
    for frames in frame_list:
        fp, mbs, co, n = synth.workload(wl, n_frames=frames)
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
        tasks = n * ((fp.pic_height_in_mbs + 3) // 4)
        wpw = 3 * lib.dryv_recon_debug_band_teams() if hasattr(lib, "dryv_recon_debug_band_teams") else 9
        n_waves = 256 * 8 * 12
        out = np.zeros((n_waves, 16), dtype=np.uint64)
        assert lib.dryv_recon_debug_band_phases(h, C.c_int(n_waves), out.ctypes.data_as(C.c_void_p)) == 0
        steps = tasks * (fp.pic_width_in_mbs + 6.0)
        print("== %d frames: instrumented kernel %.3f ms, %d band tasks" % (frames, ms.value, tasks))
        wave = np.arange(n_waves)
        i8 = bool(fp.transform_8x8_mode_flag)
        for ri, role in enumerate(ROLES[:4 if i8 else 3]):
            # (builds with the 8x8 transform: one team of four waves per workgroup, in a 12-wave index space)
            tot = out[(wave % 12) == ri if i8 else (wave % wpw) % 3 == ri].sum(axis=0).astype(np.float64)
            print("  %s: %.0f cycles/step" % (role, tot.sum() / steps))
            for name, v in zip(PHASES[role], tot):
                print("      %-30s %8.0f" % (name, v / steps))
        lib.dryv_recon_destroy(h)
        del d_m, d_c, d_o


if __name__ == "__main__":
    main()
