#!/usr/bin/env python3
"""Kernel-time sweeps for tuning: frames x grid x build variant, device-resident inputs.

    python tools/sweep.py [--frames 240,300,600] [--grid 0,1536,2048] [--variants base,DRYV_NO_WAIT] [--workload C2_1080p_intra_4x4]

A variant with extra -D flags is built as lib/libdryv_recon_var<i>.so. DRYV_NO_WAIT removes every inter-row
wait (results are wrong; the time is the dependency-free bound of the same instruction stream).
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dryv_amd import _build, abi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="300")
    ap.add_argument("--grid", default="0")
    ap.add_argument("--variants", default="base", help="comma list: base or macro names, e.g. base,DRYV_NO_WAIT")
    ap.add_argument("--workload", default="C2_1080p_intra_4x4")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    frames = [int(x) for x in a.frames.split(",")]
    grids = [int(x) for x in a.grid.split(",")]
    import torch
    srcs = [os.path.join(_build.CSRC, f) for f in ("recon_kernel.hip", "recon_band.hip", "output_pack.hip", "deblock.hip", "recon_api.hip")]
    fp, mbs, co, _ = synth.workload(a.workload, n_frames=max(frames))
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    for i, defs in enumerate("" if v == "base" else " ".join("-D" + m for m in v.split("+")) for v in a.variants.split(",")):
        if defs:
            so = os.path.join(_build.LIB, "libdryv_recon_var%d.so" % i)
            subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]
                                  + defs.split() + ["-o", so] + srcs)
            lib = abi.load_library(so)
        else:
            lib = abi.load_library()
        for g in grids:
            if g:
                os.environ["DRYV_RECON_GRID"] = str(g)
            else:
                os.environ.pop("DRYV_RECON_GRID", None)
            h = C.c_void_p()
            assert lib.dryv_recon_create(C.byref(h), 0) == 0
            for n in frames:
                best = 1e9
                for _ in range(a.reps):
                    assert lib.dryv_recon_submit_device(h, C.byref(fp), n, C.c_void_p(d_m.data_ptr()),
                                                        C.c_void_p(d_c.data_ptr()), C.c_void_p(d_o.data_ptr())) == 0
                    assert lib.dryv_recon_sync(h) in (0, abi.DRYV_E_UNSUPPORTED)
                    ms = C.c_float()
                    lib.dryv_recon_last_kernel_ms(h, C.byref(ms))
                    best = min(best, ms.value)
                print("defs=%-22r grid=%5d frames=%5d  %.3f ms  %.3f G MB/s" % (defs, g, n, best, n * per / best / 1e6), flush=True)
            lib.dryv_recon_destroy(h)


if __name__ == "__main__":
    main()
