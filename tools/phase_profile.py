#!/usr/bin/env python3
"""Diagnostic: where does a wave of recon_kernel spend its cycles?

Builds libdryv_recon_prof.so with -DDRYV_PHASE_PROFILE (s_memtime stamps around each phase of the
macroblock loop, summed per wave into a buffer of their own), runs the C2 workload once and prints
the share of wave-cycles per phase. The stamps fence overlaps the real kernel has, so read the
SHARES, not the run time (cdna_hip_programming.md §7). Never used by tests, bench or the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dryv_amd import _build, abi, synth  # noqa: E402

PHASES = ["claim", "record/consts", "wait vmcnt(0)", "residuals", "poll row above", "neighbour fetch",
          "chroma pred", "luma pred", "write-out", "row tail"]


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
    n_waves = min(1024, (n * 68 + 7) // 8) * 8
    out = np.zeros((n_waves, 10), dtype=np.uint64)
    assert lib.dryv_recon_debug_phases(h, C.byref(fp), C.c_uint32(n), C.c_int(n_waves), out.ctypes.data_as(C.c_void_p)) == 0
    tot = out.sum(axis=0).astype(np.float64)
    print("instrumented kernel %.3f ms, %d waves, %d macroblocks" % (ms.value, n_waves, mbs.size))
    for name, v in zip(PHASES, tot):
        print("  %-18s %6.2f %%   %8.0f cycles/MB" % (name, 100 * v / tot.sum(), v / mbs.size))
    print("  total %.0f wave-cycles/MB" % (tot.sum() / mbs.size))
    # by position of the wave in its 4-row band (wave 0 reads the band above through L2, 1..3 through the LDS ring)
    for w in range(4):
        sub = out[w::4].sum(axis=0).astype(np.float64)
        share = 100 * sub / sub.sum()
        print("  band wave %d: " % w + "  ".join("%s %.1f%%" % (n.split()[0], v) for n, v in zip(PHASES, share) if v >= 1.0)
              + "   | cycles/MB of this wave: %.0f" % (sub.sum() / (mbs.size / 4)))
    lib.dryv_recon_destroy(h)


if __name__ == "__main__":
    main()
