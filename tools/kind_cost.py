#!/usr/bin/env python3
"""Kernel time of 300 x 1080p batches made of ONE macroblock kind each (and the two BASELINE mixes): what an Intra4x4,
Intra8x8 and Intra16x16 macroblock costs.   python tools/kind_cost.py [frames]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dryv_amd import abi, synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    lib = abi.load_library()
    h = C.c_void_p()
    assert lib.dryv_recon_create(C.byref(h), 0) == 0
    for name, kw in (("Intra16x16 only", dict(i4x4=0.0, i8x8=0.0)), ("Intra4x4 only", dict(i4x4=1.0, i8x8=0.0)),
                     ("Intra8x8 only", dict(i4x4=0.0, i8x8=1.0)), ("C2 mix 70/0/30", dict(i4x4=0.7, i8x8=0.0)),
                     ("C3 mix 35/40/25", dict(i4x4=0.35, i8x8=0.40))):
        fp = abi.make_frame_params(120, 68, transform_8x8=kw["i8x8"] > 0)
        mbs, co = synth.generate(fp, synth.config(**kw), 2, 0, frames)
        d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
        d_c = torch.from_numpy(co).cuda()
        d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
        best = 1e9
        for _ in range(8):
            assert lib.dryv_recon_submit_device(h, C.byref(fp), frames, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()),
                                                C.c_void_p(d_o.data_ptr())) == 0
            assert lib.dryv_recon_sync(h) == 0
            ms = C.c_float()
            lib.dryv_recon_last_kernel_ms(h, C.byref(ms))
            best = min(best, ms.value)
        print("%-18s %.3f ms  %.3f G MB/s" % (name, best, mbs.size / best / 1e6), flush=True)
    lib.dryv_recon_destroy(h)


if __name__ == "__main__":
    main()
