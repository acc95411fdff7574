#!/usr/bin/env python3
"""Analysis only (GPU box): kernel time of the deblocking stage of several library builds in ONE process, alternating.
usage: tools/db_ab.py [--frames N] [--rounds R] lib1 lib2 ...   (names without '/' are looked up in dryv_amd/lib/var/)"""
import argparse, ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dryv_amd import abi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=10)
    a = ap.parse_args()
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=a.frames)
    dp = abi.make_deblock_params(0, 0, 0)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_y = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    ctxs = []
    for p in a.libs:
        path = p if "/" in p else os.path.join(ROOT, "dryv_amd", "lib", "var", p if p.endswith(".so") else p + ".so")
        abi._preload_torch_hip_runtime()
        lib = C.CDLL(path)
        lib.dryv_recon_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        lib.dryv_recon_submit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dryv_recon_deblock_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        lib.dryv_recon_sync.argtypes = [C.c_void_p]
        lib.dryv_recon_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        h = C.c_void_p()
        assert lib.dryv_recon_create(C.byref(h), 0) == 0
        ctxs.append((os.path.basename(path)[:-3], lib, h))
    name0, lib0, h0 = ctxs[0]
    assert lib0.dryv_recon_submit_device(h0, C.byref(fp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()), C.c_void_p(d_y.data_ptr())) == 0
    assert lib0.dryv_recon_sync(h0) == 0
    recon = d_y.clone()
    res = {name: [] for name, _, _ in ctxs}
    for r in range(a.rounds + 2):
        order = ctxs if r % 2 == 0 else ctxs[::-1]
        for name, lib, h in order:
            d_y.copy_(recon)
            torch.cuda.synchronize()
            assert lib.dryv_recon_deblock_device(h, C.byref(fp), C.byref(dp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_y.data_ptr())) == 0
            assert lib.dryv_recon_sync(h) == 0
            ms = C.c_float()
            assert lib.dryv_recon_last_kernel_ms(h, C.byref(ms)) == 0
            if r >= 2:
                res[name].append(ms.value)
    base = statistics.median(res[ctxs[0][0]])
    for name, _, _ in ctxs:
        v = res[name]
        print("%-14s median %.4f  min %.4f  max %.4f   vs %s: %.4f" % (name, statistics.median(v), min(v), max(v), ctxs[0][0], statistics.median(v) / base))


if __name__ == "__main__":
    main()
