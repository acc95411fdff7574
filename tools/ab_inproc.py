#!/usr/bin/env python3
"""Analysis only (GPU box): kernel time of several library builds measured in ONE process, alternating build by build
(bursts of BURST queued launches each, ROUNDS rounds): box-to-box and process-to-process spread, which is as large as
most single optimisations, cancels out. Prints per build: median / min of the burst averages, and the median of the
per-round ratio to the first build.
usage: tools/ab_inproc.py [--workload W] [--frames N] [--rounds R] [--burst B] lib1.so lib2.so ...
       (a name without '/' is looked up in dryv_amd/lib/var/)"""
import argparse, ctypes as C, os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dryv_amd import abi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--workload", default="C2_1080p_intra_4x4")
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--burst", type=int, default=10)
    a = ap.parse_args()
    fp, mbs, co, n = synth.workload(a.workload, n_frames=a.frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_o = torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda")
    ctxs = []
    for p in a.libs:
        path = p if "/" in p else os.path.join(ROOT, "dryv_amd", "lib", "var", p if p.endswith(".so") else p + ".so")
        abi._preload_torch_hip_runtime()
        lib = C.CDLL(path)   # (only the five entry points used here: older builds lack newer symbols)
        lib.dryv_recon_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
        lib.dryv_recon_submit_device_queued.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.dryv_recon_sync.argtypes = [C.c_void_p]
        lib.dryv_recon_kernel_ms_stats.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        h = C.c_void_p()
        assert lib.dryv_recon_create(C.byref(h), 0) == 0
        ctxs.append((os.path.basename(path)[:-3], lib, h))

    def burst(lib, h, k):
        for _ in range(k):
            assert lib.dryv_recon_submit_device_queued(h, C.byref(fp), n, C.c_void_p(d_m.data_ptr()), C.c_void_p(d_c.data_ptr()),
                                                       C.c_void_p(d_o.data_ptr())) == 0
        assert lib.dryv_recon_sync(h) == 0
        av, lo, hi = C.c_float(), C.c_float(), C.c_float()
        assert lib.dryv_recon_kernel_ms_stats(h, k, C.byref(av), C.byref(lo), C.byref(hi)) == 0
        return av.value
    t0 = time.time()
    while time.time() - t0 < 0.3:   # clocks up
        for _, lib, h in ctxs:
            burst(lib, h, 4)
    res = {name: [] for name, _, _ in ctxs}
    for r in range(a.rounds):
        order = ctxs if r % 2 == 0 else ctxs[::-1]
        for name, lib, h in order:
            res[name].append(burst(lib, h, a.burst))
    base = res[ctxs[0][0]]
    for name, _, _ in ctxs:
        v = res[name]
        ratio = statistics.median([x / b for x, b in zip(v, base)])
        print("%-16s median %.4f  min %.4f  max %.4f   vs %s: %.4f" % (name, statistics.median(v), min(v), max(v), ctxs[0][0], ratio), flush=True)


if __name__ == "__main__":
    main()
