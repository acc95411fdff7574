#!/usr/bin/env python3
"""Analysis only (GPU box): wall time per batch when consecutive batches rotate over N contexts (N streams, N workspaces), each
launch with a grid of G workgroups -- half-size grids on three or more streams keep two launches resident per CU, so that one
launch's ramp and drain run beside another's steady state. usage: multi_ctx_overlap.py [frames] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dryv_amd
from dryv_amd import synth


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 48
    wl = sys.argv[3] if len(sys.argv) > 3 else "C2_1080p_intra_4x4"
    fp, mbs, co, n = synth.workload(wl, n_frames=frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    NMAX = 4
    outs = [torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda") for _ in range(NMAX)]

    def make(nctx, grid):
        if grid: os.environ["DRYV_RECON_GRID"] = str(grid)
        else: os.environ.pop("DRYV_RECON_GRID", None)
        return [dryv_amd.ReconContext(0) for _ in range(nctx)]

    def run(ctxs, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            c = ctxs[i % len(ctxs)]
            c.submit_device_queued(fp, n, d_m.data_ptr(), d_c.data_ptr(), outs[i % len(ctxs)].data_ptr())
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / k

    full = 1280 if "C3" in wl else 512
    configs = [(1, 0), (2, full // 2), (3, full // 2), (4, full // 2), (4, full // 4), (3, full)]
    sets = {cfg: make(*cfg) for cfg in configs}
    for cfg, ctxs in sets.items():
        run(ctxs, 12)
    for rep in range(3):
        print("  ".join("%dctx/g%d %.4f" % (c[0], c[1], run(sets[c], K)) for c in configs))
    ref = outs[0].clone()
    run(sets[(3, full // 2)], 6)
    print("outputs equal:", all(bool(torch.equal(ref, o)) for o in outs[:3]))


main()
