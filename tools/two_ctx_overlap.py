#!/usr/bin/env python3
"""Analysis only (GPU box): wall time per batch when consecutive batches go to TWO contexts (two streams, two workspaces)
alternately, against one context: how much of a launch's ramp and drain the next launch can fill.
usage: two_ctx_overlap.py [frames] [batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import dryv_amd
from dryv_amd import synth


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    outs = [torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda") for _ in range(2)]
    ctxs = [dryv_amd.ReconContext(0) for _ in range(2)]

    def run(nctx, k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            c = ctxs[i % nctx]
            c.submit_device_queued(fp, n, d_m.data_ptr(), d_c.data_ptr(), outs[i % nctx].data_ptr())
        for c in ctxs[:nctx]:
            c.sync()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / k
    for _ in range(3):
        run(1, 20); run(2, 20)
    for rep in range(3):
        a, b = run(1, K), run(2, K)
        print("one context %.4f ms per batch, two contexts %.4f ms per batch (%.3f)" % (a, b, b / a))
    same = bool(torch.equal(outs[0], outs[1]))
    print("outputs equal:", same)


if __name__ == "__main__":
    main()
