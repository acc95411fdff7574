#!/usr/bin/env python3
"""What a caller that keeps two batches in flight gets (two contexts = two streams, device-resident buffers):
the drain tail of one launch overlaps the ramp of the next. Informational; bench.py's `value` is the
one-launch-at-a-time rate.   python tools/two_stream_rate.py [frames] [launches]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dryv_amd import ReconContext, synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    outs = [torch.zeros(mbs.size * 384, dtype=torch.uint8, device="cuda") for _ in range(2)]
    ctxs = [ReconContext(0), ReconContext(0)]
    n_mbs = mbs.size
    for mode in ("one launch at a time", "two in flight"):
        for c in ctxs:
            c.submit_device(fp, n, d_m.data_ptr(), d_c.data_ptr(), outs[0].data_ptr()); c.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode.startswith("one"):
            for i in range(launches):
                ctxs[0].submit_device(fp, n, d_m.data_ptr(), d_c.data_ptr(), outs[0].data_ptr())
                ctxs[0].sync()
        else:
            for i in range(launches):
                k = i & 1
                if i >= 2:
                    ctxs[k].sync()
                ctxs[k].submit_device(fp, n, d_m.data_ptr(), d_c.data_ptr(), outs[k].data_ptr())
            ctxs[0].sync(); ctxs[1].sync()
        dt = time.perf_counter() - t0
        print("%-22s %7.3f ms per batch   %.3f G MB/s" % (mode, dt / launches * 1e3, n_mbs * launches / dt / 1e9), flush=True)
    assert torch.equal(outs[0], outs[1])


if __name__ == "__main__":
    main()
