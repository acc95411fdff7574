#!/usr/bin/env python3
"""Rate of the deblocking kernel on the bench geometry (300 x 1080p, the bench batch's records, reconstructed pictures
resident in HBM): kernel time from the library's HIP events, macroblocks/s, and the same roofline arithmetic as bench.py
(algorithmic bytes per macroblock: 384 pixels read + 384 written + 16 of the record = 784). Checks the first picture
against the oracle. usage: deblock_rate.py [frames] [--out file.json]"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from dryv_amd import abi, synth  # noqa: E402
from dryv_amd.frame import ReconContext  # noqa: E402


def main():
    import torch
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    dp = abi.make_deblock_params(0, 0, 0)
    d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
    d_c = torch.from_numpy(co).cuda()
    d_y = torch.zeros(n * per * 384, dtype=torch.uint8, device="cuda")
    with ReconContext(0) as ctx:
        ctx.submit_device(fp, n, d_m.data_ptr(), d_c.data_ptr(), d_y.data_ptr())
        ctx.sync()
        recon = d_y.clone()
        times = []
        for rep in range(12):
            d_y.copy_(recon)
            torch.cuda.synchronize()
            ctx.deblock_device(fp, dp, n, d_m.data_ptr(), d_y.data_ptr())
            ctx.sync()
            if rep >= 2:
                times.append(ctx.last_kernel_ms())
        ms = float(np.mean(times))
        got = d_y[:per * 384].cpu().numpy()
        st, want = oracle.deblock(fp, dp, 1, mbs[:per], recon[:per * 384].cpu().numpy())
        ok = bool(st == 0 and np.array_equal(got, want))
    mbps = n * per / (ms * 1e-3)
    res = {"frames": n, "kernel_ms_avg": ms, "kernel_ms_min": float(np.min(times)), "macroblocks_per_s": mbps,
           "algorithmic_bytes_per_macroblock": 784, "achieved_GBps": mbps * 784 / 1e9, "frac_of_8TBps": mbps * 784 / 8e12,
           "first_picture_matches_oracle": ok}
    print("deblock: %.3f ms per %d frames = %.2f G macroblocks/s = %.0f GB/s algorithmic (%.3f of 8 TB/s); first picture == oracle: %s"
          % (ms, n, mbps / 1e9, mbps * 784 / 1e9, mbps * 784 / 8e12, ok))
    assert ok
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
