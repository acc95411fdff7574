#!/usr/bin/env python3
"""Rate of the output stage (dryv_recon_pack_device) on the bench geometry: 300 x 1080p coded pictures cropped to
1920 x 1080, as I420 and as NV12. Pure byte moving: the figure to compare with is HBM bandwidth (bytes read + written).
usage: pack_rate.py [frames] [--out file.json]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dryv_amd import abi  # noqa: E402
from dryv_amd.frame import ReconContext  # noqa: E402


def main():
    import torch
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    fp = abi.make_frame_params(120, 68)
    src = torch.randint(0, 256, (frames * 1920 * 1088 * 3 // 2,), dtype=torch.uint8, device="cuda")
    res = []
    with ReconContext(0) as ctx:
        for name, fmt in (("I420", abi.OUT_I420), ("NV12", abi.OUT_NV12)):
            od = abi.make_output_desc(fmt, (0, 0, 0, 8))
            per = 1920 * 1080 * 3 // 2
            dst = torch.zeros(frames * per, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            for _ in range(3):
                ctx.pack_device(fp, frames, src.data_ptr(), od, dst.data_ptr())
            ctx.sync()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.pack_device(fp, frames, src.data_ptr(), od, dst.data_ptr())
            ctx.sync()
            dt = (time.perf_counter() - t0) / reps
            moved = 2 * frames * per   # bytes read + written
            # spot check against numpy on the first frame
            s0 = src[:1920 * 1088 * 3 // 2].cpu().numpy()
            Y = s0[:1920 * 1088].reshape(1088, 1920)[:1080]
            Cb = s0[1920 * 1088:1920 * 1088 + 960 * 544].reshape(544, 960)[:540]
            Cr = s0[1920 * 1088 + 960 * 544:].reshape(544, 960)[:540]
            want = np.concatenate([Y.reshape(-1)] + ([np.stack([Cb, Cr], -1).reshape(-1)] if fmt else [Cb.reshape(-1), Cr.reshape(-1)]))
            ok = bool(np.array_equal(dst[:per].cpu().numpy(), want))
            res.append({"format": name, "frames": frames, "ms": dt * 1e3, "GBps_read_plus_written": moved / dt / 1e9,
                        "macroblocks_per_s": frames * 8160 / dt, "first_frame_matches_numpy": ok})
            print("%s: %.3f ms per %d frames, %.0f GB/s (read + written), %.2f G macroblocks/s, check %s" %
                  (name, dt * 1e3, frames, moved / dt / 1e9, frames * 8160 / dt / 1e9, ok))
    if out_path:
        json.dump({"geometry": "1920x1088 coded -> 1920x1080", "results": res}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
