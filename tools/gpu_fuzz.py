#!/usr/bin/env python3
"""One-off stress on the GPU box: random geometries, mixes, QPs, scaling lists and filter offsets through reconstruction and
deblocking (device-resident path), every result against the oracles. usage: gpu_fuzz.py [seconds] [seed]
(DRYV_FUZZ_LANES=n: every batch also five times over n queue lanes)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from dryv_amd import abi, synth  # noqa: E402
from dryv_amd.frame import ReconContext  # noqa: E402


def main():
    import torch
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
    t0, n, mbs_total = time.time(), 0, 0
    last = t0
    lanes = int(os.environ.get("DRYV_FUZZ_LANES", "1"))   # > 1: every batch also five times over the queue lanes
    with ReconContext(0) as ctx, ReconContext(0) as qctx:
        if lanes > 1:
            qctx.set_queue_lanes(lanes)
        while time.time() - t0 < budget:
            # (one batch in eight wider than 64 macroblocks: the mode pre-pass then carries a right column from batch to batch)
            W = int(rng.integers(60, 150)) if rng.random() < 0.125 else int(rng.integers(1, 60))
            H, frames = int(rng.integers(1, 40)), int(rng.integers(1, 5))
            t8 = bool(rng.integers(0, 2))
            i8 = float(rng.choice([0.0, 0.3, 0.6])) if t8 else 0.0
            i4 = min(float(rng.choice([0.0, 0.3, 0.7])), 1.0 - i8)
            lo = int(rng.integers(0, 45))
            fkw = dict(cqo_cb=int(rng.integers(-12, 13)), cqo_cr=int(rng.integers(-12, 13)), transform_8x8=t8)
            if rng.random() < 0.4:
                fkw.update(scaling4x4=rng.integers(4, 64, size=(6, 16)))
                if t8:
                    fkw.update(scaling8x8=rng.integers(4, 64, size=(6, 64)))
            skw = dict(i4x4=i4, i8x8=i8, qp=(lo, int(rng.integers(lo, 52))), coded=float(rng.choice([0.2, 0.6, 1.0])),
                       max_level=int(rng.choice([15, 300, 2047])), legal_modes_only=bool(rng.random() < 0.7))
            fp = abi.make_frame_params(W, H, **fkw)
            mbs, co = synth.generate(fp, synth.config(**skw), int(rng.integers(1, 1 << 30)), 0, frames)
            st, want = oracle.reconstruct(fp, frames, mbs, co)
            d_m = torch.from_numpy(mbs.view(np.uint8).reshape(-1)).cuda()
            d_c = torch.from_numpy(co).cuda()
            d_y = torch.zeros(want.size, dtype=torch.uint8, device="cuda")
            ctx.submit_device(fp, frames, d_m.data_ptr(), d_c.data_ptr(), d_y.data_ptr())
            ctx.sync(allow_unsupported=True)
            got = d_y.cpu().numpy()
            assert np.array_equal(got, want), ("recon", W, H, frames, fkw, skw)
            if lanes > 1:
                outs = [torch.zeros(want.size, dtype=torch.uint8, device="cuda") for _ in range(5)]
                torch.cuda.synchronize()
                for o in outs:
                    qctx.submit_device_queued(fp, frames, d_m.data_ptr(), d_c.data_ptr(), o.data_ptr())
                qctx.sync(allow_unsupported=True)
                for o in outs:
                    assert np.array_equal(o.cpu().numpy(), want), ("lanes", W, H, frames, fkw, skw)
            dp = abi.make_deblock_params(int(rng.choice([0, 0, 2])), int(rng.integers(-6, 7)), int(rng.integers(-6, 7)))
            st, wantd = oracle.deblock(fp, dp, frames, mbs, want)
            ctx.deblock_device(fp, dp, frames, d_m.data_ptr(), d_y.data_ptr())
            ctx.sync()
            assert np.array_equal(d_y.cpu().numpy(), wantd), ("deblock", W, H, frames, fkw, skw)
            n += 1
            mbs_total += W * H * frames
            if time.time() - last > 30:
                last = time.time()
                print("... %d batches, %d macroblocks, %.0f s" % (n, mbs_total, last - t0), flush=True)
    print("gpu fuzz ok: %d random batches (%d macroblocks) through reconstruction and deblocking, all bit-exact" % (n, mbs_total))


if __name__ == "__main__":
    main()
