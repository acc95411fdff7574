#!/usr/bin/env python3
"""End to end from a byte stream: an all-intra 1080p Annex-B stream (written here by the CABAC encoder from the bench
workload's batch) -> host parse of every picture on all cores, straight into page-locked batch buffers -> pipelined
submit (copy-in / reconstruction / copy-out overlapped) -> pictures in host memory, checked against the oracle on a sample.
Reports the stages' rates and the end-to-end one. usage: stream_rate.py [frames] [--out file.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
from dryv_amd import abi, h264, synth  # noqa: E402
from dryv_amd.frame import ReconContext  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    fp, mbs, co, n = synth.workload("C2_1080p_intra_4x4", n_frames=frames)
    per = fp.pic_width_in_mbs * fp.pic_height_in_mbs
    t0 = time.perf_counter()
    stream = h264.encode_stream(fp, n, mbs, co, slice_qp=int(mbs["qp"][0]))
    t_enc = time.perf_counter() - t0
    print("stream: %d pictures, %.1f MB (%.0f bits per macroblock), encoded in %.1f s" %
          (n, len(stream) / 1e6, 8 * len(stream) / (n * per), t_enc), flush=True)
    res = {"frames": n, "stream_bytes": len(stream), "encode_s": t_enc, "nproc": os.cpu_count()}
    with ReconContext(0) as ctx:
        fps, n_slices = h264.stream_params(stream)
        m_pin = ctx.alloc_host((n_slices * per,), abi.MB_DESC_DTYPE)
        c_pin = ctx.alloc_host((n_slices * per, 384), np.int16)
        y_pin = ctx.alloc_host((n_slices * per * 384,), np.uint8)
        for threads in (1, 0):
            if threads == 1 and n > 32:
                # one core: a sample is enough to state the rate
                t0 = time.perf_counter()
                h264.parse_all_islices_into(stream, m_pin, c_pin, max_pictures=16, threads=1)
                dt = time.perf_counter() - t0
                res["parse_1_thread_mb_per_s"] = 16 * per / dt
                print("parse, 1 thread: %.3f M macroblocks/s" % (16 * per / dt / 1e6), flush=True)
                continue
            t0 = time.perf_counter()
            fp2, n2, info = h264.parse_all_islices_into(stream, m_pin, c_pin, threads=threads)
            t_parse = time.perf_counter() - t0
            assert n2 == n and info["tails_ok"] == 1
            res["parse_all_threads_mb_per_s"] = n * per / t_parse
            print("parse, all %d hardware threads: %.2f M macroblocks/s (%.3f s)" % (os.cpu_count(), n * per / t_parse / 1e6, t_parse), flush=True)
        assert np.array_equal(c_pin[:n * per], co)
        for _ in range(2):
            t0 = time.perf_counter()
            ctx.submit_host(fp2, n, m_pin[:n * per], c_pin[:n * per], y_pin[:n * per * 384])
            ctx.sync()
            t_rec = time.perf_counter() - t0
        res["recon_pcie_inclusive_mb_per_s"] = n * per / t_rec
        print("reconstruction from page-locked buffers (PCIe inclusive): %.1f M macroblocks/s (%.3f s)" % (n * per / t_rec / 1e6, t_rec))
        # end to end: parse + reconstruct, one after the other (no overlap between the two stages yet)
        t0 = time.perf_counter()
        h264.parse_all_islices_into(stream, m_pin, c_pin, threads=0)
        ctx.submit_host(fp2, n, m_pin[:n * per], c_pin[:n * per], y_pin[:n * per * 384])
        ctx.sync()
        t_e2e = time.perf_counter() - t0
        res["end_to_end_mb_per_s"] = n * per / t_e2e
        res["end_to_end_pictures_per_s"] = n / t_e2e
        print("stream -> pictures end to end: %.2f M macroblocks/s = %.0f 1080p pictures/s (%.3f s)" % (n * per / t_e2e / 1e6, n / t_e2e, t_e2e))
        k = min(2, n)
        st, want = oracle.reconstruct(fp, k, mbs[:k * per], co[:k * per])
        ok = bool(st == 0 and np.array_equal(y_pin[:k * per * 384], want))
        res["first_pictures_match_oracle"] = ok
        print("first %d pictures equal the oracle: %s" % (k, ok))
        assert ok
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
