#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2c
timeout -k 10 120 python - > gpurun_out/r2c/small.log 2>&1 <<'PY'
import sys
sys.path.insert(0, 'tests')
import numpy as np
import dryv_amd, oracle
from dryv_amd import abi, synth
from util import first_mismatch
ctx = dryv_amd.ReconContext(0)
for (W, H, frames, i4) in [(7, 3, 1, 0.0), (7, 5, 1, 0.0), (7, 5, 3, 0.0), (7, 5, 3, 1.0), (12, 9, 4, 0.7), (120, 68, 2, 0.7)]:
    fp = abi.make_frame_params(W, H)
    mbs, co = synth.generate(fp, synth.config(i4x4=i4, i8x8=0.0), 100, 0, frames)
    st, want = oracle.reconstruct(fp, frames, mbs, co)
    for rep in range(3):
        try:
            got = ctx.reconstruct(fp, frames, mbs, co)
            print(W, H, frames, i4, "rep", rep, first_mismatch(got, want, W, H), "kernel ms %.3f" % ctx.last_kernel_ms(), flush=True)
        except Exception as e:
            print(W, H, frames, i4, "rep", rep, "ERROR", e, flush=True)
PY
echo "small rc=$?"; cat gpurun_out/r2c/small.log
