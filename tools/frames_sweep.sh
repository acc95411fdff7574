#!/bin/bash
# tuning only: kernel time per frame as a function of the frames per launch (rounds of band tasks per resident team)
mkdir -p gpurun_out/fs
for f in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --frames-per-gpu $f > gpurun_out/fs/$f.json 2>gpurun_out/fs/$f.err || { echo "$f failed"; tail -3 gpurun_out/fs/$f.err; exit 1; }
  python - $f <<'PY'
import json,sys
f=int(sys.argv[1]); d=json.load(open('gpurun_out/fs/%d.json'%f))
ms=d['roofline']['kernel_ms_avg']
print("frames %4d  tasks/2048 = %.2f  kernel_ms %.3f  us/frame %.3f  frac %.3f" % (f, f*17/2048.0, ms, 1000*ms/f, d['roofline']['frac']), flush=True)
PY
done
