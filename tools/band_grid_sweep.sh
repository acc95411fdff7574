#!/bin/bash
# tuning only: kernel time vs persistent grid size (DRYV_RECON_GRID) on the C2 bench workload
mkdir -p gpurun_out/variants
for g in 1024 896 768 640 512; do
  DRYV_RECON_GRID=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/variants/grid_$g.json 2>gpurun_out/variants/grid_$g.err
  python - "$g" <<'PY'
import json,sys
d=json.load(open('gpurun_out/variants/grid_%s.json'%sys.argv[1]))
print("grid %-6s kernel_ms %.3f frac %.3f" % (sys.argv[1], d['roofline']['kernel_ms_avg'], d['roofline']['frac']), flush=True)
PY
done
