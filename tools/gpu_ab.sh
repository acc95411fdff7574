#!/bin/bash
# A/B on one box: kernel time of the given libraries, alternating, three rounds
mkdir -p gpurun_out/ab
for r in 1 2 3; do
  for so in "$@"; do
    n=$(basename $so .so)
    DRYV_RECON_LIB=$so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/ab/$n.$r.json 2>gpurun_out/ab/$n.$r.err || { echo "$n failed"; tail -3 gpurun_out/ab/$n.$r.err; exit 1; }
    python - $n $r <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/%s.%s.json'%(sys.argv[1],sys.argv[2])))
print("%-28s round %s kernel_ms %.3f" % (sys.argv[1], sys.argv[2], d['roofline']['kernel_ms_avg']), flush=True)
PY
  done
done
