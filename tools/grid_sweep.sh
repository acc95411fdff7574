#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for g in 512 480 448 425 400 384; do
  DRYV_RECON_GRID=$g timeout -k 10 120 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g', round(d['roofline']['kernel_ms_avg'],4))"
done
for g in 512 480 448 425 400 384; do
  DRYV_RECON_GRID=$g timeout -k 10 120 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-verify 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g', round(d['roofline']['kernel_ms_avg'],4))"
done
