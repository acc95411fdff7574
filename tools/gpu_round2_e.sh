#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2e/pytest.log
timeout -k 10 300 python tools/host_path_rate.py --frames 100 --reps 4 --out gpurun_out/r2e/host_path.json > gpurun_out/r2e/host_path.log 2>&1; echo "host path rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2e/host_path.json'))
for p in d['paths']: print("%-70s chunk %-36s %.1f M MB/s  %.1f GB/s %s" % (p['path'], p.get('chunk_frames',''), p['mb_per_s_pcie_inclusive']/1e6, p['effective_copy_GBs'], p.get('same_output_as_path_1','')))
PY
