#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 400 python -m pytest tests -m gpu -x -q --timeout 60 > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r2b/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2b/bench_band.json 2> gpurun_out/r2b/bench_band.err; echo "bench band rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2b/bench_band.json'))
print("band: value %.4g MB/s, kernel_ms %.3f, frac %.3f, verified %s" % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d.get('cpu_baseline',{}).get('gpu_output_verified_bit_exact')))
PY
timeout -k 10 300 python tools/band_phases.py 1 8 300 > gpurun_out/r2b/phases.txt 2>&1; echo "phases rc=$?"
cat gpurun_out/r2b/phases.txt
