#!/bin/bash
# three-wave teams: tests, bench, phase shares, grid-shape variants
set -o pipefail
mkdir -p gpurun_out/r2f
timeout -k 10 500 python -m pytest tests -m gpu -x -q --timeout 90 > gpurun_out/r2f/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"
tail -3 gpurun_out/r2f/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2f/bench_band.json 2> gpurun_out/r2f/bench_band.err || { echo bench failed; tail -5 gpurun_out/r2f/bench_band.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2f/bench_band.json'))
print("band: value %.4g MB/s, kernel_ms %.3f, frac %.3f" % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))
PY
timeout -k 10 300 python tools/band_phases.py 1 300 > gpurun_out/r2f/phases.txt 2>&1 || { echo phases failed; tail -5 gpurun_out/r2f/phases.txt; exit 1; }
cat gpurun_out/r2f/phases.txt
bash tools/band_variants.sh \
  "-DDRYV_BAND_TEAMS=3 -DDRYV_BAND_WGS_PER_CU=3 -DDRYV_BAND_WPS=7" \
  "-DDRYV_BAND_TEAMS=4 -DDRYV_BAND_WGS_PER_CU=2 -DDRYV_BAND_WPS=6" \
  "-DDRYV_BAND_TEAMS=5 -DDRYV_BAND_WGS_PER_CU=2 -DDRYV_BAND_WPS=8" \
  "-DDRYV_BAND_TEAMS=2 -DDRYV_BAND_WGS_PER_CU=5 -DDRYV_BAND_WPS=8" \
  "-DDRYV_BAND_TEAMS=2 -DDRYV_BAND_WGS_PER_CU=4 -DDRYV_BAND_WPS=6" \
  "-DDRYV_BAND_TEAMS=2 -DDRYV_BAND_WGS_PER_CU=3 -DDRYV_BAND_WPS=5" \
  "-DDRYV_BAND_TEAMS=1 -DDRYV_BAND_WGS_PER_CU=10 -DDRYV_BAND_WPS=8"
