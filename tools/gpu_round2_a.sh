#!/bin/bash
# first GPU contact of the band kernel: parity suite (band kernel serves every batch without 8x8 transform), then bench
set -o pipefail
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.log
tail -5 gpurun_out/r2a/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2a/bench_band.json 2> gpurun_out/r2a/bench_band.err; echo "bench band rc=$?"
cat gpurun_out/r2a/bench_band.json | cut -c1-900
DRYV_RECON_KERNEL=row timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2a/bench_row.json 2> gpurun_out/r2a/bench_row.err; echo "bench row rc=$?"
cat gpurun_out/r2a/bench_row.json | cut -c1-600
