#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r2d
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r2d/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r2d/bench.json
DRYV_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --frames-per-gpu 40 > gpurun_out/r2d/bench2.json 2> gpurun_out/r2d/bench2.err; echo "bench2 rc=$?"
tail -c 900 gpurun_out/r2d/bench2.json; tail -3 gpurun_out/r2d/bench2.err
