#!/bin/bash
# Round-end evidence on one box: full gpu test suite, bench line with cpu baseline, rocprofv3 summaries for C2 and C3,
# phase shares, band timeline, host path rates, VALU issue microbenchmark. Everything lands in gpurun_out/final/.
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
echo "bench ok"
timeout -k 10 600 python bench.py --workload C3_4k_intra_8x8 --steps 20 --warmup 5 > $O/bench_c3.json 2> $O/bench_c3.err || { echo "bench c3 failed"; tail -5 $O/bench_c3.err; exit 1; }
echo "bench c3 ok"
DRYV_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 2> $O/bench_2ranks.err | grep '^{' > $O/bench_2ranks_gloo_one_gpu.json || { echo "2-rank rehearsal failed"; tail -5 $O/bench_2ranks.err; exit 1; }
echo "2-rank launcher rehearsal ok"
timeout -k 10 600 bash tools/profile_round.sh r04_c2 > $O/profile_c2.log 2>&1 || { echo "profile c2 failed"; tail -5 $O/profile_c2.log; exit 1; }
echo "profile c2 ok"
timeout -k 10 600 bash tools/profile_round.sh r04_c3 C3_4k_intra_8x8 > $O/profile_c3.log 2>&1 || { echo "profile c3 failed"; tail -5 $O/profile_c3.log; exit 1; }
echo "profile c3 ok"
timeout -k 10 300 python tools/band_phases.py 1 300 > $O/phases.txt 2>&1 || { echo "phases failed"; tail -5 $O/phases.txt; exit 1; }
timeout -k 10 300 python tools/band_timeline.py 300 > $O/timeline.txt 2>&1 || { echo "timeline failed"; tail -5 $O/timeline.txt; exit 1; }
timeout -k 10 300 python tools/band_phases.py C3_4k_intra_8x8 100 > $O/phases_c3.txt 2>&1 || { echo "phases c3 failed"; tail -5 $O/phases_c3.txt; exit 1; }
timeout -k 10 600 bash tools/frames_sweep.sh 60 120 300 600 1200 > $O/frames_sweep.txt 2>&1 || { echo "frames sweep failed"; tail -5 $O/frames_sweep.txt; exit 1; }
timeout -k 10 300 python tools/chain_pace.py > $O/chain_pace.txt 2>&1 || echo "chain pace failed (not fatal)"
timeout -k 10 300 python tools/host_path_rate.py --frames 100 --reps 4 --out $O/host_path.json > $O/host_path.log 2>&1 || { echo "host path failed"; tail -5 $O/host_path.log; exit 1; }
timeout -k 10 200 python tools/pack_rate.py 300 --out $O/pack_rate.json > $O/pack_rate.log 2>&1 || { echo "pack rate failed"; tail -5 $O/pack_rate.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/micro/valu_rate.hip && timeout -k 5 120 /tmp/valu_rate > $O/valu_rate.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mem_pattern tools/micro/mem_pattern.hip && timeout -k 5 120 /tmp/mem_pattern > $O/mem_pattern.txt 2>&1
timeout -k 10 300 python tools/gpu_fuzz.py 120 801 > $O/gpu_fuzz.txt 2>&1 || { echo "gpu fuzz failed"; tail -5 $O/gpu_fuzz.txt; exit 1; }
timeout -k 10 300 python tools/deblock_rate.py 300 --out $O/deblock_rate.json > $O/deblock_rate.log 2>&1 || { echo "deblock rate failed"; tail -5 $O/deblock_rate.log; exit 1; }
timeout -k 10 600 python tools/stream_rate.py 300 --out $O/stream_rate.json > $O/stream_rate.log 2>&1 || { echo "stream rate failed"; tail -5 $O/stream_rate.log; exit 1; }
echo "all ok"
