#!/bin/bash
# Round-end evidence on one box: full gpu test suite, bench line with cpu baseline, rocprofv3 summaries for C2 and C3,
# phase shares, band timeline, host path rates, VALU issue microbenchmark. Everything lands in gpurun_out/final/.
set -o pipefail
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; tail -5 $O/bench_default.err; exit 1; }
echo "bench ok"
timeout -k 10 600 bash tools/profile_round.sh r02_c2 > $O/profile_c2.log 2>&1 || { echo "profile c2 failed"; tail -5 $O/profile_c2.log; exit 1; }
echo "profile c2 ok"
timeout -k 10 600 bash tools/profile_round.sh r02_c3 C3_4k_intra_8x8 > $O/profile_c3.log 2>&1 || { echo "profile c3 failed"; tail -5 $O/profile_c3.log; exit 1; }
echo "profile c3 ok"
timeout -k 10 300 python tools/band_phases.py 1 300 > $O/phases.txt 2>&1 || { echo "phases failed"; tail -5 $O/phases.txt; exit 1; }
timeout -k 10 300 python tools/band_timeline.py 300 > $O/timeline.txt 2>&1 || { echo "timeline failed"; tail -5 $O/timeline.txt; exit 1; }
timeout -k 10 300 python tools/host_path_rate.py --frames 100 --reps 4 --out $O/host_path.json > $O/host_path.log 2>&1 || { echo "host path failed"; tail -5 $O/host_path.log; exit 1; }
timeout -k 10 200 python tools/pack_rate.py 300 --out $O/pack_rate.json > $O/pack_rate.log 2>&1 || { echo "pack rate failed"; tail -5 $O/pack_rate.log; exit 1; }
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/micro/valu_rate.hip && timeout -k 5 120 /tmp/valu_rate > $O/valu_rate.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mem_pattern tools/micro/mem_pattern.hip && timeout -k 5 120 /tmp/mem_pattern > $O/mem_pattern.txt 2>&1
timeout -k 10 300 python tools/deblock_rate.py 300 --out $O/deblock_rate.json > $O/deblock_rate.log 2>&1 || { echo "deblock rate failed"; tail -5 $O/deblock_rate.log; exit 1; }
timeout -k 10 600 python tools/stream_rate.py 300 --out $O/stream_rate.json > $O/stream_rate.log 2>&1 || { echo "stream rate failed"; tail -5 $O/stream_rate.log; exit 1; }
echo "all ok"
