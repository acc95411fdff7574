#!/bin/bash
# copies the evidence tools/gpu_final.sh left in gpurun_out/ into profiles/<round>/ and refreshes
# profiles/hbm_traffic.json. Usage: tools/collect_profiles.sh r02
set -e
R=${1:-r04}; F=gpurun_out/final; P=profiles/$R; mkdir -p $P
cp $F/timeline.txt $P/band_timeline.txt
cp $F/bench_default.json $P/bench_default_line.json
for c in c2 c3; do
  cp gpurun_out/profile_${R}_$c/bench_line.json $P/${c}_bench_line.json
  cp gpurun_out/profile_${R}_$c/kernel_stats.csv $P/${c}_kernel_stats.csv
  cp gpurun_out/profile_${R}_$c/summary.json $P/${c}_summary.json
done
cp $F/bench_c3.json $P/c3_bench_line_with_cpu_baseline.json
cp $F/bench_2ranks_gloo_one_gpu.json $P/
cp $F/phases_c3.txt $P/team_phases_c3.txt
cp $F/frames_sweep.txt $P/frames_sweep.txt
[ -f $F/chain_pace.txt ] && cp $F/chain_pace.txt $P/chain_pace.txt
cp $F/deblock_rate.json $F/host_path.json $F/pack_rate.json $F/stream_rate.json $P/
cp $F/mem_pattern.txt $P/mem_pattern_microbench.txt
cp $F/phases.txt $P/team_phases.txt
cp $F/valu_rate.txt $P/valu_issue_microbench.txt
python3 tools/update_traffic.py $P
