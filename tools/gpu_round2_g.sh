#!/bin/bash
# tuning: team-count sweep (1 team per workgroup, grid = number of teams) + issue counters
set -o pipefail
mkdir -p gpurun_out/r2g gpurun_out/variants
so=dryv_amd/lib/libdryv_recon_var.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDRYV_BAND_TEAMS=1 -DDRYV_BAND_WGS_PER_CU=12 -DDRYV_BAND_WPS=8 -o $so dryv_amd/csrc/recon_kernel.hip dryv_amd/csrc/recon_band.hip dryv_amd/csrc/recon_api.hip 2>/dev/null || exit 1
for g in 1280 1536 1700 1792 2048 2304 2550 2816 3072; do
  DRYV_RECON_LIB=$so DRYV_RECON_GRID=$g timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/variants/teams_$g.json 2>gpurun_out/variants/teams_$g.err || { echo "grid $g failed"; exit 1; }
  python - "$g" <<'PY'
import json,sys
d=json.load(open('gpurun_out/variants/teams_%s.json'%sys.argv[1]))
print("teams %-6s kernel_ms %.3f frac %.3f" % (sys.argv[1], d['roofline']['kernel_ms_avg'], d['roofline']['frac']), flush=True)
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/r2g/avail.txt 2>&1 || true
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY"; do
  n=$(echo "$set" | cut -c1-24 | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r2g/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $R/gpurun_out/r2g/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python - <<'PY'
import csv,glob,collections
for f in glob.glob('gpurun_out/r2g/pmc_*/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'band_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    # a launch is one dispatch: sum over rows with same dispatch id is already done per counter? print mean per dispatch
    for k,v in sorted(acc.items()): print(f.split('/')[2], k, "n=%d mean=%.4g"%(len(v), sum(v)/len(v)))
PY
