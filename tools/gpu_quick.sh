#!/bin/bash
# quick check: gpu tests of the kernel, bench, VALU/SALU/LDS instruction counts per macroblock
set -o pipefail
mkdir -p gpurun_out/quick
timeout -k 10 400 python -m pytest tests/test_recon_gpu.py -x -q --timeout 90 > gpurun_out/quick/pytest.log 2>&1; rc=$?; tail -2 gpurun_out/quick/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err || { tail -5 gpurun_out/quick/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/quick/bench.json'))
print("bench: kernel_ms %.3f frac %.3f value %.4g verified %s" % (d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['value'], d['config'].get('shards_verified_bit_exact')))
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/quick/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $R/gpurun_out/quick/pmc.log 2>&1 || echo "pmc failed"
cd $R
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/quick/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
mb=300*120*68
for k,v in sorted(acc.items()): print("%-22s per MB %.1f" % (k, sum(v)/len(v)/mb))
PY
