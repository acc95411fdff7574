#!/bin/bash
# deblock kernel: rate + instruction counters
mkdir -p gpurun_out/db
timeout -k 10 300 python tools/deblock_rate.py 300 --out gpurun_out/db/rate.json || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $R/gpurun_out/db/pmc -- python3 $R/tools/deblock_rate.py 300 > $R/gpurun_out/db/pmc.log 2>&1 || echo "pmc failed"
cd $R
python - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob('gpurun_out/db/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'deblock_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
mb=300*120*68
for k,v in sorted(acc.items()): print("%-22s per MB %.1f" % (k, sum(v)/len(v)/mb))
PY
