#!/bin/bash
# Per-launch SQ instruction counters of recon_kernel on the C2 workload (run on the GPU box, from the repo root).
# usage: tools/pmc_counts.sh <outdir>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m ${1:-$R/gpurun_out/pmc})
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
  --kernel-trace --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
f = glob.glob(out + "/run/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "recon_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
mbs = 2448000.0
for k in sorted(acc):
    v = sum(acc[k]) / len(acc[k])
    print("%-18s %14.0f   %8.1f per MB" % (k, v, v / mbs))
PY
