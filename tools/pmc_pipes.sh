#!/bin/bash
# Which pipe of the CU is busy? (run on the GPU box from the repo root)  usage: tools/pmc_pipes.sh <outdir>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m ${1:-$R/gpurun_out/pipes})
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$tag.log 2>&1 || echo "set failed: $set"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "recon_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = sum(acc[k]) / len(acc[k])
        print("%-24s %16.0f   %10.1f per MB" % (k, v, v / 2448000.0))
PY
