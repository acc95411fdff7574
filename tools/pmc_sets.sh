#!/bin/bash
# Generic PMC collection over bench.py (run on the GPU box from the repo root).
#   tools/pmc_sets.sh <outdir> "<set 1 counters>" "<set 2 counters>" ...
# One rocprofv3 --pmc pass per set (never combined with the trace domains gpurun refuses); prints per-MB averages.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/set$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/set$i.log 2>&1 || echo "set failed: $set"
done
python3 - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "recon_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        v = sum(acc[k]) / len(acc[k])
        print("%-24s %16.0f   %10.2f per MB" % (k, v, v / 2448000.0))
PY
