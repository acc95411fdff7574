#!/bin/bash
# Instruction counts of tuning variants (run on the GPU box from the repo root):
#   tools/pmc_variant.sh <outdir> "<counters>" VARIANT [VARIANT...]     VARIANT = base | MACRO[+MACRO...]
# Builds lib/libdryv_recon_pv<i>.so with the given -D flags and runs bench.py under rocprofv3 --pmc with
# DRYV_RECON_LIB pointing at it. The difference to `base` is what the skipped phase executes.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$(realpath -m $1); CTRS=$2; shift 2
mkdir -p $OUT
i=0
for v in "$@"; do
  i=$((i+1))
  so=$R/dryv_amd/lib/libdryv_recon.so
  if [ "$v" != base ]; then
    so=$R/dryv_amd/lib/libdryv_recon_pv$i.so
    defs=$(echo $v | sed 's/+/ -D/g; s/^/-D/')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $defs -o $so $R/dryv_amd/csrc/recon_kernel.hip $R/dryv_amd/csrc/recon_band.hip $R/dryv_amd/csrc/output_pack.hip $R/dryv_amd/csrc/deblock.hip $R/dryv_amd/csrc/recon_api.hip
  fi
  (cd /tmp && export TMPDIR=/tmp && DRYV_RECON_LIB=$so rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/v$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/v$i.log 2>&1) || echo "failed: $v"
  python3 - $OUT/v$i "$v" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "recon_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-44s " % sys.argv[2] + "  ".join("%s %.1f" % (k.replace("SQ_INSTS_", "").replace("SQ_", ""), sum(v) / len(v) / 2448000.0) for k, v in sorted(acc.items())), flush=True)
PY
done
