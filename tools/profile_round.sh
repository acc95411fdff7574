#!/bin/bash
# Collects the judged evidence for the current kernel on the GPU box (run from the repo root):
#   rocprofv3 --kernel-trace --stats of the default bench command, and FETCH_SIZE / WRITE_SIZE in separate
#   PMC passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2).
# usage: tools/profile_round.sh <tag>     -> gpurun_out/profile_<tag>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cur}
OUT=$(realpath -m $R/gpurun_out/profile_$TAG)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(out + "/kt/*/*_kernel_stats.csv")[0])))
k = [r for r in rows if "recon_kernel" in r["Name"]][0]
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline",
        "kernel": "dryv::recon_kernel", "calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]),
        "min_ns": int(k["MinNs"]), "max_ns": int(k["MaxNs"]), "percentage_of_gpu_time": float(k["Percentage"])}
for name in ("fetch", "write"):
    f = glob.glob(out + "/%s/*/*_counter_collection.csv" % name)[0]
    rr = [r for r in csv.DictReader(open(f)) if "recon_kernel" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rr]
    summ[name.upper() + "_SIZE_KB_per_launch_raw"] = sum(vals) / len(vals)
    summ.update(vgpr=int(rr[0]["VGPR_Count"]), scratch=int(rr[0]["Scratch_Size"]), lds=int(rr[0]["LDS_Block_Size"]),
                workgroup=int(rr[0]["Workgroup_Size"]), grid=int(rr[0]["Grid_Size"]))
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
with open(out + "/kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(rows[0].keys())
    for r in rows:
        r = dict(r); r["Name"] = r["Name"][:100]; w.writerow(r.values())
print(json.dumps(summ, indent=1))
PY
