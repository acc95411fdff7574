#!/bin/bash
# Collects the judged evidence for the current kernels on the GPU box (run from the repo root):
#   rocprofv3 --kernel-trace --stats of the default bench command (same pre-roll / warm-up / steps as the driver's run),
#   FETCH_SIZE / WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md: TCC has 4 slots, FETCH_SIZE costs 3,
#   WRITE_SIZE 2), and two SQ instruction / wait counter passes.
# usage: tools/profile_round.sh <tag> [workload]    -> gpurun_out/profile_<tag>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-cur}
WL=${2:-C2_1080p_intra_4x4}
OUT=$(realpath -m $R/gpurun_out/profile_$TAG)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline --no-verify"
$B > $OUT/bench_line.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- $B > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/insts -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/insts.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/waits -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/waits.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT --kernel-trace --output-format csv -d $OUT/active -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/active.log 2>&1
python3 - $OUT $R $WL <<'PY'
import csv, glob, json, sys, os
out, root, wl = sys.argv[1:4]
sys.path.insert(0, root)
line = json.loads(open(out + "/bench_line.json").read().strip().splitlines()[-1])
kname = line["roofline"]["kernel"]
rows = list(csv.DictReader(open(glob.glob(out + "/kt/*/*_kernel_stats.csv")[0])))
k = [r for r in rows if kname in r["Name"]][0]
# the timed region only: the last `steps` dispatches of the kernel in the trace
tr = [r for r in csv.DictReader(open(glob.glob(out + "/kt/*/*_kernel_trace.csv")[0])) if kname in r["Kernel_Name"]]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr[-20:]]
summ = {"command": "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload %s --steps 20 --warmup 5 --no-cpu-baseline --no-verify" % wl,
        "kernel": k["Name"][:80], "calls_incl_preroll_and_warmup": int(k["Calls"]), "avg_ns_all_calls": float(k["AverageNs"]),
        "min_ns": int(k["MinNs"]), "max_ns": int(k["MaxNs"]), "percentage_of_gpu_time": float(k["Percentage"]),
        "timed_calls": len(last), "timed_calls_avg_ns": sum(last) / len(last),
        "bench_line_same_box_unprofiled": {"kernel_ms_avg": line["roofline"]["kernel_ms_avg"], "frac": line["roofline"]["frac"],
                                           "value": line["value"]}}
mbs = line["config"]["macroblocks_per_step"]
def counters(name):
    f = glob.glob(out + "/%s/*/*_counter_collection.csv" % name)[0]
    rr = [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
    acc = {}
    for r in rr:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {c: sum(v) / len(v) for c, v in acc.items()}, rr[0]
for name in ("fetch", "write"):
    c, r0 = counters(name)
    summ[name.upper() + "_SIZE_KB_per_launch_raw"] = list(c.values())[0]
    summ.update(vgpr=int(r0["VGPR_Count"]), sgpr=int(r0.get("SGPR_Count", 0) or 0), scratch=int(r0["Scratch_Size"]),
                lds=int(r0["LDS_Block_Size"]), workgroup=int(r0["Workgroup_Size"]), grid=int(r0["Grid_Size"]))
ci, _ = counters("insts")
cw, _ = counters("waits")
summ["per_macroblock"] = {c: v / mbs for c, v in ci.items()}
ca, _ = counters("active")
# SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md, cycle constants); per macroblock
summ["active_quad_cycles_per_macroblock"] = {c: v / mbs for c, v in ca.items()}
summ["wave_cycle_shares"] = {c: v for c, v in cw.items()}
fetch = summ["FETCH_SIZE_KB_per_launch_raw"] * 1024
write = summ["WRITE_SIZE_KB_per_launch_raw"] * 1024
summ["hbm_bytes_per_launch_guide_rule"] = 2 * fetch + write   # FETCH_SIZE doubled (gfx950: 16 B/lane streams tallied at half)
summ["algorithmic_bytes_per_launch"] = line["roofline"]["algorithmic_bytes_per_launch"]
summ["kernel_source_sha"] = line["roofline"]["kernel_source_sha"]
json.dump(summ, open(out + "/summary.json", "w"), indent=1)
with open(out + "/kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(rows[0].keys())
    for r in rows:
        r = dict(r); r["Name"] = r["Name"][:100]; w.writerow(r.values())
print(json.dumps(summ, indent=1))
PY
