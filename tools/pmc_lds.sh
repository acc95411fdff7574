#!/bin/bash
# analysis only (GPU box): LDS bank conflicts and wave-cycle shares of the libraries in dryv_amd/lib/var/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${WL:-C2_1080p_intra_4x4}   # WL=C3_4k_intra_8x8 for the 4K batch
OUT=$R/gpurun_out/lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $R/dryv_amd/lib/var/*.so; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 - <<'PY' | tee $OUT/summary.txt
import csv,glob,collections,os
for d in sorted(glob.glob('gpurun_out/lds/pmc_*/')):
    n=os.path.basename(d[:-1])[4:]
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: dur.append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    a={k: sum(v)/len(v) for k,v in acc.items()}
    if not a: continue
    mbs = 100*240*135 if os.environ.get('WL','').startswith('C3') else 2448000
    ns = min(dur) if dur else 0
    # LDS-array cycles per CU against the kernel's duration (2.4 GHz: an upper bound of the clock under load)
    print("%-10s conflict/active %.3f  (active %.0f M = %.2f of 256 CUs x %.3f ms at 2.4 GHz)  wait_any %.2f  active_any %.2f of wave cycles; valu/MB %.1f" % (n, a['SQ_LDS_BANK_CONFLICT']/a['SQ_LDS_IDX_ACTIVE'], a['SQ_LDS_IDX_ACTIVE']/1e6, a['SQ_LDS_IDX_ACTIVE']/(256*ns*2.4) if ns else 0, ns/1e6, a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'], a['SQ_ACTIVE_INST_ANY']/a['SQ_WAVE_CYCLES'], a['SQ_INSTS_VALU']/mbs))
PY
