#!/bin/bash
# analysis only (GPU box): LDS bank conflicts and wave-cycle shares of the libraries in dryv_amd/lib/var/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lds; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $R/dryv_amd/lib/var/*.so; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 - <<'PY' | tee $OUT/summary.txt
import csv,glob,collections,os
for d in sorted(glob.glob('gpurun_out/lds/pmc_*/')):
    n=os.path.basename(d[:-1])[4:]
    acc=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    a={k: sum(v)/len(v) for k,v in acc.items()}
    if not a: continue
    print("%-10s conflict/active %.3f  (active %.0f M)  wait_any %.2f  active_any %.2f of wave cycles; valu/MB %.1f" % (n, a['SQ_LDS_BANK_CONFLICT']/a['SQ_LDS_IDX_ACTIVE'], a['SQ_LDS_IDX_ACTIVE']/1e6, a['SQ_WAIT_ANY']/a['SQ_WAVE_CYCLES'], a['SQ_ACTIVE_INST_ANY']/a['SQ_WAVE_CYCLES'], a['SQ_INSTS_VALU']/2448000))
PY
