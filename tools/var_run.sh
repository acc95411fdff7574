#!/bin/bash
# analysis only (GPU box): for every library in dryv_amd/lib/var/ (or those named): kernel time alternating over ROUNDS rounds
# (bench.py, unprofiled) and one PMC pass (SQ_INSTS_VALU / SALU / LDS / BRANCH per macroblock). -> gpurun_out/var/summary.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
ROUNDS=${ROUNDS:-2}
WL=${WL:-C2_1080p_intra_4x4}   # WL=C3_4k_intra_8x8 for the 4K batch
OUT=$R/gpurun_out/var
rm -rf $OUT; mkdir -p $OUT
if [ $# -gt 0 ]; then LIBS=""; for n in "$@"; do LIBS="$LIBS $R/dryv_amd/lib/var/$n.so"; done; else LIBS=$(ls $R/dryv_amd/lib/var/*.so); fi
cd $R
for r in $(seq 1 $ROUNDS); do
  for so in $LIBS; do
    n=$(basename $so .so)
    DRYV_RECON_LIB=$so timeout -k 10 200 python3 bench.py --workload $WL --steps 20 --warmup 5 --no-cpu-baseline ${VERIFY:---no-verify} > $OUT/$n.$r.json 2>$OUT/$n.$r.err || echo "$n round $r failed: $(tail -1 $OUT/$n.$r.err)"
    echo "timed $n $r" >> $OUT/progress.txt
  done
done
cd /tmp && export TMPDIR=/tmp
if [ -z "$NOPMC" ]; then
for so in $LIBS; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/pmc_$n.log 2>&1 || echo "pmc $n failed"
  echo "pmc $n" >> $OUT/progress.txt
done
fi
cd $R
python3 - <<'PY' | tee $OUT/summary.txt
import csv,glob,collections,os,json
res=collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/var/*.json')):
    n,r,_=os.path.basename(f).rsplit('.',2)
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception: continue
    res.setdefault(n,{'ms':[]})['ms'].append(d['roofline']['kernel_ms_avg'])
for d in sorted(glob.glob('gpurun_out/var/pmc_*/')):
    n=os.path.basename(d[:-1])[4:]
    acc=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    mb=100*240*135 if os.environ.get('WL','').startswith('C3') else 300*120*68
    res.setdefault(n,{'ms':[]}).update({k: sum(v)/len(v)/mb for k,v in acc.items()})
for n,v in res.items():
    print("%-14s ms %-28s valu %6.1f salu %6.1f lds %5.1f branch %5.1f" % (n, ' '.join('%.3f'%x for x in v['ms']), v.get('SQ_INSTS_VALU',0), v.get('SQ_INSTS_SALU',0), v.get('SQ_INSTS_LDS',0), v.get('SQ_INSTS_BRANCH',0)))
PY
