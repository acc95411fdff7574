#!/bin/bash
# analysis only (GPU box): SQ_INSTS_VALU / SALU / LDS per macroblock and kernel time of every library in dryv_amd/lib/var/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ablate
cd /tmp && export TMPDIR=/tmp
for so in $R/dryv_amd/lib/var/*.so; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $R/gpurun_out/ablate/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $R/gpurun_out/ablate/pmc_$n.log 2>&1 || echo "pmc $n failed"
  echo "done $n" >> $R/gpurun_out/ablate/progress.txt
done
cd $R
python3 - <<'PY' | tee gpurun_out/ablate/summary.txt
import csv,glob,collections,os
res={}
for d in sorted(glob.glob('gpurun_out/ablate/pmc_*/')):
    n=os.path.basename(d[:-1])[4:]
    acc=collections.defaultdict(list); dur=[]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
    mb=300*120*68
    res[n]={k: sum(v)/len(v)/mb for k,v in acc.items()}
    res[n]['ms']=min(dur) if dur else 0
full=res.get('full',{})
for n,v in sorted(res.items()):
    print("%-8s valu %6.1f (%+6.1f) salu %6.1f (%+6.1f) lds %5.1f branch %5.1f  ms %.3f" % (n, v.get('SQ_INSTS_VALU',0), v.get('SQ_INSTS_VALU',0)-full.get('SQ_INSTS_VALU',0),
          v.get('SQ_INSTS_SALU',0), v.get('SQ_INSTS_SALU',0)-full.get('SQ_INSTS_SALU',0), v.get('SQ_INSTS_LDS',0), v.get('SQ_INSTS_BRANCH',0), v['ms']))
PY
