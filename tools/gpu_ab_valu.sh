#!/bin/bash
# A/B of two library builds on one box: kernel time (alternating, three rounds) and vector instructions per macroblock
bash tools/gpu_ab.sh "$@"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$R/$so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/ab/pmc_$n -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $R/gpurun_out/ab/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python - "$@" <<'PY'
import csv,glob,collections,sys,os
for so in sys.argv[1:]:
    n=os.path.basename(so)[:-3]
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/ab/pmc_%s/**/*counter_collection.csv'%n, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    mb=300*120*68
    print(n, {k: round(sum(v)/len(v)/mb,1) for k,v in sorted(acc.items())})
PY
