R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in dbold db32; do
  DRYV_RECON_LIB=$R/dryv_amd/lib/var/$v.so timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/dbpmc/$v -- python3 $R/tools/deblock_rate.py 60 > $R/gpurun_out/dbpmc_$v.log 2>&1 || echo fail $v
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for v in ("dbold","db32"):
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/dbpmc/%s/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'deblock' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    mb=60*120*68
    print(v, {k: round(sum(x)/len(x)/mb,1) for k,x in acc.items()})
PY
