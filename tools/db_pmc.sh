#!/bin/bash
# analysis only (GPU box): SQ counters of the deblocking kernel for the libraries named (dryv_amd/lib/var/<name>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_IFETCH SQ_WAIT_INST_LDS"; do
    n=$(echo $set | md5sum | cut -c1-6)
    DRYV_RECON_LIB=$R/dryv_amd/lib/var/$v.so timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/dbpmc/$v.$n -- python3 $R/tools/deblock_rate.py 300 > $R/gpurun_out/dbpmc_$v.$n.log 2>&1 || echo fail $v $n
  done
done
cd $R
python3 - "$@" <<'PY'
import csv,glob,collections,sys
for v in sys.argv[1:]:
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/dbpmc/%s.*/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'deblock' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    mb=300*120*68
    print(v, {k: round(sum(x)/len(x)/mb,1) for k,x in sorted(acc.items())})
PY
