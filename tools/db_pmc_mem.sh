#!/bin/bash
# analysis only (GPU box): memory-side counters of the deblocking kernel for the libraries named (dryv_amd/lib/var/<name>.so)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  k=0
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
             "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
             "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_WRITEBACK_sum TCC_NORMAL_WRITEBACK_sum" \
             "TCP_TA_TCP_STATE_READ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
             "TA_BUSY_sum TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
    k=$((k+1))
    DRYV_RECON_LIB=$R/dryv_amd/lib/var/$v.so timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/dbmem/$v.$k -- python3 $R/tools/deblock_rate.py 300 > $R/gpurun_out/dbmem_$v.$k.log 2>&1 || echo fail $v $k
  done
done
cd $R
python3 - "$@" <<'PY'
import csv,glob,collections,sys
for v in sys.argv[1:]:
    acc=collections.defaultdict(list)
    for f in glob.glob('gpurun_out/dbmem/%s.*/**/*counter_collection.csv'%v, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'deblock' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(v)
    for k,x in sorted(acc.items()): print("   %-40s %.3f M" % (k, sum(x)/len(x)/1e6))
PY
