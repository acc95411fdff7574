#!/bin/bash
# analysis only (GPU box): address-translation and L1 <-> L2 request counters of the libraries in dryv_amd/lib/var/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${WL:-C2_1080p_intra_4x4}
OUT=$R/gpurun_out/mem; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum"
P2="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum"
P4="TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
for so in $R/dryv_amd/lib/var/*.so; do
  n=$(basename $so .so)
  k=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    k=$((k+1))
    DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pmc_${n}_$k -- python3 $R/bench.py --workload $WL --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/pmc_${n}_$k.log 2>&1 || echo "pmc $n $k failed"
  done
  echo "done $n" >> $OUT/progress.txt
done
cd $R
python3 - <<'PY' | tee $OUT/summary.txt
import csv,glob,collections,os
res=collections.OrderedDict()
for d in sorted(glob.glob('gpurun_out/mem/pmc_*/')):
    n=os.path.basename(d[:-1])[4:].rsplit('_',1)[0]
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k,v in acc.items(): res.setdefault(n,{})[k]=sum(v)/len(v)
for n,a in res.items():
    g=lambda k: a.get(k,0.0)
    print("%-8s utcl1 req %.1f M miss %.2f M (%.2f %%) miss-under-miss %.2f M thrash-stall %.1f M multi-miss-stall %.1f M inflight-max-stall %.1f M | pending-stall %.0f M cyc | rd req %.1f M lat %.0f cyc  wr req %.1f M lat %.0f cyc | L2 req %.1f M hit %.1f M miss %.1f M tag-stall %.1f M" % (
        n, g('TCP_UTCL1_REQUEST_sum')/1e6, g('TCP_UTCL1_TRANSLATION_MISS_sum')/1e6, 100*g('TCP_UTCL1_TRANSLATION_MISS_sum')/max(g('TCP_UTCL1_REQUEST_sum'),1),
        g('TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum')/1e6, g('TCP_UTCL1_THRASHING_STALL_sum')/1e6, g('TCP_UTCL1_STALL_MULTI_MISS_sum')/1e6, g('TCP_UTCL1_STALL_INFLIGHT_MAX_sum')/1e6,
        g('TCP_PENDING_STALL_CYCLES_sum')/1e6,
        g('TCP_TCC_READ_REQ_sum')/1e6, g('TCP_TCC_READ_REQ_LATENCY_sum')/max(g('TCP_TCC_READ_REQ_sum'),1),
        g('TCP_TCC_WRITE_REQ_sum')/1e6, g('TCP_TCC_WRITE_REQ_LATENCY_sum')/max(g('TCP_TCC_WRITE_REQ_sum'),1),
        g('TCC_REQ_sum')/1e6, g('TCC_HIT_sum')/1e6, g('TCC_MISS_sum')/1e6, g('TCC_TAG_STALL_sum')/1e6))
PY
