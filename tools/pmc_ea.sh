#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ea; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for so in $R/dryv_amd/lib/var/*.so; do
  n=$(basename $so .so)
  DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $OUT/pmc_$n -- python3 $R/bench.py --workload ${WL:-C2_1080p_intra_4x4} --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $OUT/pmc_$n.log 2>&1 || echo "pmc $n failed"
done
cd $R
python3 - <<'PY'
import csv,glob,collections,os
for d in sorted(glob.glob('gpurun_out/ea/pmc_*/')):
    n=os.path.basename(d[:-1])[4:]
    acc=collections.defaultdict(list)
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'band_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    a={k: sum(v)/len(v) for k,v in acc.items()}
    rd=a.get('TCC_EA0_RDREQ_sum',0); rd32=a.get('TCC_EA0_RDREQ_32B_sum',0); wr=a.get('TCC_EA0_WRREQ_sum',0); wr64=a.get('TCC_EA0_WRREQ_64B_sum',0)
    print("%-6s rdreq %.2f M (32B: %.2f M) -> %.3f GB if 64B else | wrreq %.2f M (64B: %.2f M) -> %.3f GB" % (n, rd/1e6, rd32/1e6, (rd32*32+(rd-rd32)*64)/1e9, wr/1e6, wr64/1e6, (wr64*64+(wr-wr64)*32)/1e9))
PY
