#!/bin/bash
# tuning only: HBM write / fetch counters and kernel time of band-kernel build variants (-D flags)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/tv
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  so=$R/dryv_amd/lib/libdryv_recon_var.so
  (cd $R && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w $v -o $so dryv_amd/csrc/recon_band.hip dryv_amd/csrc/output_pack.hip dryv_amd/csrc/deblock.hip dryv_amd/csrc/recon_api.hip) || exit 1
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    DRYV_RECON_LIB=$so timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/tv/${name}_$c -- python3 $R/bench.py --steps 3 --warmup 1 --preroll-ms 0 --no-cpu-baseline --no-verify > $R/gpurun_out/tv/${name}_$c.log 2>&1 || echo "$c failed"
  done
  cd $R
  DRYV_RECON_LIB=$so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/tv/$name.json 2>/dev/null
  python - "$name" <<'PY'
import csv,glob,json,sys
n=sys.argv[1]
out={}
for c in ("FETCH_SIZE","WRITE_SIZE"):
    vals=[]
    for f in glob.glob('gpurun_out/tv/%s_%s/**/*counter_collection.csv'%(n,c), recursive=True):
        vals+=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'band_kernel' in r['Kernel_Name'] and r['Counter_Name']==c]
    out[c]=sum(vals)/max(len(vals),1)*1024/1e9
d=json.load(open('gpurun_out/tv/%s.json'%n))
print("%-36s kernel_ms %.3f  FETCH raw %.2f GB  WRITE %.2f GB" % (n, d['roofline']['kernel_ms_avg'], out['FETCH_SIZE'], out['WRITE_SIZE']), flush=True)
PY
done
