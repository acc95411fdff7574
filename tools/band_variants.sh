#!/bin/bash
# tuning only: kernel time of band-kernel build variants (-D flags) on the C2 bench workload
set -o pipefail
mkdir -p gpurun_out/variants
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9=\n' '_')
  so=dryv_amd/lib/libdryv_recon_var.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $v -o $so dryv_amd/csrc/recon_band.hip dryv_amd/csrc/output_pack.hip dryv_amd/csrc/deblock.hip dryv_amd/csrc/recon_api.hip || exit 1
  DRYV_RECON_LIB=$so timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > gpurun_out/variants/$name.json 2>gpurun_out/variants/$name.err
  python - "$name" <<'PY'
import json,sys
d=json.load(open('gpurun_out/variants/%s.json'%sys.argv[1]))
print("%-40s kernel_ms %.3f frac %.3f" % (sys.argv[1], d['roofline']['kernel_ms_avg'], d['roofline']['frac']), flush=True)
PY
done
