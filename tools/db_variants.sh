#!/bin/bash
# tuning only: deblock kernel time of build variants (-D flags) on 300 x 1080p
for v in "$@"; do
  so=dryv_amd/lib/libdryv_recon_var.so
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w $v -o $so dryv_amd/csrc/recon_band.hip dryv_amd/csrc/output_pack.hip dryv_amd/csrc/deblock.hip dryv_amd/csrc/recon_api.hip || exit 1
  for r in 1 2; do echo -n "[$v] "; DRYV_RECON_LIB=$so timeout -k 10 200 python tools/deblock_rate.py 300 2>&1 | grep -E "deblock:|Error|assert" | cut -c1-70; done
done
