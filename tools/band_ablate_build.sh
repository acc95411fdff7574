#!/bin/bash
# analysis only: builds the full band kernel and one library per ablated phase (-DDRYV_BAND_EXP_SKIP=1<<k) into
# dryv_amd/lib/var/ (run in the build container; the libraries travel to the GPU box with the snapshot)
set -e
cd "$(dirname "$0")/.."
mkdir -p dryv_amd/lib/var
SRC="dryv_amd/csrc/recon_band.hip dryv_amd/csrc/output_pack.hip dryv_amd/csrc/deblock.hip dryv_amd/csrc/recon_api.hip"
build() { hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $2 -o dryv_amd/lib/var/$1.so $SRC 2>/dev/null; }
build full "" &
for k in ${@:-0 1 2 3 4 5 6 7 8 9}; do
  build skip$k "-DDRYV_BAND_EXP_SKIP=$((1 << k))" &
  if (( $(jobs -r | wc -l) >= 6 )); then wait -n; fi
done
wait
ls -la dryv_amd/lib/var
