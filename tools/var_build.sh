#!/bin/bash
# analysis only: builds library variants into dryv_amd/lib/var/: tools/var_build.sh name1 "flags1" name2 "flags2" ...
set -e
cd "$(dirname "$0")/.."
mkdir -p dryv_amd/lib/var
SRC="dryv_amd/csrc/recon_band.hip dryv_amd/csrc/output_pack.hip dryv_amd/csrc/deblock.hip dryv_amd/csrc/recon_api.hip"
while [ $# -gt 0 ]; do
  n=$1; f=$2; shift 2
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $f -o dryv_amd/lib/var/$n.so $SRC 2>/tmp/var_$n.err || { echo "build $n failed"; tail -5 /tmp/var_$n.err; } ) &
  if (( $(jobs -r | wc -l) >= 7 )); then wait -n; fi
done
wait
ls dryv_amd/lib/var
