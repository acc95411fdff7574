#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/i8
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 300 > gpurun_out/i8/pytest.log 2>&1; rc=$?; tail -5 gpurun_out/i8/pytest.log
[ $rc -eq 0 ] || exit 1
for k in band row; do
  DRYV_RECON_KERNEL=$k timeout -k 10 300 python bench.py --workload C3_4k_intra_8x8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/i8/c3_$k.json 2> gpurun_out/i8/c3_$k.err || { echo "c3 $k failed"; tail -5 gpurun_out/i8/c3_$k.err; exit 1; }
  python - $k <<'PY'
import json,sys
d=json.load(open('gpurun_out/i8/c3_%s.json'%sys.argv[1]))
print("C3 %s: kernel %s ms %.3f frac %.3f verified %s" % (sys.argv[1], d['roofline']['kernel'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], d['config'].get('shards_verified_bit_exact')))
PY
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/i8/c2.json 2>gpurun_out/i8/c2.err && python -c "
import json; d=json.load(open('gpurun_out/i8/c2.json')); print('C2 kernel ms %.3f'%d['roofline']['kernel_ms_avg'])"
