for g in 1280 1536 1700 1792 1920 2048; do
  DRYV_RECON_GRID=$g DRYV_RECON_LIB=dryv_amd/lib/var/t1w8.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('grid', sys.argv[1], round(d['roofline']['kernel_ms_avg'],4))" $g
done
