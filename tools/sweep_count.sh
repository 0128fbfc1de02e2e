for g in ${GRIDS:-24 32 48 64}; do for w in ${PPWS:-128 192 256 384}; do
  echo -n "grid/CU=$g ppw=$w : "
  PVV_GRID_PER_CU=$g PVV_PIX_PER_WAVE=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
