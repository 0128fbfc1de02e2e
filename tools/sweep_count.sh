# sweep of the persistent-grid knobs of the count kernel (run through gpurun)
for g in ${GRIDS:-4 8 16 24 48}; do for it in ${ITEMS:-6}; do
  echo -n "grid/CU=$g items/CU=$it : "
  PVV_GRID_PER_CU=$g PVV_ITEMS_PER_CU=$it python bench.py --batch ${BATCH:-64} --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'])"
done; done
