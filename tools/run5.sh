#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5f
mkdir -p $OUT
T=build/variants/t.so
for cfg in "cfg3 64" "cfg3 32" "cfg3 128" "cfg5 16" "cfg3 48"; do set -- $cfg
  timeout 300 python tools/variant_ab.py $T $T@PVV_RUN_R=3 $T@PVV_RUN_R=4 $T@PVV_RUN_R=5 $T@PVV_RUN_R=2 $T@PVV_RUN_R=9 --mode v3 --config $1 --batch $2 --rotate 3 --rounds 24 > $OUT/ab_$1_$2.log 2>&1; grep '^{' $OUT/ab_$1_$2.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],d['B'],d['lib'],d['ms_mean'],d['ratio'])"
done
for r in 0 5; do PVV_RUN_R=$r PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64 --out $OUT/census_R$r.json > /dev/null 2>&1; python -c "
import json
c=json.load(open('$OUT/census_R$r.json'))['cases'][0]
print('R=$r', {k:c[k] for k in ('items','chunks','span_us','life_us_median','life_us_max','survivors_per_chunk_mean')}, c['working_blocks_alive_at_fraction_of_span'])"; done
