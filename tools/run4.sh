#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5d
mkdir -p $OUT
PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64,cfg5:16 --out $OUT/filter_census.json > $OUT/census.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5d/filter_census.json"))
for c in d["cases"]:
    print(c["case"], {k:c[k] for k in ("items","chunks","span_us","life_us_median","life_us_max","entry_us_max")}, c["working_blocks_alive_at_fraction_of_span"], c["by_items_per_block"])
PY
for cfg in "cfg3 64" "cfg3 32" "cfg3 24" "cfg5 16" "cfg3 128" "cfg3 96"; do set -- $cfg
  timeout 300 python tools/variant_ab.py build/variants/new.so build/variants/new.so build/variants/dyn.so --mode v3 --config $1 --batch $2 --rotate 3 --rounds 30 > $OUT/ab_$1_$2.log 2>&1; tail -2 $OUT/ab_$1_$2.log | cut -c1-200
done
LD_PRELOAD=$PWD/build/variants/dyn.so timeout 900 python -m pytest tests/test_gpu_staged.py -m gpu -x -q > $OUT/test.log 2>&1
tail -3 $OUT/test.log
