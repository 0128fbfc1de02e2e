#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5b
mkdir -p $OUT
build/mb/cp2 > $OUT/count_pipe2.txt 2>&1
cat $OUT/count_pipe2.txt
PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64 --out $OUT/filter_census.json > $OUT/census.log 2>&1
PVV_RUN_R=5 PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64 --out $OUT/filter_census_R5.json > $OUT/census_R5.log 2>&1
PVV_RUN_R=2 PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64 --out $OUT/filter_census_R2.json > $OUT/census_R2.log 2>&1
python - <<'PY'
import json
for f in ("filter_census","filter_census_R5","filter_census_R2"):
    d=json.load(open("gpurun_out/r5b/%s.json"%f))["cases"][0]
    print(f, {k:d[k] for k in ("items","chunks","span_us","life_us_median","life_us_max","entry_us_median","entry_us_p90","entry_us_max","survivors_per_chunk_mean")}, d["working_blocks_alive_at_fraction_of_span"], d["by_items_per_block"])
PY
