#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_staged.py -m gpu -x -q 2>&1 | tail -2
PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64,cfg5:16 --out $OUT/filter_census.json > $OUT/census.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5i/filter_census.json"))
for c in d["cases"]:
    print(c["case"], {k:c[k] for k in ("items","chunks","span_us","life_us_median","life_us_max","entry_us_max","survivors_per_chunk_mean")}, c["working_blocks_alive_at_fraction_of_span"], c["by_items_per_block"])
    print({k:v["share_of_working_blocks_cycles"] for k,v in c["phases"].items()})
PY
for cfg in "cfg3 64" "cfg3 32" "cfg3 24" "cfg5 16" "cfg3 128" "cfg3 96" "cfg5 4"; do set -- $cfg
  timeout 300 python tools/variant_ab.py build/variants/base.so build/variants/base.so build/variants/share.so --mode v3 --config $1 --batch $2 --rotate 3 --rounds 30 > $OUT/ab_$1_$2.log 2>&1; grep '^{' $OUT/ab_$1_$2.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],d['B'],d['lib'],d['ms_mean'],d['ratio'],d['win_sum'])"
done
