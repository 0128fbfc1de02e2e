#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5p
mkdir -p $OUT
for cfg in "cfg3 64" "cfg3 8" "cfg3 1" "cfg5 16" "cfg3 32"; do set -- $cfg
  timeout 300 python tools/variant_ab.py build/variants/base.so build/variants/base.so build/variants/sg80.so build/variants/sg80b.so build/variants/base.so build/variants/sg80.so build/variants/sg80b.so --mode v3 --config $1 --batch $2 --rotate 3 --rounds 30 > $OUT/ab_$1_$2.log 2>&1; grep '^{' $OUT/ab_$1_$2.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'],d['B'],d['lib'],d['ms_mean'],d['ratio'],d['win_sum'])"
done
