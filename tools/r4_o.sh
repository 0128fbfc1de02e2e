#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_o}
mkdir -p $OUT
V=build/variants
ab() { python tools/variant_ab.py $V/t8_ref.so@PVV_RUN_R=1 "$@" >> $OUT/ab.txt 2>&1; }
ab $V/t9.so $V/t8_ref.so $V/t9.so $V/t8_ref.so --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 16
ab $V/t9.so $V/t8_ref.so --mode v3 --config cfg3 --batch 32 --rotate 3 --rounds 16
ab $V/t9.so $V/t8_ref.so --mode v3 --config cfg3 --batch 128 --rotate 2 --rounds 12
ab $V/t9.so $V/t8_ref.so --mode v3 --config cfg5 --batch 16 --rotate 2 --rounds 8
ab $V/t9.so $V/t8_ref.so --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 16 --outlier 0.095
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
