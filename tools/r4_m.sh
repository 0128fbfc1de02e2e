#!/bin/bash
# round 4: the first stage's share of the chunks again, now that the second launch keeps eliminating (round 3: a quarter won)
OUT=$PWD/gpurun_out/${1:-r4_m}
mkdir -p $OUT
V=build/variants
ab() { python tools/variant_ab.py $V/s8_2.so@PVV_RUN_R=1 $V/s8_2.so $V/s8_1.so $V/s6_1.so $V/s5_1.so $V/s8_3.so "$@" >> $OUT/ab.txt 2>&1; }
ab --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 16
ab --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 16 --outlier 0.095
ab --mode v3 --config cfg3 --batch 32 --rotate 3 --rounds 16
ab --mode v3 --config cfg3 --batch 128 --rotate 2 --rounds 12
ab --mode v3 --config cfg5 --batch 16 --rotate 2 --rounds 8
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
