#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_g}
mkdir -p $OUT
V=build/variants
python tools/variant_ab.py $V/t4.so $V/t3.so@PVV_RUN_R=3 $V/t4.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 24 >> $OUT/ab.txt 2>&1
python tools/variant_ab.py $V/t4.so $V/t3.so@PVV_RUN_R=2 $V/t4.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 32 --rotate 3 --rounds 24 >> $OUT/ab.txt 2>&1
python tools/variant_ab.py $V/t4.so $V/t3.so@PVV_RUN_R=5 $V/t4.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 128 --rotate 3 --rounds 16 >> $OUT/ab.txt 2>&1
python tools/variant_ab.py $V/t4.so $V/t3.so@PVV_RUN_R=9 $V/t4.so@PVV_FILTER_OLD=1 --mode v3 --config cfg5 --batch 16 --rotate 3 --rounds 12 >> $OUT/ab.txt 2>&1
python tools/variant_ab.py $V/t4.so $V/t3.so@PVV_RUN_R=1 $V/t4.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 16 --rotate 3 --rounds 24 >> $OUT/ab.txt 2>&1
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
