#!/bin/bash
# round 4, call 6: run length of the run-owning filter launch (auto = at most one item per block) vs forced lengths vs round 3's
OUT=$PWD/gpurun_out/${1:-r4_f}
mkdir -p $OUT
V=build/variants
for cb in "cfg3 64" "cfg3 32" "cfg3 128" "cfg3 16" "cfg3 48"; do set -- $cb
  python tools/variant_ab.py $V/t3.so $V/t3.so@PVV_RUN_R=1 $V/t3.so@PVV_RUN_R=2 $V/t3.so@PVV_RUN_R=3 $V/t3.so@PVV_RUN_R=5 $V/t3.so@PVV_RUN_R=9 $V/t3.so@PVV_FILTER_OLD=1 --mode v3 --config $1 --batch $2 --rotate 3 --rounds 16 >> $OUT/ab.txt 2>&1
done
python tools/variant_ab.py $V/t3.so $V/t3.so@PVV_RUN_R=5 $V/t3.so@PVV_RUN_R=9 $V/t3.so@PVV_RUN_R=22 $V/t3.so@PVV_RUN_R=44 $V/t3.so@PVV_FILTER_OLD=1 --mode v3 --config cfg5 --batch 16 --rotate 3 --rounds 12 >> $OUT/ab.txt 2>&1
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ratio'], d['win_sum'])"
