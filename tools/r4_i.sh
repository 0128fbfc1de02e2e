#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_i}
mkdir -p $OUT
PVV_SOAK_CASES=100 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
V=build/variants
ab() { python tools/variant_ab.py $V/t5.so@PVV_RUN_R=1 "$@" >> $OUT/ab.txt 2>&1; }
ab $V/t6.so $V/t5.so $V/t6.so@PVV_RUN_R=2 $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 20
ab $V/t6.so $V/t5.so $V/t6.so@PVV_RUN_R=3 $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 32 --rotate 3 --rounds 20
ab $V/t6.so $V/t5.so $V/t6.so@PVV_RUN_R=2 $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 16 --rotate 3 --rounds 20
ab $V/t6.so $V/t5.so $V/t6.so@PVV_RUN_R=2 $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 24 --rotate 3 --rounds 20
ab $V/t6.so $V/t5.so $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 128 --rotate 3 --rounds 12
ab $V/t6.so $V/t5.so $V/t6.so@PVV_FILTER_OLD=1 --mode v3 --config cfg5 --batch 16 --rotate 3 --rounds 10
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
