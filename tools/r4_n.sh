#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_n}
mkdir -p $OUT
python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
# the same staged tests with the EIGHTH first stage forced at every size (a tuning build preloaded over the product library)
PVV_STAGE_EIGHTH=1 PVV_SOAK_CASES=100 LD_PRELOAD=$PWD/build/variants/t8.so python -m pytest tests/test_gpu_staged.py -m gpu -q -x > $OUT/tests_eighth.log 2>&1
tail -3 $OUT/tests_eighth.log
PVV_STAGE_EIGHTH=1 PVV_FILTER_OLD=1 LD_PRELOAD=$PWD/build/variants/t8.so python -m pytest tests/test_gpu_staged.py -m gpu -q -x -k "soak or full_size or branches" > $OUT/tests_eighth_old.log 2>&1
tail -3 $OUT/tests_eighth_old.log
V=build/variants
ab() { python tools/variant_ab.py $V/t8.so@PVV_RUN_R=1 "$@" >> $OUT/ab.txt 2>&1; }
ab $V/t8.so $V/t8.so@PVV_STAGE_EIGHTH=0 $V/t8.so@PVV_STAGE_EIGHTH=1 --mode v3 --config cfg3 --batch 96 --rotate 2 --rounds 12
ab $V/t8.so $V/t8.so@PVV_STAGE_EIGHTH=0 $V/t8.so@PVV_STAGE_EIGHTH=1 --mode v3 --config cfg3 --batch 128 --rotate 2 --rounds 12
ab $V/t8.so $V/t8.so@PVV_STAGE_EIGHTH=0 $V/t8.so@PVV_STAGE_EIGHTH=1 --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 12
ab $V/t8.so $V/t8.so@PVV_STAGE_EIGHTH=0 $V/t8.so@PVV_STAGE_EIGHTH=1 --mode v3 --config cfg5 --batch 16 --rotate 2 --rounds 8
ab $V/t8.so $V/t8.so@PVV_STAGE_EIGHTH=0 $V/t8.so@PVV_STAGE_EIGHTH=1 --mode v3 --config cfg5 --batch 8 --rotate 2 --rounds 8
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
