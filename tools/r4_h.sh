#!/bin/bash
# round 4, call 8: cooperative elimination (shared miss counters) -- exactness, then A/B vs the run-local version (t3) and round 3's
OUT=$PWD/gpurun_out/${1:-r4_h}
mkdir -p $OUT
PVV_SOAK_CASES=150 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
V=build/variants
ab() { python tools/variant_ab.py $V/t5.so@PVV_RUN_R=1 "$@" >> $OUT/ab.txt 2>&1; }   # (the first entry is a warm-up: it reads ~1.5 % slow whatever it is)
ab $V/t5.so $V/t5.so@PVV_RUN_R=2 $V/t5.so@PVV_RUN_R=3 $V/t5.so@PVV_RUN_R=5 $V/t3.so@PVV_RUN_R=3 $V/t5.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 64 --rotate 3 --rounds 20
ab $V/t5.so $V/t5.so@PVV_RUN_R=2 $V/t5.so@PVV_RUN_R=3 $V/t3.so@PVV_RUN_R=2 $V/t5.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 32 --rotate 3 --rounds 20
ab $V/t5.so $V/t5.so@PVV_RUN_R=3 $V/t5.so@PVV_RUN_R=5 $V/t3.so@PVV_RUN_R=5 $V/t5.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 128 --rotate 3 --rounds 14
ab $V/t5.so $V/t5.so@PVV_RUN_R=4 $V/t5.so@PVV_RUN_R=9 $V/t3.so@PVV_RUN_R=9 $V/t5.so@PVV_FILTER_OLD=1 --mode v3 --config cfg5 --batch 16 --rotate 3 --rounds 10
ab $V/t5.so $V/t5.so@PVV_RUN_R=2 $V/t5.so@PVV_FILTER_OLD=1 --mode v3 --config cfg3 --batch 16 --rotate 3 --rounds 20
grep -a '^{' $OUT/ab.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'], d['B'], d['lib'].ljust(28), d['ms_mean'], d['ms_sem'], d['ratio'], d['win_sum'])"
