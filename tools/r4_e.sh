#!/bin/bash
# round 4, call 5: run-owning filter launch with progressive elimination -- exactness (staged tests + soak) and A/B vs round 3's
OUT=$PWD/gpurun_out/${1:-r4_e}
mkdir -p $OUT
PVV_SOAK_CASES=120 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -q -x > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
V=build/variants
for cb in "cfg3 64" "cfg3 32" "cfg3 16" "cfg5 16" "cfg4 32" "cfg3 128"; do set -- $cb
  python tools/variant_ab.py $V/t2.so $V/t2.so@PVV_FILTER_OLD=1 --mode v3 --config $1 --batch $2 --rotate 3 --rounds 16 >> $OUT/ab.txt 2>&1
done
grep -a '^{' $OUT/ab.txt | cut -c1-220
