#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_q}
mkdir -p $OUT
python tools/estimate_ab.py --cases cfg3:1,cfg3:2,cfg3:4,cfg3:8,cfg3:16,cfg3:32,cfg3:64 > $OUT/est_clean.jsonl 2> $OUT/err.log
python tools/estimate_ab.py --cases cfg3:4,cfg3:16,cfg3:64 --outlier 0.095 > $OUT/est_noisy.jsonl 2>> $OUT/err.log
python tools/estimate_ab.py --cases cfg3:16,cfg3:64 --outlier 0.3 >> $OUT/est_noisy.jsonl 2>> $OUT/err.log
python tools/estimate_ab.py --cases cfg5:16 --hn 2048 >> $OUT/est_clean.jsonl 2>> $OUT/err.log
PVV_STAGE_EIGHTH=0 LD_PRELOAD=$PWD/build/variants/t11.so python tools/estimate_ab.py --cases cfg3:8,cfg3:64 > $OUT/est_quarter.jsonl 2>> $OUT/err.log
PVV_STAGE_EIGHTH=1 LD_PRELOAD=$PWD/build/variants/t11.so python tools/estimate_ab.py --cases cfg3:8,cfg3:64 > $OUT/est_eighth.jsonl 2>> $OUT/err.log
tail -3 $OUT/err.log; for f in est_clean est_noisy est_quarter est_eighth; do echo == $f; cat $OUT/$f.jsonl | cut -c1-220; done
