#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_l}
mkdir -p $OUT
for o in 0.0 0.03 0.05 0.095 0.2 0.3 0.4; do
  python tools/staged_ab.py --cases cfg3:24,cfg3:48,cfg3:128 --outlier $o --rotate 2 >> $OUT/staged_ab_outliers.jsonl 2>> $OUT/err.log
done
for o in 0.01 0.02 0.03; do
  python tools/staged_ab.py --cases cfg3:16,cfg3:32 --outlier $o --rotate 3 >> $OUT/staged_ab_outliers.jsonl 2>> $OUT/err.log
done
python tools/staged_ab.py --cases cfg3:64 --outlier 0.25 --rotate 3 >> $OUT/staged_ab_outliers.jsonl 2>> $OUT/err.log
python - <<'PY'
import json
for l in open('gpurun_out/r4_l/staged_ab_outliers.jsonl'):
    d=json.loads(l); print(d['case'], d['outlier'], d['mean_winner_ratio'], d['full']['ms_per_call'], d['staged']['ms_per_call'], d['auto']['ms_per_call'], d['speedup_call'], d['staged_equals_full'])
PY
