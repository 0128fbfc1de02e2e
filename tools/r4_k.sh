#!/bin/bash
# round 4: the break-even of staged vs full counting with the run-owning, cooperatively eliminating second launch
OUT=$PWD/gpurun_out/${1:-r4_k}
mkdir -p $OUT
for o in 0.0 0.05 0.095 0.15 0.2 0.3; do
  python tools/staged_ab.py --cases cfg3:16,cfg3:32,cfg3:64,cfg5:16 --outlier $o --rotate 3 >> $OUT/staged_ab_outliers.jsonl 2>> $OUT/err.log
done
python tools/staged_ab.py --cases cfg3:8,cfg3:16,cfg3:32,cfg3:64,cfg4:32,cfg5:16 --rotate 3 --out $OUT/staged_ab.json > $OUT/staged_ab.log 2>> $OUT/err.log
python - <<'PY'
import json
for l in open('gpurun_out/r4_k/staged_ab_outliers.jsonl'):
    d=json.loads(l); print(d['case'], d['outlier'], d['mean_winner_ratio'], d['full']['ms_per_call'], d['staged']['ms_per_call'], d['auto']['ms_per_call'], d['speedup_call'], d['staged_equals_full'])
PY
