#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5o
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/test.log 2>&1
tail -4 $OUT/test.log
python bench.py --no-cpu-baseline --no-sustained > $OUT/bench.json 2> $OUT/bench.err; tail -c 400 $OUT/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r5o/bench.json") if l.startswith("{")][-1])
e=d["extra"]
print(d["value"], d["ms_per_step"])
for k in e:
    if "pnp" in k or "estimate" in k: print(k, e[k] if not isinstance(e[k],dict) else json.dumps(e[k])[:300])
PY
