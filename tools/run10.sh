#!/bin/bash
mkdir -p gpurun_out/r5n
{
python tools/estimate_ab.py --cases cfg3:64,cfg3:8 --outlier 0.3 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:64,cfg3:16,cfg3:8 --gen '{"wrong_region": 0.3, "kp_outlier": [0.01, 0.4]}' 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:64 --gen '{"wrong_region": 0.4}' 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:16 --gen '{"H": 256, "W": 256, "fg": 0.33, "wrong_region": 0.3, "kp_outlier": [0.01, 0.4]}' 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:12,cfg3:24,cfg3:48,cfg3:4 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:64,cfg3:16 --hn 1024 2>&1 | grep "^{"
python tools/estimate_ab.py --cases cfg3:64,cfg3:16 --hn 2048 2>&1 | grep "^{"
} | tee gpurun_out/r5n/estimate_ab.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['case'],d['hn'],d.get('gen'),d['outlier'],d['full'],d['staged'],d['auto'],d['speedup'],d['staged_equals_full'],d['auto_equals_full'])"
timeout 600 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
