#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_t}
mkdir -p $OUT
PVV_SOAK_CASES=300 python -m pytest tests/test_gpu_staged.py tests/test_gpu_parity.py -m gpu -q > $OUT/tests.log 2>&1
tail -4 $OUT/tests.log
