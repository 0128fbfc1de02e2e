#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_s}
mkdir -p $OUT
python -m pytest tests/test_gpu_decode_layout.py tests/test_gpu_staged.py -m gpu -q > $OUT/tests.log 2>&1
tail -5 $OUT/tests.log
