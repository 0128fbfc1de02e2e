#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_p}
mkdir -p $OUT
V=build/variants
python tools/variant_ab.py $V/e0.so@PVV_MASK_GRID_PER_CU=1 $V/e1.so $V/e0.so $V/e1.so $V/e0.so $V/e1.so@PVV_MASK_DEFER=0 --mode decode --rotate 3 --rounds 20 > $OUT/ab_decode.txt 2>&1
python tools/variant_ab.py $V/e0.so@PVV_RUN_R=1 $V/e1.so --mode v3 --rotate 3 --rounds 20 > $OUT/ab_v3.txt 2>&1
python -m pytest tests/test_gpu_decode_layout.py -m gpu -q 2>&1 | tail -3
export TMPDIR=/tmp; bash tools/trace_rows.sh r4_p cfg3_B64_decode_fused > $OUT/gaps.log 2>&1
cat $OUT/ab_decode.txt $OUT/ab_v3.txt | grep '^{' | cut -c1-200; cat $OUT/gaps.txt
