#!/bin/bash
# round 4, call 4: one-process A/B of the deferred mask writer (grid, store policy, inline) on whole fused decode calls
OUT=$PWD/gpurun_out/${1:-r4_d}
mkdir -p $OUT
V=build/variants
python tools/variant_ab.py $V/t.so $V/t.so@PVV_MASK_GRID_PER_CU=1 $V/t.so@PVV_MASK_GRID_PER_CU=4 $V/t.so@PVV_MASK_GRID_PER_CU=8 $V/tp.so $V/tp.so@PVV_MASK_GRID_PER_CU=1 $V/t.so@PVV_MASK_DEFER=0 --mode decode --rotate 3 --rounds 20 > $OUT/ab_decode.txt 2>&1
python tools/variant_ab.py $V/t.so --mode v3 --rotate 3 --rounds 20 > $OUT/ab_v3.txt 2>&1
python tools/variant_ab.py $V/t.so $V/t.so@PVV_MASK_GRID_PER_CU=1 $V/tp.so $V/t.so@PVV_MASK_DEFER=0 --mode decode --batch 8 --rotate 3 --rounds 20 > $OUT/ab_decode_B8.txt 2>&1
cat $OUT/ab_decode.txt $OUT/ab_v3.txt $OUT/ab_decode_B8.txt | cut -c1-250
