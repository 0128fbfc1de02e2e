#!/bin/bash
# round 4, first GPU call: full GPU suite (incl. the reference's own Python on the boundary), the new bench line, the
# true-cycle VALU microbenchmark, and the real-caller-layout rows before any kernel change
OUT=$PWD/gpurun_out/${1:-r4_a}
mkdir -p $OUT
python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > $OUT/tests.log
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
build/mb/valu3 > $OUT/valu3.txt 2>&1
python tools/config_bench.py --rows cfg3_B64,cfg3_B64_planar_vertex,cfg3_B64_decode_fused,cfg3_B64_decode_unfused,cfg2_B1,cfg2_B1_decode_fused,cfg2_B1_decode_unfused --out $OUT/configs.json > $OUT/configs.log 2>&1
tail -5 $OUT/tests.log; tail -c 400 $OUT/bench.json; tail -3 $OUT/bench.err; tail -3 $OUT/configs.log | cut -c1-300
