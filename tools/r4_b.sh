#!/bin/bash
# round 4: GPU suite + the real-caller-layout rows after k_tile_scan_seg2 / the deferred mask
OUT=$PWD/gpurun_out/${1:-r4_b}
mkdir -p $OUT
python -m pytest tests -m gpu -q -s 2>&1 | tail -80 > $OUT/tests.log
python tools/config_bench.py --rows cfg3_B64,cfg3_B64_decode_fused,cfg3_B64_decode_unfused,cfg2_B1,cfg2_B1_decode_fused,cfg3_B8_shard_of_8gpu,default_path_hn128_maxnum100_B64 --out $OUT/configs.json > $OUT/configs.log 2>&1
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -25 $OUT/tests.log; tail -c 300 $OUT/bench.json; tail -3 $OUT/bench.err
