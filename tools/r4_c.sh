#!/bin/bash
# round 4, call 3: full suite (whole log), staged test repeated (an 80 ms outlier was seen once), decode rows with the mask
# writer forked behind the compaction, the one-block streaming microbenchmark, counter names of this rocprofv3
OUT=$PWD/gpurun_out/${1:-r4_c}
mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/tests.log 2>&1
for i in 1 2 3; do python -m pytest tests/test_gpu_staged.py -m gpu -q -k "stage_marks" 2>&1 | tail -3 >> $OUT/staged_repeat.log; done
python tools/config_bench.py --rows cfg3_B64,cfg3_B64_decode_fused,cfg3_B64_decode_unfused,cfg2_B1_decode_fused,cfg3_B8_shard_of_8gpu --out $OUT/configs.json > $OUT/configs.log 2>&1
build/mb/obs > $OUT/one_block_stream.txt 2>&1
export TMPDIR=/tmp; (cd /tmp && rocprofv3 -L > $OUT/counters.txt 2>&1)
grep -a "ref-glue\|passed\|failed" $OUT/tests.log | tail -20; cat $OUT/staged_repeat.log; cat $OUT/one_block_stream.txt; wc -l $OUT/counters.txt
