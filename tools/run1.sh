#!/bin/bash
# round 5, GPU run 1: phase census of k_count_filter_runs, the matrix-pipe counters of the count kernels (main + side), quick sanity
set -u
OUT=$PWD/gpurun_out/r5a
mkdir -p $OUT
export TMPDIR=/tmp
PVV_LIBPATH=build/variants/stamps.so timeout 300 python tools/census_filter.py --cases cfg3:64,cfg5:16,cfg3:32 --out $OUT/filter_census.json > $OUT/census.log 2>&1
tail -3 $OUT/census.log
BENCH="python $PWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stream --no-side-legs"
SIDE="python $PWD/tools/prof_side.py"
mkdir -p $OUT/prof/side
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/prof/pmc5 -o pmc5 --output-format csv -- $BENCH > $OUT/bench_pmc5.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof/pmc1 -o pmc1 --output-format csv -- $BENCH > $OUT/bench_pmc1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/prof/pmc2 -o pmc2 --output-format csv -- $BENCH > $OUT/bench_pmc2.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT/prof/side/pmc5 -o pmc5 --output-format csv -- $SIDE > $OUT/side_pmc5.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $OUT/prof/side/pmc1 -o pmc1 --output-format csv -- $SIDE > $OUT/side_pmc1.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT/prof/side/pmc2 -o pmc2 --output-format csv -- $SIDE > $OUT/side_pmc2.log 2>&1)
python tools/summarize_prof.py $OUT/prof --json $OUT/prof_summary.json > $OUT/prof_summary.txt 2>&1
python tools/summarize_prof.py $OUT/prof/side --json $OUT/prof_side_summary.json > $OUT/prof_side_summary.txt 2>&1
rm -rf $OUT/prof
timeout 600 python -m pytest tests/test_gpu_staged.py tests/test_gpu_golden.py -m gpu -x -q > $OUT/test.log 2>&1
tail -3 $OUT/test.log
grep -A12 "pmc k_count_filter_runs" $OUT/prof_summary.txt | head -30
