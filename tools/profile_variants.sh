#!/bin/bash
# PMC profile of the count-kernel variants (run through gpurun): tools/profile_variants.sh "bf16 fast"
set -u
OUT=$PWD/gpurun_out/prof_var; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline"
cd /tmp
for k in ${1:-bf16}; do
PVV_COUNT_KERNEL=$k rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/$k -o p --output-format csv -- $BENCH > $OUT/$k.log 2>&1
PVV_COUNT_KERNEL=$k rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT -d $OUT/${k}b -o p --output-format csv -- $BENCH >> $OUT/$k.log 2>&1
done
