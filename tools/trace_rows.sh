#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace of back-to-back calls for each row given, summarised by trace_gaps.py.
# usage: tools/trace_rows.sh <outdir under gpurun_out> row1 row2 ...
set -u
OUT=$PWD/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for row in "$@"; do
  rocprofv3 --kernel-trace -d $OUT/trace_$row -o t --output-format csv -- python $ROOT/tools/trace_calls.py $row 60 > $OUT/trace_$row.log 2>&1
  echo "== $row" | tee -a $OUT/gaps.txt
  python $ROOT/tools/trace_gaps.py $OUT/trace_$row --json $OUT/gaps_$row.json | tee -a $OUT/gaps.txt
  rm -rf $OUT/trace_$row
done
