#!/bin/bash
# Per-kernel durations of whole calls for several experimental builds (build/variants/<name>.so) on one row.
# usage: tools/trace_variants.sh <row> name1 name2 ...
export TMPDIR=/tmp
ROOT=$PWD
row=$1; shift
cd /tmp
for v in "$@"; do
  echo "== $v $row"
  PVV_LIBPATH=$ROOT/build/variants/$v.so rocprofv3 --kernel-trace -d /tmp/trv_$v -o t --output-format csv -- python $ROOT/tools/trace_calls_capi.py $row 40 > /tmp/log_$v 2>&1
  python $ROOT/tools/trace_gaps.py /tmp/trv_$v | sed -n 3,9p
  rm -rf /tmp/trv_$v
done
