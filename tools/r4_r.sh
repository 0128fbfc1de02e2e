#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_r}
mkdir -p $OUT
python -m pytest tests -m gpu -q -s > $OUT/tests.log 2>&1
grep -a "passed\|failed\|ref-glue\|ref-pin" $OUT/tests.log | tail -14
python -c "
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -2
