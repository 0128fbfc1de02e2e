#!/bin/bash
# Second half of the evidence run: the bench lines again, AFTER tools/refresh_profiles.py has written profiles/call_pmc.json from
# the PMC passes of evidence.sh on the same tree -- so that the lines' static counter figures are those of the committed profile.
#   gpurun -- 'bash tools/evidence_bench.sh <tag>'   then   python tools/refresh_profiles.py - gpurun_out/<tag> r05 --bench-only
set -u
TAG=${1:-evb}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> $OUT/bench_driver_command.err
python bench.py --no-cpu-baseline --extras --steps 50 > $OUT/bench_extras.json 2> $OUT/bench_extras.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace --output-format csv -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-two-stream --no-side-legs > $OUT/bench_under_rocprof.log 2>&1)
grep "^{" $OUT/bench_under_rocprof.log | tail -1 > $OUT/bench_under_rocprof.json
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_raw.csv 2>/dev/null
rm -rf $OUT/trace
tail -c 300 $OUT/bench_default.json
