export TMPDIR=/tmp
ROOT=$PWD
mkdir -p $ROOT/gpurun_out/r2e
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o t --output-format csv -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/r2e/bench_trace.log 2>&1
python $ROOT/tools/trace_gaps.py /tmp/tr --skip 8 | head -12
