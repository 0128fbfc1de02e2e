#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dist.py -m gpu -x -q > $OUT/test.log 2>&1
tail -15 $OUT/test.log
for impl in rccl torch; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline --no-side-legs --exchange-impl $impl > $OUT/bench_torchrun1_$impl.json 2> $OUT/bench_torchrun1_$impl.err; tail -c 300 $OUT/bench_torchrun1_$impl.err
done
python bench.py --no-cpu-baseline --no-side-legs > $OUT/bench_nogroup.json 2>/dev/null
python - <<'PY'
import json
for f in ("bench_nogroup","bench_torchrun1_rccl","bench_torchrun1_torch"):
    try:
        d=json.loads([l for l in open("gpurun_out/r5k/%s.json"%f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    print(f, d["value"], d["ms_per_step"], d["step_ms"], d["extra"].get("exchange_impl"), d["extra"].get("exchange_calibration"))
PY
