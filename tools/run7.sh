#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r5j
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/test.log 2>&1
tail -5 $OUT/test.log
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1500 $OUT/bench_default.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err; tail -c 600 $OUT/bench_torchrun1.err
python - <<'PY'
import json
for f in ("bench_default","bench_torchrun1"):
    try:
        d=json.loads([l for l in open("gpurun_out/r5j/%s.json"%f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    print(f, d["metric"], d["value"], d["ms_per_step"], d["scaling"], d.get("value_strong"), d.get("value_weak"))
    print("  roofline", {k:d["roofline"][k] for k in ("bound","achieved","peak","frac","traffic","mfma_busy_frac","busy_frac_counter","share_of_call")})
    print("  dense_eq", d["roofline_dense_equivalent"]["frac"], "sustained", d["extra"].get("sustained"))
    print("  cpu", d["cpu_baseline"] and {k:d["cpu_baseline"][k] for k in ("value","cores","spread","single_thread","thread_probe","same_idxs_gpu_check")})
    print("  predicted", d["extra"]["predicted_8gpu"])
    print("  exchange", d["extra"].get("exchange"), d["extra"].get("exchange_calibration"))
PY
