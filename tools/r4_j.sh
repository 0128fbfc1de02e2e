#!/bin/bash
OUT=$PWD/gpurun_out/${1:-r4_j}
mkdir -p $OUT
python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1
tail -6 $OUT/tests.log
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
tail -3 $OUT/bench.err
python - <<'PY'
import json,sys
b=json.loads([l for l in open(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/r4_j/bench.json') if l.startswith('{')][-1])
print(b['value'], b['ms_per_step'], b['value_at_rho_0.90'], b['roofline']['frac'])
e=b['extra']
for k in ('noisy_field','v3_plus_estimate_images_per_s','un_pnp_fused_one_pass_images_per_s','decode_fused_images_per_s','decode_fused_vs_headline','decode_fused_mask_equals_torch_argmax','decode_unfused_argmax_plus_v3_images_per_s','decode_fused_scan','kernels_inside_calls_ms'):
    print(k, json.dumps(e.get(k))[:600])
PY
