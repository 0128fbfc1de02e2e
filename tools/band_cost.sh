for b in 1 8 64; do
for s in 1 0.0001; do
PVV_DEBUG_BAND_SCALE=$s PVV_LIBPATH=build/variants/tune.so python tools/variant_time.py --batch $b --tag band$s | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['B'], d['tag'], d['kernel_ms_avg'], d['win_sum'])"
done; done
