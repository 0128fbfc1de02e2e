export LD_PRELOAD=$PWD/build/variants/tuning.so
fmt='import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r["case"], "full", r["full"], "staged", r["staged"], "staged/full %.3f" % (r["staged"] / r["full"]))'
for e in 0 1 0 1; do echo "EIGHTH=$e"; PVV_STAGE_EIGHTH=$e python tools/estimate_ab.py --cases cfg3:6,cfg3:8,cfg3:16,cfg3:32,cfg3:64 2>&1 | grep "^{" | python -c "$fmt"; done
for e in 0 1; do echo "cfg5 EIGHTH=$e"; PVV_STAGE_EIGHTH=$e python tools/estimate_ab.py --cases cfg5:4,cfg5:16 --hn 2048 2>&1 | grep "^{" | python -c "$fmt"; done
