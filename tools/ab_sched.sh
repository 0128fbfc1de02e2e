# Whole-call A/B (cold batches, one process) of scheduling variants of the count kernel: build them with
# tools/build_variant.sh, list them in run(), then   gpurun -- "bash tools/ab_sched.sh"
run() { python tools/variant_ab.py build/variants/sched.so build/variants/devE.so --mode v3 --rotate 2 --rounds $1 --batch $2 --config $3 | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]; print('$3 B',r[0]['B'],[(x['lib'][:-3],x['ms_mean'],x['ratio']) for x in r], len(set(x['out_sum'] for x in r))==1)"; }
for b in 8 16 64; do run 20 $b cfg3; done
for b in 2 4 16; do run 6 $b cfg5; done
for b in 4 16 32; do run 15 $b cfg4; done
