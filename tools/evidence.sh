#!/bin/bash
# One gpurun call that produces everything tools/refresh_profiles.py copies into profiles/ for a round:
#   gpurun --timeout 1500 -- 'bash tools/evidence.sh r2f'     then     python tools/refresh_profiles.py gpurun_out/prof_r2f gpurun_out/r2f r02
# Build BEFORE the call, in the container (hipcc cross-compiles; the artefacts travel with the snapshot):
#   tools/build_variant.sh stamps -DPVV_TUNING -DPVV_STAMPS          # census_filter.py, census_count.py
#   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form tools/microbench/count_pipe3.hip -o build/mb/cp3
#   (same flags) tools/microbench/count_pipe2.hip -o build/mb/cp2    # round 5's version, kept for continuity
set -u
TAG=${1:-ev}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1
# the raw rocprofv3 output is tens of MB (gpurun merges at most 64 MiB back): summarise here, keep the stats CSV only
python tools/summarize_prof.py $PWD/gpurun_out/prof_$TAG --json $OUT/prof_summary.json > $OUT/prof_summary.txt 2>&1
python tools/summarize_prof.py $PWD/gpurun_out/prof_$TAG/side --json $OUT/prof_side_summary.json > $OUT/prof_side_summary.txt 2>&1
cp $(find $PWD/gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_raw.csv 2>/dev/null
# the JSON line the PROFILED process itself printed: its HIP-event kernel time and rocprofv3's average are one process
grep "^{" $PWD/gpurun_out/prof_$TAG/bench_trace.log | tail -1 > $OUT/bench_under_rocprof.json
du -sh $PWD/gpurun_out/prof_$TAG >> $OUT/profile.log
rm -rf $PWD/gpurun_out/prof_$TAG
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python bench.py --no-cpu-baseline --extras --steps 50 > $OUT/bench_extras.json 2> $OUT/bench_extras.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu-baseline > $OUT/bench_torchrun1.json 2> $OUT/bench_torchrun1.err
python tools/config_bench.py --out $OUT/configs.json > $OUT/configs.log 2>&1
bash tools/trace_rows.sh $TAG cfg2_B1 cfg3_B8_shard_of_8gpu cfg3_B64 default_path_hn128_maxnum100_B64 default_path_hn128_maxnum100_B1 cfg3_B64_decode_fused cfg2_B1_decode_fused cfg3_B64_planar_vertex > $OUT/gaps.log 2>&1
# staged vs full count pass: the BASELINE configs, and config 3 / 5 over outlier fractions (AUTO's break-even, profiles/DESIGN_rounds_1-4.md 4.7)
python tools/staged_ab.py --cases cfg3:8,cfg3:16,cfg3:32,cfg3:64,cfg4:32,cfg5:16 --rotate 3 --out $OUT/staged_ab.json > $OUT/staged_ab.log 2>&1
rm -f $OUT/staged_ab_outliers.jsonl
for o in 0.03 0.05 0.095 0.2 0.3; do python tools/staged_ab.py --cases cfg3:16,cfg3:32,cfg3:64,cfg3:128,cfg5:16 --outlier $o --rotate 2 >> $OUT/staged_ab_outliers.jsonl 2>> $OUT/staged_ab.log; done
python -c "import json,sys; json.dump([json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')], open(sys.argv[2],'w'), indent=1)" $OUT/staged_ab_outliers.jsonl $OUT/staged_ab_outliers.json
# round 5: AUTO's regret on structured errors; the phase census of the filter kernel (needs build/variants/stamps.so: tools/build_variant.sh
# stamps -DPVV_TUNING -DPVV_STAMPS, built before the call); the per-tile loop microbenchmark
python tools/auto_regret.py --out $OUT/auto_regret.json > $OUT/auto_regret.log 2>&1
if [ -f build/variants/stamps.so ]; then PVV_LIBPATH=build/variants/stamps.so python tools/census_filter.py --cases cfg3:64,cfg5:16 --out $OUT/filter_census.json > $OUT/census.log 2>&1; fi
if [ -x build/mb/cp2 ]; then build/mb/cp2 > $OUT/count_pipe2.txt 2>&1; fi
# round 6: the effective shader clock INSIDE the count kernels + the phase census of k_count_bf16 (same instrumented build), and the
# per-tile loop microbenchmark with both device counters (hipcc ... tools/microbench/count_pipe3.hip -o build/mb/cp3, built before the call)
if [ -f build/variants/stamps.so ]; then PVV_LIBPATH=build/variants/stamps.so python tools/census_count.py --out $OUT/count_census.json > $OUT/count_census.log 2>&1; fi
if [ -x build/mb/cp3 ]; then build/mb/cp3 > $OUT/count_pipe3.txt 2>&1; fi
tail -3 $OUT/profile.log; tail -c 600 $OUT/bench_default.json; grep -c "" $OUT/configs.log
