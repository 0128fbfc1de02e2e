#!/bin/bash
# why k_compact_hyp does not get faster when its gathers coalesce.  Cache counters of the kernel on the SAME
# batch in the two vertex layouts -- contiguous [B,H,W,K,2] (a thread gathers 9 x 8 B out of its pixel's 72-byte record) and
# planar (storage [B,2K,H,W]: 18 x 4 B, consecutive foreground pixels of a row are consecutive addresses in every plane) --
# in separate rocprofv3 --pmc passes over tools/trace_calls.py.   usage (GPU box): bash tools/compact_counters.sh <outdir>
set -u
OUT=$PWD/gpurun_out/${1:-cc}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
cd /tmp
for row in cfg3_B64 cfg3_B64_planar_vertex; do
  i=0
  for pmc in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_LATENCY_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    rocprofv3 --pmc $pmc -d $OUT/raw_$row/pmc$i -o p --output-format csv -- python $ROOT/tools/trace_calls.py $row 30 > $OUT/raw_${row}_$i.log 2>&1
  done
  rocprofv3 --kernel-trace --stats -d $OUT/raw_$row/trace -o trace --output-format csv -- python $ROOT/tools/trace_calls.py $row 30 > $OUT/raw_${row}_t.log 2>&1
  python $ROOT/tools/summarize_prof.py $OUT/raw_$row --json $OUT/compact_counters_$row.json > $OUT/compact_counters_$row.txt 2>&1
  rm -rf $OUT/raw_$row
done
python - "$OUT" <<'PY'
import json, sys
o = sys.argv[1]
res = {}
for row in ("cfg3_B64", "cfg3_B64_planar_vertex"):
    s = json.load(open("%s/compact_counters_%s.json" % (o, row)))
    res[row] = {"avg_us": s["kernel_stats"]["k_compact_hyp"]["avg_us"],
                **{k: v["main_mean"] for k, v in s["pmc"]["k_compact_hyp"].items()}}
json.dump(res, open(o + "/compact_layout_counters.json", "w"), indent=1)
for k in sorted(set(res["cfg3_B64"]) | set(res["cfg3_B64_planar_vertex"])):
    a, b = res["cfg3_B64"].get(k), res["cfg3_B64_planar_vertex"].get(k)
    print("%-32s contiguous %14.6g   planar %14.6g   ratio %.3f" % (k, a or 0, b or 0, (b / a) if a and b else 0))
PY
