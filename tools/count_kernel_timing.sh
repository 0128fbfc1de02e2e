export TMPDIR=/tmp; ROOT=$PWD; python tools/count_kernel_timing.py
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/trk -o t --output-format csv -- python $ROOT/tools/count_kernel_timing.py > /tmp/kt.log 2>&1; tail -3 /tmp/kt.log
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/trk/**/*kernel_trace.csv',recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
inpipe=[];alone=[]
for i,(s,e,n) in enumerate(rows):
    if 'k_count_bf16' in n:
        prev=rows[i-1][2] if i else ''
        (alone if 'k_count_bf16' in prev else inpipe).append((e-s)/1e3)
print('rocprof: count kernel after k_compact_hyp (in pipeline) n=%d mean %.2f us; after another count kernel n=%d mean %.2f us'%(len(inpipe),sum(inpipe)/len(inpipe),len(alone),sum(alone)/max(1,len(alone))))
PY
