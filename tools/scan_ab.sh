export TMPDIR=/tmp
ROOT=$PWD
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/ps -o t --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --steps 60 > /tmp/ps.log 2>&1
grep "^{" /tmp/ps.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('under rocprof', d['value'], d['ms_per_step'])"
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ps/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('k_') or 'k_' in r['Name'][:40]: print(r['Name'][:40], r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
cd $ROOT
python bench.py --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['ms_per_step'], d['step_ms'], d['extra'].get('two_stream_images_per_s'))"
python tools/two_stream.py 2>/dev/null | tail -1
