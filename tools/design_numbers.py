import json
p='DESIGN.md'
s=open(p).read()
a=s.index('## 5. Measurement (bench.py)')
b=s.index('## 7. SURVEY')
c=json.load(open('profiles/r02_configs.json'))['rows']
def row(name,label,note=''):
    v=c[name]
    return "| %s | %.1f k | %.4f | %.4f | %.4f | %.2f | %.2f%s |" % (label, v['images_per_s_wall']/1e3, v['event_ms_per_call_median'], v['graph_replay_ms_per_call'], v['count_kernel_ms'], v['tevals_per_s'], v['roofline_frac_hbm_8TBs'], note)
table='\n'.join([
 "  | Config | images/s (wall) | ms/call (events, median) | ms/call (graph replay) | count kernel ms | T eval/s | frac |",
 "  |---|---|---|---|---|---|---|",
 "  "+row('cfg2_B1','2: 480×640, K=9, 512 hyp, B=1',' (round 1: 0.0403 ms, 0.0141 ms, 0.20)'),
 "  "+row('cfg2_dense_tn30000_B1','2, dense stress: foreground ≈ 30 000 pixels, B=1',' (SURVEY §8(d))'),
 "  "+row('cfg3_B2','3: same, B=2'),
 "  "+row('cfg3_B4','3: same, B=4'),
 "  "+row('cfg3_B8_shard_of_8gpu','3: same, B=8 (= 64 images over 8 GPUs)',' (round 1: 0.0702 ms, 0.0384 ms)'),
 "  "+row('cfg3_B16','3: same, B=16'),
 "  "+row('cfg3_B32','3: same, B=32'),
 "  "+row('cfg3_B64','3: same, **B=64** (one batch replayed: warm caches)',' (round 1: 0.2699 ms, 0.1971 ms)'),
 "  "+row('cfg4_B32','4: sparse/occluded (tn≈1.5 k), 1024 hyp, B=32',' (few foreground pixels: the dense-bytes figure overstates the work)'),
 "  "+row('cfg4_B4_shard_of_8gpu','4: same, B=4 (shard of 8 GPUs)'),
 "  "+row('cfg5_B16','5: 540×720, K=17, 2048 hyp, tn capped at 30 000, B=16',' (1.04 G evaluations per image: out of reach of the HBM figure for any exact method, SURVEY §8(d))'),
 "  "+row('cfg5_B2_shard_of_8gpu','5: same, B=2 (shard of 8 GPUs)'),
 "  "+row('default_path_hn128_maxnum100_B64','reference default call (resnet18.py:75: 128 hyp, max_num=100), B=64',' (round 1: 0.0859 ms)'),
 "  "+row('default_path_hn128_maxnum100_B1','same, B=1',' (round 1: 0.0338 ms)'),
])
bd=json.load(open('profiles/r02_bench_default.json'))
ex=json.load(open('profiles/r02_bench_extras.json'))['extra']
import csv
ks={r['Name']: r for r in csv.DictReader(open('profiles/r02_kernel_stats.csv'))}
us=lambda n: "%.1f" % (float(ks[n]['AverageNs'])/1e3)
scan_name=[n for n in ks if n.startswith('k_tile_scan')][0]
gp=json.load(open('profiles/r02_gaps_cfg3_B64.json'))['kernels']
warm_parts=[v['dur_us_median'] for k,v in sorted(gp.items())]
under=json.load(open('profiles/r02_bench_under_rocprof.json'))
tr=json.load(open('profiles/r02_bench_torchrun_1rank.json'))
sec=open('tools/design_sections_5_6.md.tpl').read()
vals=dict(v="%.1f"%(bd['value']/1e3), ms="%.4f"%bd['ms_per_step'], med="%.4f"%bd['step_ms']['median'], p10="%.4f"%bd['step_ms']['p10'], p90="%.4f"%bd['step_ms']['p90'],
           k="%.4f"%bd['roofline']['kernel_ms_avg'], frac="%.0f"%(100*bd['roofline']['frac']), b1="%.1f"%(1e3*ex['cfg2_B1_ms_per_image']),
           est="%.1f"%(ex['v3_plus_estimate_images_per_s']/1e3), df="%.0f"%(ex['decode_fused_images_per_s']/1e3), du="%.0f"%(ex['decode_unfused_images_per_s']/1e3),
           one="%.1f"%(ex['decode_un_pnp_one_pass_images_per_s']/1e3), two="%.1f"%(ex['decode_un_pnp_two_calls_images_per_s']/1e3),
           dp="%.0f"%(ex['default_path_hn128_maxnum100_images_per_s']/1e3), table=table,
           k_under="%.4f"%under['roofline']['kernel_ms_avg'], calls=ks['k_count_bf16']['Calls'], k_stats="%.4f"%(float(ks['k_count_bf16']['AverageNs'])/1e6),
           cold_scan=us(scan_name), cold_k2=us('k_compact_hyp'), cold_count=us('k_count_bf16'), cold_refit=us('k_select_refit'), cold_fin=us('k_finalize_v3'),
           warm=" + ".join("%.1f"%x for x in warm_parts)+" = %.0f"%sum(warm_parts), ts="%.1f"%(bd['extra']['two_stream_images_per_s']/1e3),
           tr_ms="%.4f"%tr['ms_per_step'])
for k,v in vals.items():
    sec=sec.replace('{{'+k+'}}', v)
assert '{{' not in sec, sec[sec.index('{{'):sec.index('{{')+30]
s=s[:a]+sec+s[b:]
open(p,'w').write(s)
