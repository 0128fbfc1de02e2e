"""Regenerates DESIGN.md section 5 from profiles/r04_* (tools/design_sections_5.md.tpl)."""
import csv
import json
p = 'DESIGN.md'
s = open(p).read()
a = s.index('## 5. Measurement (bench.py)')
b = s.index('## 6. Multi-GPU')
T = 'r04'
c = json.load(open('profiles/%s_configs.json' % T))['rows']
r3 = json.load(open('profiles/r03_configs.json'))['rows']


def row(name, label):
    v = c[name]
    old = r3.get(name, {}).get('event_ms_per_call_median')
    return "  | %s | %.1f k | %.4f%s | %.4f | %.4f%s |" % (label, v['images_per_s_wall'] / 1e3, v['event_ms_per_call_median'],
                                                         (" (r3: %.4f)" % old) if old else "", v['graph_replay_ms_per_call'],
                                                         v['count_kernel_ms'], ' (staged)' if v.get('count_pass_staged') else '')


table = '\n'.join([
    "  | Config | images/s (wall) | ms/call (events, median) | ms/call (graph replay) | count pass ms (inside calls) |",
    "  |---|---|---|---|---|",
    row('cfg2_B1', '2: 480×640, K=9, 512 hyp, B=1'),
    row('cfg2_dense_tn30000_B1', '2, dense stress: foreground ≈ 30 000 pixels, B=1'),
    row('cfg3_B2', '3: same, B=2'), row('cfg3_B4', '3: same, B=4'),
    row('cfg3_B8_shard_of_8gpu', '3: same, B=8 (a strong-scaling shard of 8 GPUs)'),
    row('cfg3_B16', '3: same, B=16'), row('cfg3_B32', '3: same, B=32'),
    row('cfg3_B64', '3: same, **B=64** (one batch replayed: warm caches)'),
    row('cfg4_B32', '4: sparse/occluded (tn≈1.5 k), 1024 hyp, B=32'), row('cfg4_B4_shard_of_8gpu', '4: same, B=4 (shard of 8 GPUs)'),
    row('cfg5_B16', '5: 540×720, K=17, 2048 hyp, tn capped at 30 000, B=16'), row('cfg5_B2_shard_of_8gpu', '5: same, B=2 (shard of 8 GPUs)'),
    row('default_path_hn128_maxnum100_B64', 'reference default call (resnet18.py:75: 128 hyp, max_num=100), B=64'),
    row('default_path_hn128_maxnum100_B1', 'same, B=1'),
    row('cfg3_B64_planar_vertex', '**real caller\'s layout**: config 3, B=64, planar vertex view (int64 mask given)'),
    row('cfg3_B64_decode_fused', 'same, fused decode: seg logits + planar vertex → mask + keypoints (`pvv_decode_keypoint_v3`)'),
    row('cfg3_B64_decode_unfused', 'same, `torch.argmax` + v3 (what resnet18.py:69-71 runs)'),
    row('cfg2_B1_decode_fused', 'fused decode, B=1'), row('cfg2_B1_decode_unfused', '`torch.argmax` + v3, B=1'),
    row('tless_crop128_B8', '**T-LESS-like detector crops** (SURVEY §8(d)): 128×128, 35 % foreground (tn ≈ 5.7 k), K=9, 512 hyp, B=8'),
    row('tless_crop256_B8', 'same, 256×256 (tn ≈ 22.8 k), B=8 — staged by AUTO on the hint\'s tn (§4.7)'),
    row('tless_crop256_B16_decode_fused', 'same, 256×256, B=16, fused decode (seg logits + planar vertex)'),
])
bd = json.load(open('profiles/%s_bench_default.json' % T))
ex = json.load(open('profiles/%s_bench_extras.json' % T))['extra']
ks = {r['Name']: r for r in csv.DictReader(open('profiles/%s_kernel_stats.csv' % T))}
us = lambda n: float(ks[n]['AverageNs']) / 1e3   # noqa: E731
under = json.load(open('profiles/%s_bench_under_rocprof.json' % T))
pmc = json.load(open('profiles/call_pmc.json'))
sec = open('tools/design_sections_5.md.tpl').read()
passk = ['k_count_bf16<1>', 'k_lead', 'k_count_filter_runs']
parts = [us(k) for k in passk]
callk = ['k_tile_scan', 'k_compact_hyp'] + passk + ['k_select_refit', 'k_finalize_v3']
cold = [us(k) for k in callk]
cb = bd['cpu_baseline']
rs, rc, rl, rv, e = bd['roofline_scan'], bd['roofline_compact'], bd['roofline'], bd['roofline_valu'], bd['extra']
K = pmc['kernels']
nz, est = e['noisy_field'], e['estimate_4096_count_pass']
vals = dict(
    v="%.1f" % (bd['value'] / 1e3), ms="%.4f" % bd['ms_per_step'], med="%.4f" % bd['step_ms']['median'], p10="%.4f" % bd['step_ms']['p10'],
    p90="%.4f" % bd['step_ms']['p90'], k="%.4f" % bd['roofline_contract_count_pass']['kernel_ms_avg'],
    contract_frac="%.2f" % bd['roofline_contract_count_pass']['frac_not_a_bound'],
    sum_stages="%.4f" % e['kernels_inside_calls_ms']['sum_avg_ms'],
    k_under="%.4f" % under['roofline_contract_count_pass']['kernel_ms_avg'], k_stats="%.4f" % (sum(parts) / 1e3),
    k_stats_parts=" + ".join("%.1f" % x for x in parts) + " µs",
    call_gbs="%.0f" % rl['achieved'], call_frac="%.2f" % rl['frac'], call_vs_probe="%.2f" % rl['frac_of_stream_read_this_box'],
    probe="%.0f" % e['stream_read_probe']['GBs'],
    traffic_mb="%.0f" % (rl['traffic'] / 1e6), traffic_frac="%.2f" % rl['traffic_frac'],
    traffic_parts=", ".join("%s %.1f" % (k.replace('k_', '').replace('count_bf16<1>', 'first').replace('count_filter_runs', 'second launch'), K[k]['hbm_bytes'] / 1e6) for k in callk),
    scan_us="%.1f" % (rs['ms_avg'] * 1e3), scan_gbs="%.0f" % rs['achieved'], scan_frac="%.2f" % rs['frac'],
    scan_of_probe="%.2f" % rs['frac_of_stream_read'], scan_traffic_mb="%.1f" % (rs['traffic'] / 1e6),
    scan_tbs_prof="%.1f" % (rs['bytes'] / (cold[0] * 1e-6) / 1e12),
    cmp_us="%.1f" % (rc['ms_avg'] * 1e3), cmp_alg_mb="%.0f" % (rc['algorithmic_bytes'] / 1e6),
    cmp_traffic_mb="%.0f" % (rc['traffic'] / 1e6), cmp_gbs="%.0f" % rc['achieved'],
    valu_issued="%.1f" % (rv['issued_valu_wave_instructions'] / 1e6),
    valu_parts=" + ".join("%.1f" % (K[k]['SQ_INSTS_VALU'] / 1e6) for k in passk), valu_frac="%.2f" % rv['frac'],
    valu_busy="%.2f" % rv['busy_frac'], valu_busy_parts=" / ".join("%.2f" % K[k]['valu_busy'] for k in passk),
    est_busy="%.2f" % min(1.0, pmc['estimate_4096']['valu_busy']),
    cold_scan="%.1f" % cold[0], cold_k2="%.1f" % cold[1], cold_first="%.1f" % cold[2], cold_lead="%.1f" % cold[3],
    cold_filter="%.1f" % cold[4], cold_refit="%.1f" % cold[5], cold_fin="%.1f" % cold[6], cold_sum="%.0f" % sum(cold),
    rho_noisy="%.3f" % nz['mean_winner_ratio_rho'], thr_noisy="%.3f" % nz['auto_stage_threshold'], v_noisy="%.1f" % (nz['images_per_s'] / 1e3),
    noisy_ratio="%.2f" % nz['vs_clean_headline'],
    ts="%.1f" % (e['two_stream_images_per_s'] / 1e3), b1="%.1f" % (1e3 * ex['cfg2_B1_ms_per_image']),
    est="%.1f" % (e['v3_plus_estimate_images_per_s'] / 1e3), one="%.1f" % (e['un_pnp_fused_one_pass_images_per_s'] / 1e3),
    est_ms="%.3f" % est['ms_inside_calls_median'], est_tevals="%.1f" % est['T_evaluations_per_s'],
    est_issued="%.0f" % (pmc['estimate_4096']['SQ_INSTS_VALU'] / 1e6),
    df="%.1f" % (e['decode_fused_images_per_s'] / 1e3), df_ratio="%.2f" % e['decode_fused_vs_headline'],
    du="%.1f" % (e['decode_unfused_argmax_plus_v3_images_per_s'] / 1e3),
    dp="%.0f" % (ex['default_path_hn128_maxnum100_images_per_s'] / 1e3),
    table=table, cfg5_ms="%.4f" % c['cfg5_B16']['event_ms_per_call_median'],
    ref_ratio="%.0f" % (13.0 / bd['roofline_contract_count_pass']['kernel_ms_avg']),
    cpu1="%.1f" % cb['single_thread']['value'], cpuN="%.0f" % cb['value'], cpu_cores=str(cb['cores']), cpu_model=cb['cpu_model'],
    cpu_probe=", ".join("%d → %.0f" % (t['threads'], t['images_per_s']) for t in cb['thread_probe']['table']),
    cpu_diff="%.2g" % cb['same_idxs_gpu_check']['means_max_abs_diff'])
for k, v in vals.items():
    sec = sec.replace('{{' + k + '}}', v)
assert '{{' not in sec, sec[sec.index('{{'):sec.index('{{') + 30]
s = s[:a] + sec + s[b:]
open(p, 'w').write(s)
print("section 5 regenerated from profiles/%s_*" % T)
