"""Regenerates DESIGN.md sections 5-6 from profiles/r03_* (tools/design_sections_5_6.md.tpl)."""
import csv
import json
p = 'DESIGN.md'
s = open(p).read()
a = s.index('## 5. Measurement (bench.py)')
b = s.index('## 7. SURVEY')
c = json.load(open('profiles/r03_configs.json'))['rows']
def row(name, label, note=''):
    v = c[name]
    return "| %s | %.1f k | %.4f | %.4f | %.4f%s | %.2f%s |" % (label, v['images_per_s_wall'] / 1e3, v['event_ms_per_call_median'], v['graph_replay_ms_per_call'],
                                                            v['count_kernel_ms'], ' (staged)' if v.get('count_pass_staged') else '', v['roofline_frac_hbm_8TBs'], note)
table = '\n'.join([
 "  | Config | images/s (wall) | ms/call (events, median) | ms/call (graph replay) | count pass ms (inside calls) | frac |",
 "  |---|---|---|---|---|---|",
 "  " + row('cfg2_B1', '2: 480×640, K=9, 512 hyp, B=1', ' (round 2: 0.0339 ms)'),
 "  " + row('cfg2_dense_tn30000_B1', '2, dense stress: foreground ≈ 30 000 pixels, B=1', ' (SURVEY §8(d))'),
 "  " + row('cfg3_B2', '3: same, B=2'),
 "  " + row('cfg3_B4', '3: same, B=4'),
 "  " + row('cfg3_B8_shard_of_8gpu', '3: same, B=8 (a strong-scaling shard of 8 GPUs)', ' (round 2: 0.0642 ms)'),
 "  " + row('cfg3_B16', '3: same, B=16'),
 "  " + row('cfg3_B32', '3: same, B=32', ' (round 2: 0.1533 ms)'),
 "  " + row('cfg3_B64', '3: same, **B=64** (one batch replayed: warm caches)', ' (round 2: 0.2612 ms, count kernel 0.1901)'),
 "  " + row('cfg4_B32', '4: sparse/occluded (tn≈1.5 k), 1024 hyp, B=32', ' (round 2: 0.1016 ms)'),
 "  " + row('cfg4_B4_shard_of_8gpu', '4: same, B=4 (shard of 8 GPUs)'),
 "  " + row('cfg5_B16', '5: 540×720, K=17, 2048 hyp, tn capped at 30 000, B=16', ' (round 2: 1.5847 ms, count kernel 1.4536)'),
 "  " + row('cfg5_B2_shard_of_8gpu', '5: same, B=2 (shard of 8 GPUs)'),
 "  " + row('default_path_hn128_maxnum100_B64', 'reference default call (resnet18.py:75: 128 hyp, max_num=100), B=64'),
 "  " + row('default_path_hn128_maxnum100_B1', 'same, B=1'),
])
bd = json.load(open('profiles/r03_bench_default.json'))
ex = json.load(open('profiles/r03_bench_extras.json'))['extra']
ks = {r['Name']: r for r in csv.DictReader(open('profiles/r03_kernel_stats.csv'))}
us = lambda n: float(ks[n]['AverageNs']) / 1e3
under = json.load(open('profiles/r03_bench_under_rocprof.json'))
tr = json.load(open('profiles/r03_bench_torchrun_1rank.json'))
pmc = json.load(open('profiles/count_kernel_pmc.json'))
front = json.load(open('profiles/front_kernels_pmc.json'))
sec = open('tools/design_sections_5_6.md.tpl').read()
parts = [us('k_count_bf16<1>'), us('k_lead'), us('k_count_bf16<2>')]
cold = [us('k_tile_scan'), us('k_compact_hyp')] + parts + [us('k_select_refit'), us('k_finalize_v3')]
cb = bd['cpu_baseline']
rs, rc = bd['roofline_scan'], bd['roofline_compact']
vals = dict(v="%.1f" % (bd['value'] / 1e3), ms="%.4f" % bd['ms_per_step'], med="%.4f" % bd['step_ms']['median'], p10="%.4f" % bd['step_ms']['p10'],
            p90="%.4f" % bd['step_ms']['p90'], k="%.4f" % bd['roofline']['kernel_ms_avg'], frac="%.2f" % bd['roofline']['frac'],
            sum_stages="%.4f" % bd['extra']['kernels_inside_calls_ms']['sum_avg_ms'],
            k_under="%.4f" % under['roofline']['kernel_ms_avg'], k_stats="%.4f" % (sum(parts) / 1e3),
            k_stats_parts=" + ".join("%.1f" % x for x in parts) + " µs",
            traffic_mb="%.0f" % (pmc['hbm_bytes_per_launch'] / 1e6),
            call_frac="%.2f" % bd['roofline_call']['frac'], call_gbs="%.0f" % bd['roofline_call']['achieved'],
            probe="%.0f" % bd['extra']['stream_read_probe']['GBs'],
            read_ms="%.3f" % (bd['roofline']['algorithmic_bytes'] / (bd['extra']['stream_read_probe']['GBs'] * 1e9) * 1e3),
            scan_us="%.1f" % (rs['ms_avg'] * 1e3), scan_gbs="%.0f" % rs['achieved'], scan_frac="%.2f" % rs['frac'],
            scan_of_probe="%.2f" % rs['frac_of_stream_read'],
            cmp_us="%.1f" % (rc['ms_avg'] * 1e3), cmp_alg_mb="%.0f" % (rc['algorithmic_bytes'] / 1e6),
            cmp_traffic_mb="%.0f" % (front['k_compact_hyp']['hbm_bytes_per_launch'] / 1e6), cmp_gbs="%.0f" % rc['achieved'],
            valu_frac="%.2f" % bd['roofline_valu']['frac'],
            cold_scan="%.1f" % cold[0], cold_k2="%.1f" % cold[1], cold_first="%.1f" % cold[2], cold_lead="%.1f" % cold[3],
            cold_filter="%.1f" % cold[4], cold_refit="%.1f" % cold[5], cold_fin="%.1f" % cold[6], cold_sum="%.0f" % sum(cold),
            ts="%.1f" % (bd['extra']['two_stream_images_per_s'] / 1e3), b1="%.1f" % (1e3 * ex['cfg2_B1_ms_per_image']),
            est="%.1f" % (ex['v3_plus_estimate_images_per_s'] / 1e3), df="%.0f" % (ex['decode_fused_images_per_s'] / 1e3),
            du="%.0f" % (ex['decode_unfused_images_per_s'] / 1e3), one="%.1f" % (ex['decode_un_pnp_one_pass_images_per_s'] / 1e3),
            two="%.1f" % (ex['decode_un_pnp_two_calls_images_per_s'] / 1e3), dp="%.0f" % (ex['default_path_hn128_maxnum100_images_per_s'] / 1e3),
            table=table, cfg4_ms="%.4f" % c['cfg4_B32']['event_ms_per_call_median'],
            ref_ratio="%.0f" % (13.0 / bd['roofline']['kernel_ms_avg']),
            cpu1="%.1f" % cb['single_thread']['value'], cpuN="%.0f" % cb['value'], cpu_cores=str(cb['cores']), cpu_model=cb['cpu_model'],
            cpu_diff="%.2g" % cb['same_idxs_gpu_check']['means_max_abs_diff'], tr_ms="%.4f" % tr['ms_per_step'])
for k, v in vals.items():
    sec = sec.replace('{{' + k + '}}', v)
assert '{{' not in sec, sec[sec.index('{{'):sec.index('{{') + 30]
s = s[:a] + sec + "\n" + s[b:]
# section 4.6: the three launches of the staged pass (last cell of their table rows) and their sum
import re
for key, val in (("`k_count_bf16<first>` |", parts[0]), ("`k_lead` (`count_prune.hpp`) |", parts[1]), ("`k_count_bf16<filter>` |", parts[2])):
    i = s.index("| " + key)
    j = s.index("\n", i)
    line = s[i:j]
    line = re.sub(r"\| [^|]*\|$", "| %.1f |" % val, line)
    s = s[:i] + line + s[j:]
s = re.sub(r"Sum [0-9.]+ µs against 187 µs for the full kernel on the same box \(−[0-9]+ %\)",
           "Sum %.0f µs against 187 µs for the full kernel on the same box (−%.0f %%)" % (sum(parts), 100 * (1 - sum(parts) / 187.0)), s)
# section 4.6: the full-vs-staged table from profiles/r03_staged_ab.json
ab = {r['case']: r for r in json.load(open('profiles/r03_staged_ab.json'))}
notes = {'cfg3:8': 'not staged by AUTO (two extra launches on a latency-bound call)', 'cfg3:16': 'staged by AUTO from here on (2.3·10¹⁰)',
         'cfg3:32': '', 'cfg3:64': '**the benchmark workload**',
         'cfg4:32': 'nothing to stage (tn ≈ 0.5–2 k: every image below 8 chunks): forced, the first launch counts everything and `k_lead` and the filter launch find the `any_staged` word 0 and leave at once; AUTO learns the images\' `tn` (and ρ = 0.63) from the stage hint and does not stage from the second call on',
         'cfg5:16': '540×720, K = 17, 2048 hypotheses, tn = 30 000'}
label = {'cfg3:8': 'config 3, B = 8', 'cfg3:16': 'config 3, B = 16', 'cfg3:32': 'config 3, B = 32', 'cfg3:64': 'config 3, **B = 64**',
         'cfg4:32': 'config 4, B = 32', 'cfg5:16': 'config 5, B = 16'}
rows = ["| Case | full ms/call | staged ms/call | AUTO ms/call | staged vs full | |", "|---|---|---|---|---|---|"]
for k in ('cfg3:8', 'cfg3:16', 'cfg3:32', 'cfg3:64', 'cfg4:32', 'cfg5:16'):
    r = ab[k]
    rows.append("| %s | %.4f | %.4f | %.4f | %+.0f %% | %s |" % (label[k], r['full']['ms_per_call'], r['staged']['ms_per_call'], r['auto']['ms_per_call'],
                                                               100 * (r['full']['ms_per_call'] / r['staged']['ms_per_call'] - 1), notes[k]))
ta, tb = s.index('<!-- staged_ab_table -->'), s.index('<!-- /staged_ab_table -->')
s = s[:ta] + '<!-- staged_ab_table -->\n' + "\n".join(rows) + '\n' + s[tb:]
open(p, 'w').write(s)
