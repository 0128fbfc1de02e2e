## 5. Measurement (bench.py)

* Step = one `ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)` through the drop-in Python API,
  device-resident synthetic batch, device RNG, + (whenever a process group exists) the RCCL `all_gather` of `[B,K,2]`
  (§6).  Workload = BASELINE config 3, **64 images per GPU**: N = 1 is the configuration the roofline target is quoted on.
  The DEFAULT line also carries (outside the timed region, each timed like it): the same call on fields with 9.5 % outlier
  pixels (`value_at_rho_0.90`), the path `cfg.test.un_pnp` runs (v3 + the 4096-hypothesis estimate, as the reference's two
  calls and as one fused pass), and the fused decode on the layout the real caller passes (§7).  `--extras` adds config 2
  (B = 1 latency), the default path, the uncertainty PnP, the ADD-S search.
* Protocol (SURVEY §8(d), VERDICT r1 #3): steps cycle over **3 distinct device-resident batches** (4.7 GB: neither the
  256 MiB Infinity Cache nor L2 holds a step's inputs from the step before); **per-step HIP events** on the launch stream →
  `step_ms` median / p10 / p90 beside the contract's wall-clock `ms_per_step`; a disclosed **clock pre-warm** (60 ms of
  untimed steps before the W warm-up steps: after the GPU-idle data generation the chip needs ≈ 60 steps to reach steady
  clocks, `tools/clock_ramp.py`).  With it a 20-step run and a 200-step run agree within 1.5 %.
* **The kernels' durations come from inside the calls** (VERDICT r2 #3c): the library records a HIP event at every stage
  boundary of a call when asked to (`pvv_problem.ev_marks`), `bench.py` asks for that in 36 calls cycling over the
  rotating batches after the same pre-warm and averages the last 30 — the sample `rocprofv3 --kernel-trace` sees (the
  records themselves cost ≈ 1–2 µs per stage: the sum of the stages, {{sum_stages}} ms, reads that much above the step).
  In the profiled process of `profiles/r04_*` the two agree: count pass {{k_under}} ms by events, {{k_stats}} ms =
  {{k_stats_parts}} by `rocprofv3 --stats` (`r04_kernel_stats.csv`, all launches, ramp included).
* **The roofline blocks of the line, all bounded and reproducible by hand from `profiles/`** (VERDICT r3 #1; rounds 1–3 led with
  the dense-field bytes ÷ the count PASS, a figure that exceeded 1 once the pass stopped touching the dense field):
  * `roofline` = the WHOLE CALL against HBM: SURVEY §8(d)'s dense-field bytes of the batch, 22 480 896 B/image × 64 =
    **1 438 777 344 B** (`[H,W,K,2]` f32 + u8 mask + hypotheses in + int32 counts out — the conservative u8-mask figure although
    the bench feeds the int64 mask `argmax` emits) ÷ `ms_per_step` = **{{call_gbs}} GB/s = {{call_frac}} of the 8 TB/s peak**
    (round 3: 0.87, round 2: 0.66).  The call consumes the dense field exactly once, so this is bounded by the peak as long as
    the call is slower than one streaming read of its input at peak — it is {{call_vs_probe}}× FASTER than the box's own
    read-once stream of those bytes ({{probe}} GB/s, `pvv_stream_read_probe` in the same run), because after the first two kernels
    it works on the compacted 2 % of the field.  `traffic` = what the call's seven kernels really move, Σ per-kernel
    `(fetch_correction · FETCH_SIZE + WRITE_SIZE) · 1024` from separate `rocprofv3 --pmc` passes (`profiles/call_pmc.json`,
    static, FETCH_SIZE doubled where MI355X_MICROARCH.md prescribes it: the wide streaming reads) = **{{traffic_mb}} MB** per
    step ({{traffic_parts}}) → `traffic_frac` = **{{traffic_frac}}**: the call is not HBM-bound.
  * `roofline_scan`: `k_tile_scan` moves the 157 MB int64 mask in {{scan_us}} µs = {{scan_gbs}} GB/s = {{scan_frac}} of spec,
    **{{scan_of_probe}} of the box's streaming rate** (in-call events, ≈ 2 µs above `rocprofv3`'s {{cold_scan}} µs = {{scan_tbs_prof}} TB/s);
    counter bytes {{scan_traffic_mb}} MB = the known byte count, no re-reads.
  * `roofline_compact`: `k_compact_hyp` in {{cmp_us}} µs for {{cmp_alg_mb}} MB that must move ({{cmp_traffic_mb}} MB counted:
    72-byte pixel records fetched in 32-byte sectors) = {{cmp_gbs}} GB/s — a queue of short-lived gather blocks bound by the
    ≈ 450-cycle latency of their L2 requests, not by their number (§4.5: the planar layout cuts the L1 accesses by 60 % and the
    duration by 5 %).
  * `roofline_valu`: the count pass on ISSUED instructions — `SQ_INSTS_VALU` of its kernels ({{valu_issued}} M wave-instructions:
    {{valu_parts}}; round 3: 57.1 M) ÷ its duration inside calls against 1024 SIMDs × 2.4 GHz / 2 cycles (§4.1) =
    **{{valu_frac}}**; `busy_frac` = the counter figure proper, Σ `SQ_ACTIVE_INST_VALU`·4 / 1024 ÷ Σ `GRBM_GUI_ACTIVE`/8 over the
    pass = **{{valu_busy}}** ({{valu_busy_parts}}) — the share of SIMD cycles in which a VALU instruction was executing; the
    4096-hypothesis full pass of the estimate, whose loop is not interrupted by per-chunk prologues, reaches {{est_busy}}.
  * `roofline_contract_count_pass` keeps the figure of rounds 1–3 for continuity (dense-field bytes ÷ the count pass:
    {{contract_frac}}, `frac_not_a_bound`).
* Round-4 numbers (MI355X, `profiles/r04_*`, one box, rotating batches): **{{v}} k images/s** at B = 64
  ({{ms}} ms/step wall, {{med}} median, p10/p90 {{p10}}/{{p90}}; round 3: 310.6 k, 0.2061 ms; round 2: 235.6 k).  Per call in
  the profiled run (cold inputs, `r04_kernel_stats.csv`): `k_tile_scan` {{cold_scan}} µs, `k_compact_hyp` {{cold_k2}},
  `k_count_bf16<first>` {{cold_first}}, `k_lead` {{cold_lead}}, `k_count_filter_runs` {{cold_filter}}, `k_select_refit`
  {{cold_refit}}, `k_finalize_v3` {{cold_fin}} = {{cold_sum}} µs (round 3: 26.9 + 34.6 + 56.4 + 10.7 + 59.5 + 12.8 + 4.9 = 206).
  **On fields with 9.5 % outlier pixels** (ρ = {{rho_noisy}}; AUTO stages: threshold {{thr_noisy}}): **{{v_noisy}} k images/s**
  = {{noisy_ratio}} of the clean figure (round 3's kernels and threshold: 249.8 k).  The same steps alternating over two
  streams, as a caller decoding a sequence of batches can issue them (`clean_pvnet_amd.pipeline.StreamRing`;
  `extra.two_stream_images_per_s`, never `value`): **{{ts}} k images/s**.
  **The `cfg.test.un_pnp` path** (resnet18.py:70-72: v3 + the 4096-hypothesis estimate, always counted in full): {{est}} k images/s
  as the reference's two calls, {{one}} k as one fused pass on seg logits + planar vertex; its count kernel (`k_count_bf16<0>`,
  4096 hypotheses) runs {{est_ms}} ms inside calls = {{est_tevals}} T evaluations/s, {{est_issued}} M issued VALU
  wave-instructions, VALU-busy {{est_busy}} (`call_pmc.json` `estimate_4096`).  **Fused decode on the real caller's layout** (§7):
  **{{df}} k images/s = {{df_ratio}} of the headline** (round 3: 254 k = 0.83), against {{du}} k for `torch.argmax` + v3.
  Extras (`r04_bench_extras.json`): B = 1 latency {{b1}} µs/call; the reference's default non-`un_pnp` call {{dp}} k images/s.
  Host-buffer note: the boundary takes device pointers; a caller holding the 1.57 GB batch in host memory would be PCIe-bound at
  63 GB/s ≈ 2.6 k images/s — never the reported value.
* All BASELINE configs, the shards an 8-GPU strong split produces and the real caller's layouts, one MI355X, `AUTO` count mode
  (`profiles/r04_configs.json` from `tools/config_bench.py`; one batch replayed — warm caches; count pass = its duration
  inside the calls):

{{table}}

  Config 5 (540×720, K = 17, 2048 hypotheses) gained the most: 0.8201 → {{cfg5_ms}} ms per call (the eighth first stage and
  the pooled misses: 44 remaining chunks per image).  Host side: one call costs 27–32 µs of host time on an idle stream, so below
  B ≈ 2 the eager wall clock is host-bound; a captured graph removes that (the replay column; the fused decode's replay is slower
  than its eager call because no side stream is used under capture, §7).  Small batches (B ≤ 8) did not move this round: the five
  dependent phases at 2.5–5 µs each stand (§4.4, §4.5; the single-walker front of VERDICT r3 #7 cannot stream, §4.5).
* The reference's own kernel on the same GPU (`oracle/_ref`, `tests/test_ref_pin.py::test_reference_kernel_timed_on_the_same_gpu`):
  `voting_for_hypothesis_kernel` + `torch.sum` for ONE 480×640 image (K = 9, 512 hypotheses, what P:155-159 runs per
  image and round) takes 0.21 ms on the MI355X, i.e. 13 ms for the 64 images whose winners the staged pass finds in
  {{k}} ms (×{{ref_ratio}}), with identical winner counts.
* `cpu_baseline` (SURVEY §8(d)) = the oracle ("port": the reference has no CPU path) on the box's host: **one thread**
  {{cpu1}} images/s and **OpenMP** {{cpuN}} images/s on {{cpu_cores}} threads of an {{cpu_model}} — the thread count a probe picked
  ({{cpu_probe}}: visible CPUs ≠ usable CPUs under cgroup quotas, and the figure differs from box to box: round 3's driver box
  read 112 on 32 threads) — and the SAME 8 images with the SAME injected index pairs once through the GPU path in the same run:
  winner counts equal, means within the contract (`same_idxs_gpu_check`: max |Δ| = {{cpu_diff}} px).  A baseline, not a target.

