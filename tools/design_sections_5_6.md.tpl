## 5. Measurement (bench.py)

* Step = one `ransac_voting_layer_v3(mask, vertex, 512, inlier_thresh=0.99)` through the drop-in Python API,
  device-resident synthetic batch, device RNG, + (whenever a process group exists) the RCCL `all_gather` of `[B,K,2]`,
  completed inside the step it belongs to.  Workload = BASELINE config 3: **global batch 64**; N = 1 puts it on one GPU
  (the configuration the roofline target is quoted on), N > 1 is **strong scaling** — the same 64 images in contiguous
  shards of 64/N (`"scaling": "strong"`); the weak figure (64 per GPU) and the variant that overlaps the exchange with the
  next step are in `extra`.  `--extras` adds config 2 (B = 1 latency), v3+estimate, the default path, `decode_keypoint`.
* Protocol (SURVEY §8(d), VERDICT r1 #3): steps cycle over **3 distinct device-resident batches** (4.7 GB: neither the
  256 MiB Infinity Cache nor L2 holds a step's inputs from the step before — the mask scan then runs at 4.4 instead of
  5.2 TB/s and the step costs ≈ +10 µs against replaying one batch); **per-step HIP events** on the launch stream →
  `step_ms` median / p10 / p90 beside the contract's wall-clock `ms_per_step`; a disclosed **clock pre-warm** (60 ms of
  untimed steps before the W warm-up steps): after the GPU-idle data generation the chip needs ≈60 steps to reach steady
  clocks (`tools/clock_ramp.py`: 0.335 ms/step at step 8, 0.300 at step 30, 0.279 from step 60 on), so a 25-step run
  timed the ramp.  That is the explanation of round 1's "driver 0.3122 ms vs README 0.2727 ms" (VERDICT r1 weak #3): the
  driver's K = 20 / W = 5 run sat in the ramp, the 200-step run mostly beyond it, and the traced kernel sum of 0.307 ms
  came from a 20-step run as well — the 0.273 was not "below the sum of its own kernels", it was a later part of the
  ramp.  With the pre-warm a 20-step/5-warm-up run and a 200-step run agree within 1.5 % (0.2780 vs 0.2745 ms, one box).
* `roofline.achieved` = ALGORITHMIC bytes per launch ÷ the count kernel's average duration measured with HIP
  events on the launch stream around re-launches of that kernel alone (`pvv_rerun_count_kernel`).
  Algorithmic bytes = SURVEY §8(d)'s dense-field figure, 22 480 896 B/image (`[H,W,K,2]` f32 + u8 mask +
  hypotheses in + int32 counts out) × 64 images = **1 438 777 344 B per launch** (the conservative u8-mask
  figure although the bench feeds the int64 mask `argmax` emits). `peak` = 8000 GB/s (HBM3E spec).
  `traffic` = `(FETCH_SIZE + WRITE_SIZE)·1024` from separate `rocprofv3 --pmc` passes of the same command — a **static**
  figure read from `profiles/count_kernel_pmc.json` (`traffic_source` says so): 52 MB per launch, far *below* the
  algorithmic bytes because the kernel reads the compacted foreground only; no wasted re-reads.
* `roofline_valu` — the engineering figure: evaluations/s against the VALU-issue ceiling of the steady-state loop
  (1024 SIMDs × device max clock × 512 evaluations per matrix-core tile ÷ (21 VALU × 4.2 cycles)).  0.69 at the nominal
  2.4 GHz (the profiled box ran the kernel at 2.39 GHz); the rest is prologues, flagged tiles (9 % at B = 64, 17 % at
  B = 8: `tools/band_cost.sh`) and the ends of the kernel (`SQ_ACTIVE_INST_VALU`: 97 % busy, `profiles/r02_summary.json`).
* Round-2 numbers (MI355X, `profiles/r02_*`, one box, rotating batches): **{{v}} k images/s** at B = 64
  ({{ms}} ms/step wall, {{med}} median, p10/p90 {{p10}}/{{p90}}); count kernel {{k}} ms by HIP events ({{k_under}} ms in the
  process that `rocprofv3 --kernel-trace --stats` profiled, whose own average over all {{calls}} launches — ramp included — is
  {{k_stats}} ms: `r02_bench_under_rocprof.json`, `r02_kernel_stats.csv`) ⇒ **{{frac}} % of the HBM roofline** (target ≥ 40 %).
  The kernel is timed after the same clock pre-warm as the steps: measured right after an idle moment it read 0.217 ms
  on the same box (and that is what round 1's 0.2232 ms was).  Per call in the profiled run (cold inputs):
  `k_tile_scan` {{cold_scan}} µs (39.1 before it became persistent with read-ahead), `k_compact_hyp` {{cold_k2}}, `k_count_bf16` {{cold_count}},
  `k_select_refit` {{cold_refit}}, `k_finalize_v3` {{cold_fin}}; replaying one warm batch without pre-warm
  (`profiles/r02_gaps_cfg3_B64.json`): {{warm}} µs (round 1, same protocol: 28.8 + 24.5 + 11.7 + 218.9 + 15.2 + 4.6 =
  304 µs).  The same steps alternating over two streams, as a caller decoding a sequence of batches can issue them
  (`clean_pvnet_amd.pipeline.StreamRing`; `extra.two_stream_images_per_s`, never `value`): **{{ts}} k images/s**.  Extras
  (`r02_bench_extras.json`): B = 1 latency {{b1}} µs/call (round 1: 38); v3 + estimate (4096 hypotheses) {{est}} k images/s;
  fused `decode_keypoint` {{df}} k vs {{du}} k images/s for `torch.argmax` + v3; un_pnp one pass {{one}} k vs {{two}} k; the
  reference's default non-`un_pnp` call {{dp}} k images/s (round 1: 769 k).  Host-buffer note: the boundary takes device
  pointers; a caller holding the 1.57 GB batch in host memory would be PCIe-bound at 63 GB/s ≈ 2.6 k images/s — never
  the reported value.
* All BASELINE configs and the shards the 8-GPU split produces, one MI355X, default kernel (`profiles/r02_configs.json`
  from `tools/config_bench.py`; one batch replayed — warm caches —, `frac` = dense-field bytes ÷ kernel time ÷ 8 TB/s):

{{table}}

  Host side: one call costs 27–32 µs of host time on an idle stream (`host_ms_per_call_idle_stream`: 5 launches ≈ 3.5 µs
  each + tensor allocation), so below B ≈ 2 the eager wall clock is host-bound; a captured graph removes that (the
  replay column: the GPU side is what remains).  Per-kernel at B = 1 (`r02_gaps_cfg2_B1.json`): scan 5.4, compact +
  hypotheses 7.2, count 12.8, refit 5.2, finalize 4.0 µs.  **The small-batch targets of VERDICT r1 #1 (B = 1 ≤ 20 µs,
  B = 8 ≤ 45 µs, default path ≥ 1.5 M images/s) are NOT met**: B = 1 went 40.3 → 34 µs, B = 8 70 → 64 µs (the item-size sweep of
  §4.4 took 5 µs off its count kernel, the merged front end 1–3 µs), the default path went 730 k → 900–950 k images/s.  What was learned trying (§4.4, §4.5): a kernel boundary costs 1.45 µs, but every *dependent
  phase* — barrier + memory round trip — costs 2.5–5 µs whether it is its own launch or a phase of a fused kernel
  (tickets, in-kernel hand-offs and a fully fused back end were built, are bit-exact, and are not faster); the path has
  five such phases between the mask and the keypoints (scan → prefix/compaction → counts → arg-max + refit → policy over
  the keypoints), and a shard of 8 images is a third of one generation of the count kernel's work items.
* The reference's own kernel on the same GPU (`oracle/_ref`, `tests/test_ref_pin.py::test_reference_kernel_timed_on_the_same_gpu`):
  `voting_for_hypothesis_kernel` + `torch.sum` for ONE 480×640 image (K = 9, 512 hypotheses, what P:155-159 runs per
  image and round) takes 0.21 ms on the MI355X, i.e. 13 ms for the 64 images that `k_count_bf16` counts in 0.2 ms
  (×60), with identical counts.
* `cpu_baseline` = the oracle ("port": the reference has no CPU path) on the box's host cores via OpenMP,
  ~12 s of single-image `ransac_voting_layer_v3` calls over 8 of the timed images, with the OpenMP thread count
  that looked fastest on the box (visible CPUs ≠ usable CPUs under cgroup quotas): 38–177 images/s on 16–128 threads of
  an EPYC 9575F, depending on what else the host runs. A baseline, not a target.

## 6. Multi-GPU

Images are independent units ⇒ contiguous batch shards, one process per GPU, no data-path collective; the
one exchange is an `all_gather_into_tensor` of the per-image results (`clean-pvnet_amd/dist.py`: 72 B of
means (+144 B of covariance) per image — latency-bound, xGMI bandwidth irrelevant). `nccl` = RCCL on the
GPUs; the identical code runs under `gloo` in `tests/test_dist.py` (world sizes 2 and 3, uneven shards, a rank without
any image — it enters the collective with zero rows, ADVICE r1). With `sharded_vote(..., seed=s)` every rank votes with
the common key and `first_image` = the index of its first image, so the device-RNG result of image i is the same for
every number of GPUs (`test_python_layers_are_invariant_to_sharding_with_a_common_seed`). The reference has no
counterpart (its only multi-GPU mechanism is `nn.DataParallel` for training).

What has executed on hardware (one GPU is all `gpurun` gives): a real RCCL process group of ONE rank —
`tests/test_gpu_dist.py` launches `bench.py` exactly as the driver launches it for N > 1 (`python -m
torch.distributed.run --nproc-per-node 1 … bench.py --gpus 1`: `init_process_group("nccl", device_id=…)`, an
`all_gather_into_tensor` in every step, barrier + max-over-ranks timing; `rccl_ranks` in the JSON comes from the
collective's own result) and runs the HIP layer under `sharded_vote` in a one-rank `nccl` group against the un-sharded
call (`profiles/r02_bench_torchrun_1rank.json`: {{tr_ms}} vs {{ms}} ms/step without the group).  No scaling curve could be
measured here.  **Expectation for the driver's strong-scaling run of config 3** from the single-GPU shard timings above:
64 images take 0.271 ms on one GPU; a shard of 8 takes 0.064 ms (+ the ≈10–30 µs exchange) ⇒ speed-up ≈ 3.0–3.6× on
8 GPUs, efficiency ≈ 40–45 %, because a shard of 8 images is latency-bound (§5): near-linear *weak* scaling (64 images per
GPU: the exchange is the only addition), not near-linear strong scaling at this problem size.  `num_cus()` is per
device now.

