#!/usr/bin/env python
"""The two side paths of the bench line as a small profiling target (tools/profile_bench.sh runs it under rocprofv3):
  * the un_pnp path of resnet18.py:70-72 -- ransac_voting_layer_v3 (512 hypotheses) followed by the estimate (4096) with its
    count pass forced to the FULL kernel: k_count_bf16<0> at 4096 hypotheses, nothing else in this process launches that
    instantiation;
  * the fused decode on the real caller's layout (seg logits + planar vertex: k_tile_scan_seg2, k_mask_from_lists on the
    side stream).
BASELINE config 3 at B = 64, two rotating batches, 12 calls each after a warm-up."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    import lib
    lib._register_clean_pvnet_amd()
    from clean_pvnet_amd import ransac_voting as ext
    from clean_pvnet_amd import synth
    from lib.csrc.ransac_voting.ransac_voting_gpu import estimate_voting_distribution_with_mean, ransac_voting_layer_v3
    dev = torch.device("cuda:0")
    cfg = dict(synth.CONFIGS["cfg3"])
    B, H, W, K, hn = cfg["B"], cfg["H"], cfg["W"], cfg["K"], cfg["hn"]
    gen = {k: v for k, v in cfg.items() if k not in ("B", "hn")}
    batches = [synth.make_batch(B=B, **gen, first_index=r * B, device=dev) for r in range(2)]
    nets = []
    for d in batches:
        x = torch.empty(B, 2 + 2 * K, H, W, device=dev)
        x[:, :2] = torch.randn(B, 2, H, W, device=dev) * 0.1
        x[:, 0] += 3.0 * (d["mask"] == 0)
        x[:, 1] += 3.0 * (d["mask"] != 0)
        x[:, 2:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
        nets.append((x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)))
    for phase in ("warm", "run"):
        n = 6 if phase == "warm" else 12
        for i in range(n):
            d = batches[i % 2]
            mean = ransac_voting_layer_v3(d["mask"], d["vertex"], hn, inlier_thresh=0.99)
            # the estimate's count pass IN FULL (PVV_COUNT_FULL): the 4096-hypothesis k_count_bf16<0> is the VALU-saturated kernel the
            # counters are wanted for; AUTO counts an estimate of this size in stages since round 5, whose launches would mix with
            # v3's of the same names in the trace
            ext.estimate_voting_distribution(d["mask"], d["vertex"], mean, 4096, 0.99, 5, 30000, None, None, 7 + i, False, 0, ext.COUNT_FULL)
        for i in range(n):
            seg, vtx = nets[i % 2]
            ext.decode_keypoint_v3(seg, vtx, hn, 0.99, 5, 30000, None, None, 7 + i, ext.SINGULAR_REFERENCE)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
