"""GPU: ABI v8 -- the library's lifecycle (pvv_shutdown + re-initialisation), the caller-provided result buffer (out=), the
PVV_FLAG_DEVICE_RNG promise and the workspace it shrinks, and the count-pass re-run on a workspace made under that flag."""
import ctypes

import pytest
import torch

from tests import capi

pytestmark = pytest.mark.gpu


def test_out_buffer_receives_the_keypoints(synth, pkg, gpu):
    """ransac_voting_layer_v3(..., out=buf): the keypoints land in the caller's [b,vn,2] tensor -- e.g. this rank's rows of a
    persistent gather buffer -- and equal the call without it under the same seed; a wrong shape is refused."""
    from clean_pvnet_amd.dist import GatherBuffer
    from lib.csrc.ransac_voting.ransac_voting_gpu import ransac_voting_layer, ransac_voting_layer_v3
    d = synth.make_batch(**{**synth.CONFIGS["cfg1"], "B": 5}, device=gpu)
    want = ransac_voting_layer_v3(d["mask"], d["vertex"], 64, inlier_thresh=0.99, seed=11)
    buf = GatherBuffer(5, (4, 2), gpu)
    assert buf.mine.data_ptr() == buf.full.data_ptr() and buf.mine.shape == (5, 4, 2)
    got = ransac_voting_layer_v3(d["mask"], d["vertex"], 64, inlier_thresh=0.99, seed=11, out=buf.mine)
    assert got.data_ptr() == buf.mine.data_ptr() and torch.equal(buf.gather(), want)
    part = torch.full((7, 4, 2), -1.0, device=gpu)                         # a slice of a larger buffer: only its rows are written
    ransac_voting_layer(d["mask"], d["vertex"], 64, inlier_thresh=0.99, seed=11, out=part[1:6])
    assert torch.equal(part[1:6], want) and bool((part[0] == -1).all()) and bool((part[6] == -1).all())
    with pytest.raises((RuntimeError, ValueError)):
        ransac_voting_layer_v3(d["mask"], d["vertex"], 64, inlier_thresh=0.99, out=torch.empty(4, 4, 2, device=gpu))
    with pytest.raises(RuntimeError):
        ransac_voting_layer_v3(d["mask"], d["vertex"], 64, inlier_thresh=0.99, out=torch.empty(5, 4, 2))   # a CPU tensor


def test_shutdown_releases_and_the_next_call_reinitialises(synth, pkg, gpu):
    """pvv_shutdown(): after calls that created the library's hidden state -- the pinned stage hint (AUTO on a problem that may
    stage) and the side stream with its events (the deferred mask of the fused decode on a large batch) -- everything is released,
    the hint is forgotten, and the same calls afterwards give the same results again (and build the state anew)."""
    from clean_pvnet_amd import ransac_voting as ext
    B, H, W, K = 24, 300, 404, 3                                           # 2.9 M pixels: the deferred-mask path
    d = synth.make_batch(B=B, H=H, W=W, K=K, fg=0.06, sigma=0.05, seed=77, planar=True, device=gpu)
    x = torch.empty(B, 2 + 2 * K, H, W, device=gpu)
    x[:, 0] = 3.0 * (d["mask"] == 0)
    x[:, 1] = 3.0 * (d["mask"] != 0)
    x[:, 2:] = d["vertex"].permute(0, 3, 4, 1, 2).reshape(B, 2 * K, H, W)
    seg, ver = x[:, :2], x[:, 2:].permute(0, 2, 3, 1).view(B, H, W, K, 2)
    big = synth.make_batch(**{**synth.CONFIGS["cfg3"], "B": 16}, device=gpu)    # K*hn*B*rows reaches the staging bound: hint claimed

    def calls():
        o = ext.decode_keypoint_v3(seg, ver, 256, 0.99, 5, 30000, None, None, 5, ext.SINGULAR_REFERENCE)
        k = ext.ransac_voting_v3(big["mask"], big["vertex"], 512, 0.99, 5, 30000, None, None, 9, ext.SINGULAR_REFERENCE)
        torch.cuda.synchronize()
        return o[0].clone(), o[1].clone(), k[0].clone(), k[1].clone()
    first = calls()
    assert torch.equal(first[1], (d["mask"] != 0).long())
    assert ext.stage_hint(big["mask"], big["vertex"], 512)[0]                  # the hint holds this shape's report
    ext.shutdown()
    assert not ext.stage_hint(big["mask"], big["vertex"], 512)[0]              # forgotten: like the first call of a process
    ext.shutdown()                                                              # twice: nothing left, still fine
    again = calls()
    for a, b in zip(first, again):
        assert torch.equal(a, b)
    assert ext.stage_hint(big["mask"], big["vertex"], 512)[0]                  # built anew
    ext.shutdown()


def test_device_rng_flag_is_a_promise(synth, pkg, gpu):
    """PVV_FLAG_DEVICE_RNG through the C ABI: the lean workspace works for a call that injects nothing (same keypoints as without
    the flag, same seed), and a call that sets the flag and injects index pairs or draws anyway is refused before any launch."""
    L = capi.load()
    d = synth.make_batch(**{**synth.CONFIGS["cfg2"], "B": 2}, device=gpu)
    mask, vertex = d["mask"], d["vertex"]

    def run(flags, idxs=None, selection=None):
        p = capi.problem(mask, vertex, 128, 0.99, seed=5, flags=flags)
        n = L.pvv_workspace_bytes(ctypes.byref(p))
        ws = torch.empty(n, dtype=torch.uint8, device=gpu)
        out = torch.empty(2, 9, 2, device=gpu)
        rc = L.pvv_ransac_voting_v3(ctypes.byref(p), capi.ptr(mask), capi.ptr(vertex), capi.ptr(idxs), capi.ptr(selection), capi.ptr(ws), n,
                                    capi.ptr(out), None, None, capi.stream())
        torch.cuda.synchronize()
        return rc, n, out
    rc0, n0, out0 = run(0)
    rc1, n1, out1 = run(1)
    assert rc0 == 0 and rc1 == 0 and torch.equal(out0, out1)
    assert n0 - n1 == 2 * 150 * 2048 * 4                                    # the draw storage of two 480x640 images
    idxs = torch.zeros(2, 128, 9, 2, dtype=torch.int32, device=gpu)
    rc, _n, _o = run(1, idxs=idxs)
    assert rc == -1 and b"PVV_FLAG_DEVICE_RNG" in L.pvv_last_error()
    sel = torch.rand(2, 480, 640, device=gpu)
    rc, _n, _o = run(1, selection=sel)
    assert rc == -1 and b"PVV_FLAG_DEVICE_RNG" in L.pvv_last_error()


def test_rerun_count_kernel_on_a_lean_workspace(synth, pkg, gpu):
    """ext.rerun_count_kernel rebuilds the workspace layout from the problem: a workspace made by a call that injected nothing
    carries PVV_FLAG_DEVICE_RNG (no draw storage) and the re-run must say so too -- the winners after a re-run of the count pass
    are those of the call."""
    from clean_pvnet_amd import ransac_voting as ext
    d = synth.make_batch(**{**synth.CONFIGS["cfg3"], "B": 4}, device=gpu)
    out, win, _tn, ws = ext.ransac_voting_v3(d["mask"], d["vertex"], 512, 0.99, 5, 30000, None, None, 3, ext.SINGULAR_REFERENCE,
                                             count_kernel=ext.COUNT_FULL)
    assert ws.numel() == ext.workspace_bytes(4, 480, 640, 9, 512, 30000, 8, ext.COUNT_FULL, True)
    assert ws.numel() < ext.workspace_bytes(4, 480, 640, 9, 512, 30000, 8, ext.COUNT_FULL, False)
    for _ in range(2):
        ext.rerun_count_kernel(d["mask"], d["vertex"], 512, 0.99, 5, 30000, ws, True, ext.COUNT_FULL)
    torch.cuda.synchronize()
    # the counts in the workspace are those of a full pass again: select + refit them through a second call's eyes
    out2, win2, _t, _w = ext.ransac_voting_v3(d["mask"], d["vertex"], 512, 0.99, 5, 30000, None, None, 3, ext.SINGULAR_REFERENCE,
                                              count_kernel=ext.COUNT_FULL)
    assert torch.equal(out, out2) and torch.equal(win, win2)
    # a workspace made by a call that INJECTED its index pairs reserves draw storage; re-running it under the default
    # device_rng=True would shift every offset behind the tile lists -- that must be an error, not a count on garbage
    tn = [int(x) for x in (d["mask"] != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, 512, 9).to(gpu)
    out3, win3, _t3, ws3 = ext.ransac_voting_v3(d["mask"], d["vertex"], 512, 0.99, 5, 30000, idxs, None, 3, ext.SINGULAR_REFERENCE,
                                                count_kernel=ext.COUNT_FULL)
    assert ws3.numel() > ws.numel()
    with pytest.raises(RuntimeError, match="device_rng"):
        ext.rerun_count_kernel(d["mask"], d["vertex"], 512, 0.99, 5, 30000, ws3, True, ext.COUNT_FULL)
    ext.rerun_count_kernel(d["mask"], d["vertex"], 512, 0.99, 5, 30000, ws3, True, ext.COUNT_FULL, device_rng=False)
    torch.cuda.synchronize()
