"""The N > 1 path on CPU: two gloo ranks shard a batch, each votes on its shard (here with the ORACLE standing in for
the GPU layer -- tests may do that, the product may not) and the gathered result must equal the unsharded one."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lib
        lib._register_clean_pvnet_amd()
        from clean_pvnet_amd import dist as pdist
        from clean_pvnet_amd import synth
        from oracle import vote_oracle
        cfg = {**synth.CONFIGS["cfg1"]}
        cfg.pop("B")
        lo, hi = pdist.shard_bounds(batch, world, rank)
        d = synth.make_batch(B=hi - lo, **cfg, first_index=lo) if hi > lo else None     # each rank makes ITS shard only

        def vote(mask, vertex, hn, **kw):
            tn = [int(x) for x in (mask != 0).sum((1, 2))]
            idxs = synth.make_idxs(tn, hn, vertex.shape[3], first_index=lo).numpy()
            return torch.from_numpy(vote_oracle.ransac_voting_layer_v3(mask.numpy(), vertex.numpy(), hn, 0.99, idxs=idxs))

        # a rank without images (batch < world, or the tail of an uneven split) makes the same calls with zero rows:
        # sharded_vote itself must enter the collective -- ADVICE r1: it used to call the layer on an empty batch and
        # hang the others
        m = d["mask"] if d is not None else torch.zeros(0, cfg["H"], cfg["W"], dtype=torch.int64)
        v = d["vertex"] if d is not None else torch.zeros(0, cfg["H"], cfg["W"], cfg["K"], 2)
        out = pdist.sharded_vote(vote, m, v, batch, cfg["hn"])
        # seed= : every rank votes with the common key and the index of ITS first image (device-RNG results are
        # then independent of the sharding)
        seen = {}

        def vote_kw(mask, vertex, hn, **kw):
            seen.update(kw)
            return vote(mask, vertex, hn)
        out_s = pdist.sharded_vote(vote_kw, m, v, batch, cfg["hn"], seed=777)
        assert torch.equal(out_s, out) and (d is None or seen == {"seed": 777, "first_image": lo})
        cov_local = torch.arange((hi - lo) * cfg["K"] * 4, dtype=torch.float32).view(hi - lo, cfg["K"], 2, 2) + 1000 * rank
        cov = pdist.gather_results(cov_local, batch)
        # async form (what bench.py uses to overlap the exchange with the next batch's voting)
        cov2, work = pdist.gather_results(cov_local, batch, async_op=True)
        work.wait()
        assert torch.equal(cov2, cov)
        # GatherBuffer (round 5): the persistent buffer the voting call writes into (out=buf.mine), ONE in-place collective -- the
        # same rows in the same order, uneven and empty shards included, and reusable step after step
        buf = pdist.GatherBuffer(batch, (cfg["K"], 2, 2), "cpu")
        assert buf.mine.shape[0] == hi - lo and (buf.lo, buf.hi) == (lo, hi)
        for rep in range(2):
            buf.mine.copy_(cov_local + rep)
            got = buf.gather()
            assert got.shape == cov.shape and got.data_ptr() == buf.full.data_ptr()
            # (rows of rank r carry 1000 r + rep: compare with the padded-gather result shifted by rep)
            assert torch.equal(got, cov + rep), rep
        g2, w2 = buf.gather(async_op=True)
        w2.wait()
        assert torch.equal(g2, cov + 1)
        # out= through the layer: the vote function receives the buffer's rows and fills them
        kbuf = pdist.GatherBuffer(batch, (cfg["K"], 2), "cpu")
        if d is not None:
            kbuf.mine.copy_(vote(m, v, cfg["hn"]))
        assert torch.equal(kbuf.gather(), out)
        q.put((rank, out.numpy(), cov.numpy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("batch,world", [(4, 2), (5, 2), (2, 3)])
def test_gloo_sharded_vote_equals_single_process(oracle, synth, batch, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = {**synth.CONFIGS["cfg1"]}
    cfg.pop("B")
    d = synth.make_batch(B=batch, **cfg)                                            # the same images, unsharded
    tn = [int(x) for x in (d["mask"] != 0).sum((1, 2))]
    idxs = synth.make_idxs(tn, cfg["hn"], cfg["K"]).numpy()
    want = oracle.ransac_voting_layer_v3(d["mask"].numpy(), d["vertex"].numpy(), cfg["hn"], 0.99, idxs=idxs)
    for rank, out, cov in got:
        assert out.shape == (batch, cfg["K"], 2)
        np.testing.assert_array_equal(out, want)                                    # every rank holds the full result
        per = -(-batch // world)
        assert cov.shape == (batch, cfg["K"], 2, 2) and cov[0, 0, 0, 0] == 0 and (per >= batch or cov[per, 0, 0, 0] == 1000)
