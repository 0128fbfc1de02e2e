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


# ---------------------------------------------------------------------------------------------------------------------
# Round 6 (VERDICT r5 #2b, weak #8): rccl.Comm.create under injected faults.  The stand-in library below behaves like RCCL's
# bootstrap where it matters: ncclCommInitRank is COLLECTIVE AND BLOCKING -- it returns on a rank only once every rank of the
# world has entered it (a directory of marker files stands in for the bootstrap sockets) -- so a rank that fails before or
# instead of entering leaves the others parked inside it for ever.  The promise under test: every rank gets None, every rank
# has Comm.last_error set, nobody hangs (the watchdog abandons the parked call), and the process group is still usable.
# ---------------------------------------------------------------------------------------------------------------------
class _StandInRccl:
    def __init__(self, rendezvous_dir, world):
        self.dir, self.world, self.destroyed = rendezvous_dir, world, 0

    def ncclGetUniqueId(self, ref):
        for i in range(128):
            ref._obj.internal[i] = (i * 7 + 3) % 127
        return 0

    def ncclGetErrorString(self, rc):
        return b"stand-in error %d" % rc

    def ncclCommInitRank(self, handle_ref, world, uid, rank):
        import time
        assert bytes(uid.internal)[:4] == bytes([(i * 7 + 3) % 127 for i in range(4)]), "the id did not arrive"
        open(os.path.join(self.dir, "entered_%d" % rank), "w").close()
        while len([f for f in os.listdir(self.dir) if f.startswith("entered_")]) < world:
            time.sleep(0.02)                                    # parked, like a rank waiting in RCCL's bootstrap
        handle_ref._obj.value = 0x1000 + rank
        return 0

    def ncclCommDestroy(self, handle):
        self.destroyed += 1
        return 0


def _fault_worker(rank, world, port, fault, rdv, q):
    import time
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if fault:
        os.environ["PVV_RCCL_FAULT"] = fault
    else:
        os.environ.pop("PVV_RCCL_FAULT", None)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import lib
        lib._register_clean_pvnet_amd()
        from clean_pvnet_amd import rccl
        fake = _StandInRccl(rdv, world)
        t0 = time.time()
        comm = rccl.Comm.create("cpu", init_timeout_s=3.0, _lib=fake)
        took = time.time() - t0
        # the group must still work afterwards (nobody is stuck in a half-finished collective)
        x = torch.tensor([rank + 1.0])
        dist.all_reduce(x)
        entered = sorted(f for f in os.listdir(rdv) if f.startswith("entered_"))
        q.put((rank, comm is not None, rccl.Comm.last_error, took, float(x.item()), entered, fake.destroyed))
        if comm is not None:
            comm.destroy()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fault,expect_comm,max_s", [
    (2, None, True, 2.5),                 # no fault: every rank gets a communicator, at once
    (2, "1:load", False, 2.5),            # the library is missing on ONE rank: agreed before anybody draws / receives an id
    (2, "1:before_init", False, 2.5),     # one rank fails after the id broadcast: agreed BEFORE anybody enters ncclCommInitRank
    (2, "1:inside_init", False, 8.0),     # one rank dies inside: its peer is parked in the collective until the watchdog (3 s) lets go
    (3, "0:inside_init", False, 8.0),     # ... rank 0 this time, two ranks parked
])
def test_rccl_comm_create_every_rank_falls_back_together_and_nobody_hangs(tmp_path, world, fault, expect_comm, max_s):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    rdv = str(tmp_path)
    procs = [ctx.Process(target=_fault_worker, args=(r, world, port, fault, rdv, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = world * (world + 1) / 2.0
    for rank, has_comm, err, took, allsum, entered, destroyed in got:
        assert has_comm == expect_comm, (rank, err)
        assert allsum == total                                               # the process group survived
        assert took < max_s, "rank %d spent %.1f s in Comm.create" % (rank, took)
        if not expect_comm:
            assert err, "rank %d returned None without saying why" % rank     # last_error on every path
    if fault and fault.endswith(("load", "before_init")):
        assert all(e == [] for *_x, e, _d in got), "a rank entered ncclCommInitRank although a peer had already failed"
    if fault and fault.endswith("inside_init"):
        bad = int(fault.split(":")[0])
        assert all(("entered_%d" % bad) not in e for *_x, e, _d in got)
        assert any("did not return within" in err for r, _c, err, *_ in got if r != bad)      # the parked ranks say so
        assert any("injected failure inside" in err for r, _c, err, *_ in got if r == bad)
