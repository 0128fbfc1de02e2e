#!/usr/bin/env python
"""In-kernel phase timestamps of the count kernel (tuning build: tools/build_variant.sh <name> -DPVV_TUNING, run with
PVV_LIBPATH=build/variants/<name>.so): wall_clock64() (100 MHz) of three blocks at ten points, relative to block 0's
entry, median over calls.   usage: phase_stamps.py <config_bench row> [calls]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import capi  # noqa: E402
import config_bench  # noqa: E402
import variant_time  # noqa: E402

NAMES = ["entry", "htpi", "table", "item", "pix+C1", "A built", "B staged", "loop done", "flushed", "exit"]


def main():
    rowname = sys.argv[1]
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    synth = variant_time._synth()
    name, cfgname, B, over = [r for r in config_bench.ROWS if r[0] == rowname][0]
    cfg = dict(synth.CONFIGS[cfgname])
    hn, max_num = over.get("hn", cfg["hn"]), over.get("max_num", 30000)
    dev = torch.device("cuda:0")
    d = synth.make_batch(B=B, **{k: v for k, v in cfg.items() if k not in ("B", "hn")}, device=dev)
    GRID = 48 * 256 + 64
    dbg = torch.zeros(64 + 16 * GRID, dtype=torch.int64, device=dev)
    os.environ["PVV_DBG_PTR"] = str(dbg.data_ptr())
    rows = []
    for _ in range(calls):
        dbg.zero_()
        capi.v3(d["mask"], d["vertex"], hn, 0.99, max_num=max_num, seed=5)
        torch.cuda.synchronize()
        rows.append(dbg.cpu().clone())
    census = rows[-1][64:].view(-1, 16)[:, :4]     # (entry, exit, hardware id, items; the phase cycles behind them: tools/census_count.py)
    rows = torch.stack(rows[5:])[:, :48].reshape(-1, 3, 16)[:, :, :10].double()
    rel = (rows - rows[:, :1, :1]) / 100.0          # us relative to block 0's entry
    med = rel.median(0).values
    print("row %s: phase timestamps (us after block 0 entered), blocks 0 / 97 / 401" % rowname)
    for i, n in enumerate(NAMES):
        print("  %-10s %8.2f %8.2f %8.2f" % (n, med[0, i], med[1, i], med[2, i]))
    # census of the last call: when blocks ran, where, and how many items each took
    c = census[census[:, 0] != 0]
    t0 = int(c[:, 0].min())
    ent, ext = (c[:, 0] - t0).double() / 100.0, (c[:, 1] - t0).double() / 100.0
    hw, items = c[:, 2], c[:, 3]
    xcc = (hw >> 32) & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5)     # cu_id | sh_id | se_id
    key = xcc * 1024 + cu
    work = items > 0
    print("census: %d blocks ran, %d with items (%d items); entry min/median/max %.2f/%.2f/%.2f us; exit max %.2f us" % (
        len(c), int(work.sum()), int(items.sum()), ent.min(), ent.median(), ent.max(), ext.max()))
    print("   working blocks: entry max %.2f, exit median %.2f max %.2f; life median %.2f max %.2f" % (
        ent[work].max(), ext[work].median(), ext[work].max(), (ext - ent)[work].median(), (ext - ent)[work].max()))
    print("   idle blocks: life median %.2f us" % ((ext - ent)[~work].median() if (~work).any() else 0.0))
    uk, cnt = torch.unique(key[work], return_counts=True)
    hist = torch.bincount(cnt)
    print("   distinct CUs with working blocks: %d; working blocks per CU histogram: %s" % (len(uk), hist.tolist()))
    print("   working blocks per XCC: %s" % torch.bincount(xcc[work], minlength=8).tolist())
    # concurrency: working blocks alive at a few instants
    for t in (2.0, 5.0, 10.0, 20.0, 30.0):
        alive = ((ent <= t) & (ext > t) & work).sum().item()
        print("   t=%5.1f us: %4d working blocks alive" % (t, alive))


if __name__ == "__main__":
    main()
