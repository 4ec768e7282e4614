"""CPU, world_size 2, gloo: the N>1 path -- text sharding and the gatherv of hit records to rank 0."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, counts, q):
    sys.path.insert(0, ROOT)
    from libspm_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = counts[rank]
        local = torch.empty((n, 2), dtype=torch.int64)
        local[:, 0] = torch.arange(n) + 1000 * rank  # pos
        local[:, 1] = rank                            # pattern|score word
        out = sdist.gatherv_hits(local, dst=0)
        # the one-collective variant used by bench.py must deliver the same records in the same order
        cap = 16
        buf = torch.zeros((cap + 1, 2), dtype=torch.int64)
        buf[0, 0] = n
        buf[1:1 + n] = local
        fused = sdist.split_fused(sdist.gather_hits_fused(buf))
        # 24-byte journaled-sequence hit records (three int64 words) go through the same gatherv
        wide = torch.cat([local, (local[:, :1] * 3 + 1)], dim=1)
        out3 = sdist.gatherv_hits(wide, dst=0)
        # the double-buffered variant bench.py uses for C5 (on CPU tensors it degenerates to the plain gatherv)
        og = sdist.OverlappedGather(torch.device("cpu"), 16, 2)
        outs = []
        for i in range(3):
            buf = og.buffer(i)
            buf[:n] = local + i
            outs.append(og.submit(i, n))
        og.finish()
        if rank == 0:
            for i in range(3):
                assert torch.equal(outs[i], out + i)
        if rank == 0:
            assert torch.equal(fused, out)
            assert out3.shape[1] == 3 and torch.equal(out3[:, :2], out) and torch.equal(out3[:, 2], out[:, 0] * 3 + 1)
            q.put(out.numpy().copy())
        else:
            assert out is None and out3 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("counts", [(3, 5), (0, 4), (7, 0), (0, 0)])
def test_gatherv_hits_world2(counts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + sum(counts)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, counts, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    want = np.concatenate([np.stack([np.arange(c) + 1000 * r, np.full(c, r)], 1) for r, c in enumerate(counts)]
                          ).reshape(-1, 2)
    assert np.array_equal(out, want)


def test_shard_ranges_cover_text_once():
    sys.path.insert(0, ROOT)
    from libspm_amd import dist as sdist
    for n_total in (1 << 20, (1 << 34) * 3 + 4096, 1000):
        for world in (1, 2, 3, 8):
            prev = 0
            for r in range(world):
                lo, hi = sdist.shard_range(n_total, r, world)
                assert lo == prev and lo % 1024 == 0 or lo == n_total
                prev = hi
            assert prev == n_total


def _plan_worker(rank, world, port, n_total, window, n_needles, many, overflow_rank, q):
    """What bench.py's ranks do around the scan, with synthetic hits: rank r "finds" an occurrence ending at every global
    position e with e % 997 == 13 whose last symbol it owns -- in LOCAL coordinates of its resident text, then made global
    with the plan's pos_offset, exactly as spm_hip_scan does -- and the records travel as the exchange plan says."""
    sys.path.insert(0, ROOT)
    from libspm_amd import dist as sdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = sdist.ShardPlan(n_total, rank, world, window)
        xp = sdist.ExchangePlan(n_needles, many_hits=many)
        assert sp.text_begin % 1024 == 0 and sp.scan_end - sp.scan_begin == sp.hi - sp.lo
        assert sp.ovl == 0 if rank == 0 else sp.ovl >= window - 1
        local_ends = np.arange(sp.scan_begin + 1, sp.scan_end + 1, dtype=np.int64)        # exclusive ends, local coordinates
        glob = local_ends + sp.pos_offset
        mine = glob[glob % 997 == 13]
        assert all(sp.owns_end(int(e)) for e in mine[:3]) and all(sp.owns_end(int(e)) for e in mine[-3:])
        n = len(mine)
        if rank == overflow_rank:
            n = xp.cap + 5                       # more hits than the buffer holds: only `cap` of them are in it
            mine = np.resize(mine, xp.cap)
        buf = xp.new_buffer(torch.device("cpu"))
        buf[0, 0] = n
        buf[1:1 + min(n, xp.cap), 0] = torch.from_numpy(mine[:xp.cap])
        buf[1:1 + min(n, xp.cap), 1] = rank
        try:
            g = xp.exchange(buf, n)
            if rank == 0:
                rec = xp.records(g)
                q.put(("ok", rec.numpy().copy(), xp.fused))
        except OverflowError as e:
            assert rank == 0
            q.put(("overflow", str(e), xp.fused))
        dist.barrier()                           # every rank got here: nobody was left in a collective
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_needles,many,overflow_rank",
                         [(2, 1024, False, None), (3, 1024, False, None), (8, 64, False, None),      # fused all-gather
                          (3, 5000, False, None), (2, 100000, False, None),                          # count + send/recv
                          (2, 64, False, 1), (3, 1024, False, 0), (3, 3000, False, 2), (2, 3000, False, 0)])    # a rank over `cap`
def test_rank_side_plans_tile_the_text_and_fail_together(world, n_needles, many, overflow_rank):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_total, window = 3_000_000 + 1024 * 7, 153
    port = 31000 + (os.getpid() % 2000) + world * 11 + n_needles % 7 + (overflow_rank or 0)
    procs = [ctx.Process(target=_plan_worker, args=(r, world, port, n_total, window, n_needles, many, overflow_rank, q))
             for r in range(world)]
    for p in procs:
        p.start()
    kind, payload, fused = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from libspm_amd import dist as sdist
    assert fused == sdist.ExchangePlan(n_needles, many).fused == (8 * n_needles <= 8192)
    if overflow_rank is not None:
        assert kind == "overflow"
        return
    assert kind == "ok"
    ends = np.arange(1, n_total + 1, dtype=np.int64)
    want = ends[ends % 997 == 13]
    assert np.array_equal(payload[:, 0], want)                      # every position once, ascending = shard order
    owner = np.minimum(((want - 1) // sdist.shard_range(n_total, 0, world)[1]), world - 1)
    assert np.array_equal(payload[:, 1], owner)


def test_shard_plans():
    sys.path.insert(0, ROOT)
    from libspm_amd import dist as sdist
    for n_total in (1 << 20, (1 << 34) * 3 + 4096, 5000):
        for world in (1, 2, 3, 8):
            for window in (32, 103, 153, 1088, 5000):
                prev = 0
                for r in range(world):
                    sp = sdist.ShardPlan(n_total, r, world, window)
                    assert sp.lo == prev and sp.pos_offset == sp.text_begin == sp.lo - sp.ovl
                    assert sp.text_len == sp.hi - sp.text_begin and sp.scan_begin == sp.ovl
                    if sp.hi > sp.lo:                               # (a rank beyond the end of a tiny text holds nothing)
                        assert sp.text_begin % 1024 == 0
                    if sp.lo > 0 and sp.hi > sp.lo:
                        assert sp.ovl >= min(window - 1, sp.lo)     # enough left context (all there is, at the very front)
                    prev = sp.hi
                assert prev == n_total
