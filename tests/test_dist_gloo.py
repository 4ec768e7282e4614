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
