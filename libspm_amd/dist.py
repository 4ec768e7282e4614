"""Multi-GPU driver pieces: text sharding and the gatherv of hit records.

One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI on ROCm ("gloo" in the CPU tests).  The
path shards by text position: rank g owns the hits whose last symbol lies in its range and reads window_size-1
symbols of left context, so no data-path collective is needed during the scan.  The only exchange is one gatherv of
16-byte hit records to rank 0 at the end: an all_gather of the per-rank counts, then grouped send/recv
(ncclGroupStart .. ncclSend/ncclRecv .. ncclGroupEnd under torch's batch_isend_irecv) -- RCCL has no native gatherv.
Bytes per step and rank: C3 ~131 KB (one fused all-gather of fixed-size buffers), C4 ~1 MB, C5 ~26 MB
(1.08 M haplotype-coordinate records of 24 bytes; sent while the next search runs, OverlappedGather).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_total: int, rank: int, world: int, align: int = 1024):
    """Contiguous shard [lo, hi) of rank `rank`; boundaries aligned so device loads stay 16-byte aligned."""
    per = (n_total + world - 1) // world
    per = (per + align - 1) // align * align
    lo = min(n_total, rank * per)
    hi = min(n_total, lo + per)
    return lo, hi


def gatherv_hits(local: torch.Tensor, dst: int = 0, group=None):
    """local: int64 tensor [n, w] (hit records viewed as int64 words: w = 2 for the 16-byte spm_hit, 3 for the 24-byte
    spm_jst_hit).  Returns on `dst` the concatenation of every rank's records in rank order (= ascending shard
    order), elsewhere None.

    One count all-gather (ncclAllGather of one int64 per rank) + grouped send/recv.  Host synchronisation: ONE device-to-
    host copy of the gathered counts, on the root only (it has to size its receives); the other ranks know their own
    count already and never wait for the device."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # rehearsal path (gloo cannot send device tensors): stage through the host, same protocol
        out = gatherv_hits(local.cpu(), dst, group)
        return out.to(local.device) if out is not None else None
    n_mine = int(local.shape[0])
    n_local = torch.tensor([n_mine], dtype=torch.int64, device=local.device)
    counts_t = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts_t, n_local, group=group)
    local = local.contiguous()
    if rank == dst:
        counts = counts_t.cpu().tolist()  # the one host synchronisation of the exchange
        out = torch.empty((sum(counts), local.shape[1]), dtype=torch.int64, device=local.device)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        ops = []
        for r in range(world):
            if r == dst:
                out[offs[r]:offs[r + 1]].copy_(local)
            elif counts[r] > 0:
                ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return out
    if n_mine > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
            req.wait()
    return None


class OverlappedGather:
    """gatherv_hits on a side stream, double-buffered: the records of search i travel to the root while search i + 1
    runs (C5: ~26 MB per rank and search -- 1.08 M records of 24 bytes -- against 1.2 ms of compute).

        og = OverlappedGather(device, cap, words)
        buf = og.buffer(i)                  # device tensor [cap, words] to fill (waits until its last gather is done)
        ... fill buf[:n] on the current stream ...
        out = og.submit(i, n)               # root: the gathered records (ready once og.finish() / the stream order says so)
        og.finish()                         # before reading results / stopping the clock
    """

    def __init__(self, device, cap: int, words: int, dst: int = 0, group=None):
        self.bufs = [torch.zeros((cap, words), dtype=torch.int64, device=device) for _ in range(2)]
        self.done = [None, None]
        self.dst, self.group = dst, group
        self.single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
        self.comm = None if self.single or device.type != "cuda" else torch.cuda.Stream(device=device)

    def buffer(self, i: int) -> torch.Tensor:
        ev = self.done[i & 1]
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.done[i & 1] = None
        return self.bufs[i & 1]

    def submit(self, i: int, n: int):
        buf = self.bufs[i & 1][:n]
        if self.single:
            return buf
        if self.comm is None:
            return gatherv_hits(buf, self.dst, self.group)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            out = gatherv_hits(buf, self.dst, self.group)
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self.done[i & 1] = ev
        return out

    def finish(self):
        if self.comm is not None:
            self.comm.synchronize()


def gather_hits_fused(buf: torch.Tensor, group=None):
    """One-collective variant for the benchmark loop.  `buf` is an int64 tensor [cap + 1, 2]: row 0 holds the hit
    count in column 0, rows 1..count the 16-byte records.  Every rank contributes the same fixed-size buffer to ONE
    all-gather (ncclAllGather on RCCL); no count exchange, no host synchronisation on non-root ranks.  Returns the
    [world, cap + 1, 2] tensor; use split_fused() on the root.  Counts above `cap` are detected there."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf.unsqueeze(0)
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(buf.cpu()) for _ in range(world)]
        dist.all_gather(parts, buf.cpu().contiguous(), group=group)
        return torch.stack(parts).to(buf.device)
    out = torch.empty((world,) + tuple(buf.shape), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    return out


def split_fused(gathered: torch.Tensor) -> torch.Tensor:
    """Root side of gather_hits_fused: concatenation of every rank's records in rank (= shard) order."""
    cap = gathered.shape[1] - 1
    counts = gathered[:, 0, 0].cpu().tolist()
    if any(c > cap for c in counts):
        raise OverflowError(f"a rank produced {max(counts)} hits but the fused gather buffer holds {cap}")
    return torch.cat([gathered[r, 1:1 + int(c)] for r, c in enumerate(counts)]) if counts else gathered[0, 1:1]
