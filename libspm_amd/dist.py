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


class ShardPlan:
    """What one rank scans and how its hit positions become global.

    The global text [0, n_total) is cut into `world` contiguous shards, 1 KiB aligned.  Rank r OWNS the hits whose last
    symbol lies in [lo, hi); to find those that begin before `lo` it also holds `ovl` >= window_size - 1 symbols of left
    context (none on rank 0), so its resident text is global [text_begin, hi) and it scans local [scan_begin, scan_end)
    with left_context = True and pos_offset = text_begin: every hit is reported once, by its owner, in global coordinates.
    """

    __slots__ = ("rank", "world", "n_total", "lo", "hi", "ovl", "text_begin", "text_len", "scan_begin", "scan_end",
                 "pos_offset")

    def __init__(self, n_total: int, rank: int, world: int, window: int, align: int = 1024):
        self.rank, self.world, self.n_total = rank, world, n_total
        self.lo, self.hi = shard_range(n_total, rank, world, align)
        need = max(window - 1, 0)
        self.ovl = 0 if self.lo == 0 else min(self.lo, (need + align - 1) // align * align)  # keeps the shard aligned
        self.text_begin = self.lo - self.ovl
        self.text_len = (self.hi - self.lo) + self.ovl
        self.scan_begin = self.ovl
        self.scan_end = self.ovl + (self.hi - self.lo)
        self.pos_offset = self.text_begin

    def owns_end(self, end_exclusive: int) -> bool:
        """Myers reports the exclusive end e of an occurrence: its last symbol e - 1 decides the owner."""
        return self.lo <= end_exclusive - 1 < self.hi


class ExchangePlan:
    """How the hit records of one step travel to rank 0.

    Few records (<= 8 per needle fit 8 192 slots: C2, C3): ONE all-gather of fixed-size [count | records] buffers
    (gather_hits_fused), no count exchange, no host synchronisation off the root.  Many (C4: ~51 000 per rank; a repeat-rich
    text: millions): count all-gather + grouped send/recv (gatherv_hits) moves what there is, not the capacity."""

    __slots__ = ("cap", "fused", "max_hits")

    def __init__(self, n_needles: int, many_hits: bool = False):
        self.max_hits = max(1 << 20, 16 * n_needles)
        cap = 1 << 12   # small on purpose: the fused all-gather moves world x (cap + 1) x 16 bytes per step
        while cap < 8 * n_needles:
            cap <<= 1
        if many_hits:   # a needle inside a repeat stretch matches every long stretch of the same unit
            self.max_hits = 1 << 26
            cap = self.max_hits
        self.cap = cap
        self.fused = cap <= (1 << 13)

    def new_buffer(self, device) -> torch.Tensor:
        """[cap + 1, 2] int64: row 0 = [count, 0], rows 1.. = 16-byte records; the same fixed size on every rank."""
        return torch.zeros((self.cap + 1, 2), dtype=torch.int64, device=device)

    def exchange(self, hit_buf: torch.Tensor, n: int, group=None):
        """hit_buf as new_buffer() made it, holding min(n, cap) records (and, fused, the count n in row 0).  Returns what
        the root needs for records(): the fused [world, cap + 1, 2] tensor, or the concatenated records.  A rank with more
        hits than `cap` does not leave the collective: the root raises OverflowError once everybody's records are in
        (fused: split_fused reads the counts; otherwise the rank announces -1 instead of a count and sends nothing)."""
        if self.fused:
            return gather_hits_fused(hit_buf, group)
        return gatherv_hits(hit_buf[1:1 + min(n, self.cap)], 0, group, overflow=n > self.cap)

    def records(self, gathered) -> torch.Tensor:
        """Root: every rank's records in rank (= shard) order, [total, 2] int64."""
        return split_fused(gathered) if self.fused else gathered


def gatherv_hits(local: torch.Tensor, dst: int = 0, group=None, overflow: bool = False):
    """local: int64 tensor [n, w] (hit records viewed as int64 words: w = 2 for the 16-byte spm_hit, 3 for the 24-byte
    spm_jst_hit).  Returns on `dst` the concatenation of every rank's records in rank order (= ascending shard
    order), elsewhere None.

    One count all-gather (ncclAllGather of one int64 per rank) + grouped send/recv.  Host synchronisation: ONE device-to-
    host copy of the gathered counts, on the root only (it has to size its receives); the other ranks know their own
    count already and never wait for the device.

    overflow: this rank's records are incomplete (its buffer was too small).  It announces -1 instead of a count and sends
    nothing; the root receives everybody else's records and then raises OverflowError -- no rank is left waiting."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if overflow:
            raise OverflowError("this rank produced more hits than its gather buffer holds")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and local.is_cuda:
        # rehearsal path (gloo cannot send device tensors): stage through the host, same protocol
        out = gatherv_hits(local.cpu(), dst, group, overflow)
        return out.to(local.device) if out is not None else None
    n_mine = 0 if overflow else int(local.shape[0])
    n_local = torch.tensor([-1 if overflow else n_mine], dtype=torch.int64, device=local.device)
    counts_t = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(counts_t, n_local, group=group)
    local = local.contiguous()
    if rank == dst:
        counts = counts_t.cpu().tolist()  # the one host synchronisation of the exchange
        over = [r for r, c in enumerate(counts) if c < 0]
        counts = [max(c, 0) for c in counts]
        out = torch.empty((sum(counts), local.shape[1]), dtype=torch.int64, device=local.device)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        ops = []
        for r in range(world):
            if r == dst:
                out[offs[r]:offs[r + 1]].copy_(local[:counts[r]])
            elif counts[r] > 0:
                ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], r, group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if over:
            raise OverflowError(f"rank(s) {over} produced more hits than their gather buffers hold")
        return out
    if n_mine > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
            req.wait()
    return None


class OverlappedGather:
    """gatherv_hits on a side stream, double-buffered: the records of search i travel to the root while search i + 1
    runs (C5: ~26 MB per rank and search -- 1.08 M records of 24 bytes -- against 1 ms of compute).  What overlaps is the
    record transfer: the root sizes its receives from the gathered counts, so ITS submit(i) returns only once search i and
    the count all-gather are done (one device-to-host copy of `world` counts); the other ranks' submit never waits.
    `out` is allocated on the side stream: it is recorded for the caller's stream, and must not be read before finish()
    (or an event wait on the side stream).

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
        if out is not None and out.is_cuda:
            out.record_stream(torch.cuda.current_stream())  # (allocated under the side stream, used by the caller's)
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
    head = gathered[:, 0, :].cpu()
    counts = head[:, 0].tolist()
    if any(int(s) != 0 for s in head[:, 1].tolist()):
        # (spm_hip_hits_copy_fused_device: the header's second word says the scan needs its host -- a list overflowed)
        raise RuntimeError("a rank's fused buffer holds an unfinished scan: complete it (Hits.view / stats) and copy again")
    if any(c > cap for c in counts):
        raise OverflowError(f"a rank produced {max(counts)} hits but the fused gather buffer holds {cap}")
    return torch.cat([gathered[r, 1:1 + int(c)] for r, c in enumerate(counts)]) if counts else gathered[0, 1:1]
