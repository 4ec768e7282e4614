"""Thin object layer over the C ABI, used by tests, bench.py and the multi-GPU driver.

Names follow the reference's domain: a *haystack* (Text) is scanned for a set of *needles* (PatternSet); a scan
returns hit records (pos, pattern, score) -- what the reference's callback reads off the seqan2::Finder
(/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-52).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

HIT_DTYPE = np.dtype([("pos", "<u8"), ("pattern", "<u4"), ("score", "<i4")])


def _check(rc, ctx_handle):
    if rc != 0:
        msg = capi.lib().spm_hip_last_error(ctx_handle)
        raise capi.SpmError(f"libspm_hip error {rc}: {msg.decode() if msg else ''}")


class Context:
    def __init__(self, device: int = 0, stream: int | None = None):
        self._h = C.c_void_p()
        rc = capi.lib().spm_hip_init(device, C.c_void_p(stream) if stream else None, C.byref(self._h))
        if rc != 0:
            msg = capi.lib().spm_hip_last_error(None)
            raise capi.SpmError(f"spm_hip_init failed ({rc}): {msg.decode() if msg else ''}")
        self.device = device

    def synchronize(self):
        _check(capi.lib().spm_hip_synchronize(self._h), self._h)

    def close(self):
        if self._h:
            capi.lib().spm_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- haystacks ----
    def upload(self, ranks, sigma: int = 4) -> "Text":
        a = np.ascontiguousarray(ranks, dtype=np.uint8)
        h = C.c_void_p()
        _check(capi.lib().spm_hip_text_upload(self._h, a.ctypes.data_as(C.POINTER(C.c_uint8)), a.size, sigma,
                                              C.byref(h)), self._h)
        return Text(self, h, sigma)

    def wrap(self, device_ptr: int, n: int, sigma: int = 4, keepalive=None) -> "Text":
        h = C.c_void_p()
        _check(capi.lib().spm_hip_text_wrap(self._h, C.c_void_p(device_ptr), n, sigma, C.byref(h)), self._h)
        t = Text(self, h, sigma)
        t._keepalive = keepalive
        return t

    def generate(self, seed: int, global_begin: int, n: int) -> "Text":
        h = C.c_void_p()
        _check(capi.lib().spm_hip_text_generate(self._h, seed, global_begin, n, C.byref(h)), self._h)
        return Text(self, h, 4)

    def generate_repeats(self, seed: int, global_begin: int, n: int, repeat_ppm: int) -> "Text":
        """Synthetic repeat-rich dna4 text (bench workload c3r; csrc/synth.hpp)."""
        h = C.c_void_p()
        _check(capi.lib().spm_hip_text_generate_repeats(self._h, seed, global_begin, n, repeat_ppm, C.byref(h)),
               self._h)
        return Text(self, h, 4)

    # ---- needles ----
    def patterns(self, algo: int, needles, k=0, sigma: int = 4) -> "PatternSet":
        """needles: a sequence of rank arrays, or -- reads of one length -- a 2-D uint8 array, one needle per row (what the C ABI
        takes anyway: ranks back to back + offsets; 100 000 rows cost a reshape instead of 100 000 Python objects)."""
        if isinstance(needles, np.ndarray) and needles.ndim == 2:
            mat = np.ascontiguousarray(needles, dtype=np.uint8)
            n_needles = mat.shape[0]
            offs = (np.arange(n_needles + 1, dtype=np.uint64) * mat.shape[1]).astype(np.uint32)
            cat = mat.reshape(-1) if mat.size else np.zeros(1, np.uint8)
            needles = mat          # (len() below)
        else:
            needles = [np.ascontiguousarray(p, dtype=np.uint8) for p in needles]
            offs = np.zeros(len(needles) + 1, dtype=np.uint32)
            if needles:
                offs[1:] = np.cumsum([len(p) for p in needles])
            cat = np.concatenate(needles) if needles and offs[-1] else np.zeros(1, np.uint8)
        if np.isscalar(k):
            ks = np.full(max(1, len(needles)), k, dtype=np.uint16)
        else:
            ks = np.ascontiguousarray(k, dtype=np.uint16)
        h = C.c_void_p()
        _check(capi.lib().spm_hip_patterns_create(self._h, algo, cat.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                  offs.ctypes.data_as(C.POINTER(C.c_uint32)), len(needles),
                                                  ks.ctypes.data_as(C.POINTER(C.c_uint16)), sigma, C.byref(h)),
               self._h)
        return PatternSet(self, h, algo, len(needles))


class Text:
    def __init__(self, ctx, h, sigma):
        self.ctx, self._h, self.sigma = ctx, h, sigma
        self._keepalive = None

    def __len__(self):
        return int(capi.lib().spm_hip_text_length(self._h))

    @property
    def device_ptr(self) -> int:
        return int(capi.lib().spm_hip_text_device_ptr(self._h) or 0)

    def pack(self):
        """Build the optional 2-bit shadow (dna4 only): later seed-filter scans stream a quarter of the bytes."""
        _check(capi.lib().spm_hip_text_pack(self.ctx._h, self._h), self.ctx._h)
        return self

    @property
    def packed(self) -> bool:
        return bool(capi.lib().spm_hip_text_is_packed(self._h))

    def download(self, begin: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint8)
        _check(capi.lib().spm_hip_text_download(self.ctx._h, self._h, begin, n,
                                                out.ctypes.data_as(C.POINTER(C.c_uint8))), self.ctx._h)
        return out

    def close(self):
        if self._h:
            if self.ctx._h:  # the C objects point at their context: once it is gone there is nothing left to release
                capi.lib().spm_hip_text_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PatternSet:
    def __init__(self, ctx, h, algo, n):
        self.ctx, self._h, self.algo, self.n = ctx, h, algo, n

    def window_size(self, p: int = 0) -> int:
        return int(capi.lib().spm_hip_patterns_window_size(self._h, p))

    @property
    def filterable(self) -> bool:
        return bool(capi.lib().spm_hip_patterns_filterable(self._h))

    def build_stats(self) -> capi.BuildStats:
        """What spm_hip_patterns_create spent where (host ms) and what it built (passes, keys, dense, anchors)."""
        st = capi.BuildStats()
        _check(capi.lib().spm_hip_patterns_build_stats(self._h, C.byref(st)), self.ctx._h)
        return st

    def state_stride(self) -> int:
        return int(capi.lib().spm_hip_patterns_state_stride(self._h))

    def initial_state(self) -> np.ndarray:
        st = np.zeros(self.state_stride() * max(1, self.n), dtype=np.uint8)
        _check(capi.lib().spm_hip_patterns_state_init(self._h, st.ctypes.data), self.ctx._h)
        return st

    def close(self):
        if self._h:
            if self.ctx._h:  # the C objects point at their context: once it is gone there is nothing left to release
                capi.lib().spm_hip_patterns_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hits:
    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    def view(self) -> np.ndarray:
        """Hits sorted by (pattern, pos): per pattern, the order the reference's callback fires in."""
        rec = C.POINTER(capi.Hit)()
        n = C.c_uint64()
        _check(capi.lib().spm_hip_hits_view(self._h, C.byref(rec), C.byref(n)), self.ctx._h)
        if n.value == 0:
            return np.zeros(0, dtype=HIT_DTYPE)
        buf = (capi.Hit * n.value).from_address(C.addressof(rec.contents))
        return np.frombuffer(buf, dtype=HIT_DTYPE).copy()

    def device(self):
        p = C.c_void_p()
        n = C.c_uint64()
        _check(capi.lib().spm_hip_hits_device(self._h, C.byref(p), C.byref(n)), self.ctx._h)
        return int(p.value or 0), int(n.value)

    def copy_to(self, device_ptr: int, cap: int) -> int:
        """Async D2D copy of the hit records into a caller-owned device buffer; returns the hit count."""
        n = C.c_uint64()
        _check(capi.lib().spm_hip_hits_copy_device(self._h, C.c_void_p(device_ptr), cap, C.byref(n)), self.ctx._h)
        return int(n.value)

    def copy_fused(self, device_ptr: int, cap: int) -> int:
        """copy_to with a 16-byte {count, 0} header in front: the [count | records] buffer of dist.gather_hits_fused."""
        n = C.c_uint64()
        _check(capi.lib().spm_hip_hits_copy_fused(self._h, C.c_void_p(device_ptr), cap, C.byref(n)), self.ctx._h)
        return int(n.value)

    def copy_fused_device(self, device_ptr: int, cap: int) -> None:
        """copy_fused without the host knowing the count: header {count, status} written by a kernel from the scan's device
        counters (status != 0: the scan needs the host -- call view() / stats() and copy again).  No synchronisation."""
        _check(capi.lib().spm_hip_hits_copy_fused_device(self._h, C.c_void_p(device_ptr), cap), self.ctx._h)

    def stats(self) -> capi.ScanStats:
        s = capi.ScanStats()
        _check(capi.lib().spm_hip_hits_stats(self._h, C.byref(s)), self.ctx._h)
        return s

    def checksum(self) -> int:
        return int(capi.lib().spm_hip_hits_checksum(self._h))

    def close(self):
        if self._h:
            if self.ctx._h:  # the C objects point at their context: once it is gone there is nothing left to release
                capi.lib().spm_hip_hits_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scan(ctx: Context, text: Text, pats: PatternSet, begin: int = 0, end: int | None = None, *,
         engine: int = capi.ENGINE_AUTO, left_context: bool = False, pos_offset: int = 0, max_hits: int = 0,
         state_in: np.ndarray | None = None, want_state: bool = False, flags: int = 0):
    """One scan of text[begin:end) -- seqan_pattern_base::operator() for a whole needle set.

    Returns Hits, or (Hits, state_out) when want_state is set."""
    end = len(text) if end is None else end
    opts = capi.ScanOpts(engine=engine, left_context=1 if left_context else 0, pos_offset=pos_offset,
                         max_hits=max_hits, flags=flags, reserved=0)
    h = C.c_void_p()
    st_in = state_in.ctypes.data if state_in is not None else None
    st_out = None
    if want_state:
        st_out = np.zeros(pats.state_stride() * max(1, pats.n), dtype=np.uint8)
    _check(capi.lib().spm_hip_scan(ctx._h, text._h, begin, end, pats._h, C.byref(opts), st_in,
                                   st_out.ctypes.data if st_out is not None else None, C.byref(h)), ctx._h)
    hits = Hits(ctx, h)
    return (hits, st_out) if want_state else hits


def scan_segments(ctx: Context, text: Text, pats: PatternSet, seg_offsets, *, engine: int = capi.ENGINE_AUTO,
                  max_hits: int = 0, flags: int = 0) -> Hits:
    """Scan a batch of independent haystacks stored back to back (segment s = text[off[s]:off[s+1]]) in one launch."""
    offs = np.ascontiguousarray(seg_offsets, dtype=np.uint64)
    opts = capi.ScanOpts(engine=engine, left_context=0, pos_offset=0, max_hits=max_hits, flags=flags, reserved=0)
    h = C.c_void_p()
    _check(capi.lib().spm_hip_scan_segments(ctx._h, text._h, offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            len(offs) - 1, pats._h, C.byref(opts), C.byref(h)), ctx._h)
    return Hits(ctx, h)


def synth_pattern(seed_text: int, seed_pat: int, n_total: int, p: int, L: int, kmax: int):
    out = np.empty(L, dtype=np.uint8)
    o = capi.lib().spm_hip_synth_pattern(seed_text, seed_pat, n_total, p, L, kmax,
                                         out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, int(o)


def synth_repeat_pattern(seed_text: int, seed_pat: int, n_total: int, p: int, L: int, kmax: int, repeat_ppm: int,
                         across_every: int = 8):
    out = np.empty(L, dtype=np.uint8)
    o = capi.lib().spm_hip_synth_repeat_pattern(seed_text, seed_pat, n_total, p, L, kmax, repeat_ppm, across_every,
                                                out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, int(o)


def synth_repeat_text(seed: int, repeat_ppm: int, begin: int, n: int) -> np.ndarray:
    out = np.empty(n, dtype=np.uint8)
    capi.lib().spm_hip_synth_repeat_text(seed, repeat_ppm, begin, n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


ALLELE_DTYPE = np.dtype([("pos", "<u8"), ("ref_len", "<u4"), ("alt_len", "<u4"), ("alt_off", "<u8")])
JST_HIT_DTYPE = np.dtype([("pos", "<u8"), ("haplotype", "<u4"), ("pattern", "<u4"), ("score", "<i4"),
                          ("reserved", "<u4")])


def synth_variants(seed_text: int, seed_var: int, ref_begin: int, n_ref: int, n_haplotypes: int):
    """The synthetic variants of config C5: (alleles, alt_pool, coverage) as numpy arrays."""
    na, npool = C.c_uint64(0), C.c_uint64(0)
    rc = capi.lib().spm_hip_jst_synth_variants(seed_text, seed_var, ref_begin, n_ref, n_haplotypes, None, C.byref(na),
                                               None, C.byref(npool), None)
    if rc != 0:
        raise capi.SpmError("spm_hip_jst_synth_variants: invalid argument")
    alleles = np.zeros(max(1, na.value), dtype=ALLELE_DTYPE)
    pool = np.zeros(max(1, npool.value), dtype=np.uint8)
    cov = np.zeros(max(1, na.value), dtype=np.uint64)
    rc = capi.lib().spm_hip_jst_synth_variants(seed_text, seed_var, ref_begin, n_ref, n_haplotypes,
                                               alleles.ctypes.data_as(C.POINTER(capi.JstAllele)), C.byref(na),
                                               pool.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(npool),
                                               cov.ctypes.data_as(C.POINTER(C.c_uint64)))
    if rc != 0:
        raise capi.SpmError("spm_hip_jst_synth_variants failed")
    return alleles[:na.value], pool[:npool.value], cov[:na.value]


class Jst:
    """Journaled sequence tree in HBM: a reference Text + alleles + per-allele haplotype coverage."""

    def __init__(self, ctx: Context, reference: Text, alleles, alt_pool, coverage, n_haplotypes: int):
        self.ctx, self.reference, self.n_haplotypes = ctx, reference, n_haplotypes
        al = np.ascontiguousarray(alleles, dtype=ALLELE_DTYPE)
        pool = np.ascontiguousarray(alt_pool, dtype=np.uint8)
        cw = (n_haplotypes + 63) // 64
        cov = np.ascontiguousarray(coverage, dtype=np.uint64).reshape(-1)
        if cov.size != len(al) * cw:
            raise ValueError("coverage must hold ceil(n_haplotypes / 64) words per allele")
        h = C.c_void_p()
        _check(capi.lib().spm_hip_jst_create(ctx._h, reference._h, al.ctypes.data_as(C.POINTER(capi.JstAllele)),
                                             len(al), pool.ctypes.data_as(C.POINTER(C.c_uint8)), pool.size,
                                             cov.ctypes.data_as(C.POINTER(C.c_uint64)), n_haplotypes, C.byref(h)),
               ctx._h)
        self._h = h

    def haplotype_length(self, h: int) -> int:
        return int(capi.lib().spm_hip_jst_haplotype_length(self._h, h))

    def extract(self, h: int, begin: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint8)
        _check(capi.lib().spm_hip_jst_extract(self._h, h, begin, n, out.ctypes.data_as(C.POINTER(C.c_uint8))),
               self.ctx._h)
        return out

    def index(self, window: int, block_len: int = 0, block_begin: int = 0, block_end: int = 0):
        _check(capi.lib().spm_hip_jst_index(self._h, window, block_len, block_begin, block_end), self.ctx._h)
        return self.stats()

    def stats(self) -> capi.JstStats:
        st = capi.JstStats()
        _check(capi.lib().spm_hip_jst_stats(self._h, C.byref(st)), self.ctx._h)
        return st

    def search_device(self, pats: PatternSet, *, engine: int = capi.ENGINE_AUTO, max_hits: int = 0) -> "JstHits":
        """One search over all haplotypes; the records stay in HBM (arrival order) until view()/copy_to()."""
        opts = capi.ScanOpts(engine=engine, left_context=0, pos_offset=0, max_hits=max_hits, flags=0, reserved=0)
        hh = C.c_void_p()
        _check(capi.lib().spm_hip_jst_search(self._h, pats._h, C.byref(opts), C.byref(hh)), self.ctx._h)
        return JstHits(self.ctx, hh)

    def search(self, pats: PatternSet, *, engine: int = capi.ENGINE_AUTO, max_hits: int = 0) -> np.ndarray:
        """All hits over all haplotypes, sorted by (haplotype, pos, pattern): JST_HIT_DTYPE records."""
        h = self.search_device(pats, engine=engine, max_hits=max_hits)
        try:
            return h.view()
        finally:
            h.close()

    def close(self):
        if self._h:
            if self.ctx._h:
                capi.lib().spm_hip_jst_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class JstHits:
    def __init__(self, ctx, h):
        self.ctx, self._h = ctx, h

    def __len__(self):
        p, n = C.c_void_p(), C.c_uint64(0)
        _check(capi.lib().spm_hip_jst_hits_device(self._h, C.byref(p), C.byref(n)), self.ctx._h)
        return int(n.value)

    def view(self) -> np.ndarray:
        rec = C.POINTER(capi.JstHit)()
        n = C.c_uint64(0)
        _check(capi.lib().spm_hip_jst_hits_view(self._h, C.byref(rec), C.byref(n)), self.ctx._h)
        if n.value == 0:
            return np.zeros(0, dtype=JST_HIT_DTYPE)
        buf = (capi.JstHit * n.value).from_address(C.addressof(rec.contents))
        return np.frombuffer(buf, dtype=JST_HIT_DTYPE).copy()

    def copy_to(self, device_ptr: int, cap: int) -> int:
        n = C.c_uint64(0)
        _check(capi.lib().spm_hip_jst_hits_copy_device(self._h, device_ptr, cap, C.byref(n)), self.ctx._h)
        return int(n.value)

    def close(self):
        if self._h:
            if self.ctx._h:
                capi.lib().spm_hip_jst_hits_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
