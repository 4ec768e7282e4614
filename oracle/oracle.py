"""ctypes loader for the CPU oracle (oracle/libspm_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (libspm_amd) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libspm_oracle.so")

MAX_BLOCKS = 32
INFIX, PREFIX = 0, 1
SHIFTOR, MYERS, HORSPOOL = 0, 1, 2

HIT_DTYPE = np.dtype([("pos", "<u8"), ("pattern", "<u4"), ("score", "<i4")])


class MyersState(C.Structure):
    _fields_ = [
        ("vp", C.c_uint64 * MAX_BLOCKS),
        ("vn", C.c_uint64 * MAX_BLOCKS),
        ("score", C.c_int32 * MAX_BLOCKS),
        ("n_blocks", C.c_uint32),
        ("active", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "spm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def use_native_build() -> bool:
    """bench.py's cpu_baseline leg: rebuild the library for THIS machine's CPU (-march=native) and load that copy, so the
    timed loop is what a local build of the reference would get.  Falls back to the portable build (False)."""
    global _SO, _lib
    try:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
    except Exception:
        return False
    _SO = os.path.join(_HERE, "libspm_oracle_native.so")
    _lib = None
    return True


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p = C.POINTER(C.c_uint8)
        L.spm_oracle_horspool.restype = C.c_size_t
        L.spm_oracle_horspool.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_size_t]
        L.spm_oracle_naive_exact.restype = C.c_size_t
        L.spm_oracle_naive_exact.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.spm_oracle_shiftor.restype = C.c_size_t
        L.spm_oracle_shiftor.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_uint64,
                                         C.c_void_p, C.c_size_t]
        L.spm_oracle_myers_init.restype = None
        L.spm_oracle_myers_init.argtypes = [C.POINTER(MyersState), C.c_size_t, C.c_uint32, C.c_int]
        L.spm_oracle_myers_scan.restype = C.c_size_t
        L.spm_oracle_myers_scan.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int,
                                            C.c_int, C.POINTER(MyersState), C.c_uint64, C.c_void_p, C.c_size_t]
        L.spm_oracle_myers2_fast.restype = C.c_size_t
        L.spm_oracle_myers2_fast.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
                                             C.c_void_p, C.c_size_t]
        L.spm_oracle_sellers.restype = C.c_size_t
        L.spm_oracle_sellers.argtypes = [u8p, C.c_size_t, u8p, C.c_size_t, C.c_uint32, C.c_int, C.c_void_p,
                                         C.c_uint64, C.c_void_p, C.c_size_t]
        L.spm_oracle_mix64.restype = C.c_uint64
        L.spm_oracle_mix64.argtypes = [C.c_uint64]
        L.spm_oracle_text.restype = None
        L.spm_oracle_text.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, u8p]
        L.spm_oracle_repeat_text.restype = None
        L.spm_oracle_repeat_text.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p]
        L.spm_oracle_pattern.restype = C.c_uint64
        L.spm_oracle_pattern.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.spm_oracle_checksum.restype = C.c_uint64
        L.spm_oracle_checksum.argtypes = [C.c_void_p, C.c_size_t]
        L.spm_oracle_scan_multi.restype = C.c_size_t
        L.spm_oracle_scan_multi.argtypes = [C.c_int, u8p, C.c_size_t, u8p, C.POINTER(C.c_uint32), C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t]
        L.spm_oracle_char_to_rank.restype = C.c_uint8
        L.spm_oracle_char_to_rank.argtypes = [C.c_uint32, C.c_char]
        L.spm_oracle_rank_to_char.restype = C.c_char
        L.spm_oracle_rank_to_char.argtypes = [C.c_uint32, C.c_uint8]
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def encode(s: str, sigma: int = 4) -> np.ndarray:
    L = lib()
    return np.array([L.spm_oracle_char_to_rank(sigma, ch.encode()) for ch in s], dtype=np.uint8)


def decode(r, sigma: int = 4) -> str:
    L = lib()
    return "".join(L.spm_oracle_rank_to_char(sigma, int(x)).decode() for x in r)


def _positions(fn, text, pat, *extra, cap=None):
    t, tp = _u8(text)
    p, pp = _u8(pat)
    cap = cap or max(16, len(t) + 1)
    out = np.zeros(cap, dtype=np.uint64)
    n = fn(tp, len(t), pp, len(p), *extra, out.ctypes.data, cap)
    assert n <= cap
    return out[:n].copy()


def horspool(text, pat, sigma=4):
    return _positions(lib().spm_oracle_horspool, text, pat, sigma)


def naive_exact(text, pat):
    return _positions(lib().spm_oracle_naive_exact, text, pat)


def shiftor(text, pat, sigma=4, state=None, text_offset=0):
    """state: np.uint32 array of ceil(m/32) words (mutated in place) or None."""
    sp = state.ctypes.data if state is not None else None
    return _positions(lib().spm_oracle_shiftor, text, pat, sigma, sp, text_offset)


def shiftor_state(m):
    return np.full((m + 31) // 32, 0xFFFFFFFF, dtype=np.uint32)


def myers_state(m, k, cutoff=False):
    st = MyersState()
    lib().spm_oracle_myers_init(C.byref(st), m, k, 1 if cutoff else 0)
    return st


def myers(text, pat, k, sigma=4, mode=INFIX, variant=None, state=None, text_offset=0):
    """Returns HIT_DTYPE array (pos = exclusive end, score = edit distance)."""
    t, tp = _u8(text)
    p, pp = _u8(pat)
    m = len(p)
    if variant is None:
        variant = 0 if m <= 64 else 2
    if state is None:
        state = myers_state(m, k, cutoff=(variant == 2))
    cap = max(16, len(t) + 1)
    out = np.zeros(cap, dtype=HIT_DTYPE)
    n = lib().spm_oracle_myers_scan(tp, len(t), pp, m, sigma, k, mode, variant, C.byref(state), text_offset,
                                    out.ctypes.data, cap)
    return out[:n].copy()


def myers2_fast(text, pat, k, sigma=4, text_offset=0):
    """The two-block register loop the CPU baseline times (64 < |P| <= 128)."""
    t, tp = _u8(text)
    p, pp = _u8(pat)
    cap = len(t) + 1
    out = np.zeros(cap, dtype=HIT_DTYPE)
    n = lib().spm_oracle_myers2_fast(tp, len(t), pp, len(p), sigma, k, text_offset, out.ctypes.data, cap)
    assert n != 2**64 - 1
    return out[:n].copy()


def sellers(text, pat, k, mode=INFIX, col=None, text_offset=0):
    t, tp = _u8(text)
    p, pp = _u8(pat)
    cap = max(16, len(t) + 1)
    out = np.zeros(cap, dtype=HIT_DTYPE)
    cp = col.ctypes.data if col is not None else None
    n = lib().spm_oracle_sellers(tp, len(t), pp, len(p), k, mode, cp, text_offset, out.ctypes.data, cap)
    return out[:n].copy()


def text(seed, begin, n):
    out = np.empty(n, dtype=np.uint8)
    lib().spm_oracle_text(seed, begin, n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def repeat_text(seed, ppm, begin, n):
    out = np.empty(n, dtype=np.uint8)
    lib().spm_oracle_repeat_text(seed, ppm, begin, n, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def pattern(seed_text, seed_pat, n_total, p, L, kmax):
    out = np.empty(L, dtype=np.uint8)
    o = lib().spm_oracle_pattern(seed_text, seed_pat, n_total, p, L, kmax, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, int(o)


def checksum(hits: np.ndarray) -> int:
    h = np.ascontiguousarray(hits, dtype=HIT_DTYPE)
    return int(lib().spm_oracle_checksum(h.ctypes.data, len(h)))


def scan_multi(algo, text, patterns, k=0, sigma=4, threads=1, cap=1 << 22):
    """patterns: list of uint8 arrays. Returns hits sorted by (pattern, pos)."""
    t, tp = _u8(text)
    offs = np.zeros(len(patterns) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(p) for p in patterns])
    cat = np.concatenate([np.asarray(p, dtype=np.uint8) for p in patterns]) if patterns else np.zeros(0, np.uint8)
    cat, cp = _u8(cat)
    out = np.zeros(cap, dtype=HIT_DTYPE)
    n = lib().spm_oracle_scan_multi(algo, tp, len(t), cp, offs.ctypes.data_as(C.POINTER(C.c_uint32)), len(patterns),
                                    sigma, k, threads, out.ctypes.data, cap)
    if n == C.c_size_t(-1).value:
        raise OverflowError("oracle hit buffer overflow")
    return out[:n].copy()
