"""ctypes binding of the C ABI in include/spm_hip.h (libspm_amd/libspm_hip.so).

There is no CPU scan path: if the shared library is missing or HIP fails, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_PKG, "libspm_hip.so")
CSRC = os.path.join(_PKG, "csrc")

ALGO_SHIFTOR, ALGO_MYERS, ALGO_MYERS_PREFIX, ALGO_HORSPOOL = 0, 1, 2, 3
ENGINE_AUTO, ENGINE_BRUTE, ENGINE_FILTER = 0, 1, 2
SCAN_IGNORE_PACKED = 1
SCAN_DEFER = 2
MAX_NEEDLE = 2048


class SpmError(RuntimeError):
    pass


class Hit(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("pattern", C.c_uint32), ("score", C.c_int32)]


class ScanOpts(C.Structure):
    _fields_ = [
        ("engine", C.c_uint32),
        ("left_context", C.c_uint32),
        ("pos_offset", C.c_uint64),
        ("max_hits", C.c_uint64),
        ("flags", C.c_uint32),
        ("reserved", C.c_uint32),
    ]


class ScanStats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_float),
        ("ms_main", C.c_float),
        ("ms_verify", C.c_float),
        ("engine_used", C.c_uint32),
        ("fell_back", C.c_uint32),
        ("n_candidates", C.c_uint64),
        ("n_hits", C.c_uint64),
        ("main_launches", C.c_uint32),
        ("n_bands", C.c_uint32),
        ("fallback_spans", C.c_uint32),
        ("reserved", C.c_uint32),
        ("fallback_symbols", C.c_uint64),
    ]


class BuildStats(C.Structure):
    _fields_ = [
        ("ms_total", C.c_float),
        ("ms_tables", C.c_float),
        ("ms_index", C.c_float),
        ("ms_upload", C.c_float),
        ("threads", C.c_uint32),
        ("passes", C.c_uint32),
        ("dense", C.c_uint32),
        ("anchor_sixteenths", C.c_uint32),
        ("keys", C.c_uint64),
        ("stride", C.c_uint32),
        ("key_len", C.c_uint32),
        ("bytes_device", C.c_uint64),
    ]


class JstAllele(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("ref_len", C.c_uint32), ("alt_len", C.c_uint32), ("alt_off", C.c_uint64)]


class JstHit(C.Structure):
    _fields_ = [("pos", C.c_uint64), ("haplotype", C.c_uint32), ("pattern", C.c_uint32), ("score", C.c_int32),
                ("reserved", C.c_uint32)]


class JstStats(C.Structure):
    _fields_ = [
        ("haplotype_symbols", C.c_uint64),
        ("context_symbols", C.c_uint64),
        ("contexts", C.c_uint64),
        ("unique_contexts", C.c_uint64),
        ("n_blocks", C.c_uint64),
        ("block_len", C.c_uint32),
        ("window", C.c_uint32),
        ("ms_index", C.c_float),
        ("ms_scan", C.c_float),
        ("ms_main", C.c_float),
        ("ms_verify", C.c_float),
        ("ms_fanout", C.c_float),
        ("engine_used", C.c_uint32),
        ("main_launches", C.c_uint32),
        ("fell_back", C.c_uint32),
        ("segment_hits", C.c_uint64),
        ("candidates", C.c_uint64),
        ("bands", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    """Compile libspm_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp"))]
    srcs.append(os.path.join(_PKG, "..", "include", "spm_hip.h"))
    stale = not os.path.exists(SO_PATH) or any(os.path.getmtime(s) > os.path.getmtime(SO_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s"])
    return SO_PATH


_lib = None


def lib():
    """Load the shared library; fail loudly when it is absent (there is no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise SpmError(f"{SO_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). libspm_amd has no CPU fallback.")
    L = C.CDLL(SO_PATH)
    vp, u8p, u32p, u16p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)
    sig = {
        "spm_hip_init": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
        "spm_hip_destroy": (None, [vp]),
        "spm_hip_last_error": (C.c_char_p, [vp]),
        "spm_hip_synchronize": (C.c_int, [vp]),
        "spm_hip_text_upload": (C.c_int, [vp, u8p, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
        "spm_hip_text_wrap": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
        "spm_hip_text_generate": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(vp)]),
        "spm_hip_text_generate_repeats": (C.c_int, [vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
        "spm_hip_text_pack": (C.c_int, [vp, vp]),
        "spm_hip_text_is_packed": (C.c_int, [vp]),
        "spm_hip_text_download": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, u8p]),
        "spm_hip_text_length": (C.c_uint64, [vp]),
        "spm_hip_text_device_ptr": (vp, [vp]),
        "spm_hip_text_destroy": (None, [vp]),
        "spm_hip_patterns_create": (C.c_int, [vp, C.c_int, u8p, u32p, C.c_uint32, u16p, C.c_uint32, C.POINTER(vp)]),
        "spm_hip_patterns_destroy": (None, [vp]),
        "spm_hip_patterns_window_size": (C.c_uint64, [vp, C.c_uint32]),
        "spm_hip_patterns_filterable": (C.c_int, [vp]),
        "spm_hip_patterns_build_stats": (C.c_int, [vp, C.POINTER(BuildStats)]),
        "spm_hip_patterns_state_stride": (C.c_size_t, [vp]),
        "spm_hip_patterns_state_init": (C.c_int, [vp, vp]),
        "spm_hip_scan": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, vp, C.POINTER(ScanOpts), vp, vp, C.POINTER(vp)]),
        "spm_hip_scan_segments": (C.c_int, [vp, vp, C.POINTER(C.c_uint64), C.c_uint64, vp, C.POINTER(ScanOpts),
                                            C.POINTER(vp)]),
        "spm_hip_hits_view": (C.c_int, [vp, C.POINTER(C.POINTER(Hit)), C.POINTER(C.c_uint64)]),
        "spm_hip_hits_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "spm_hip_hits_copy_device": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        "spm_hip_hits_copy_fused": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        "spm_hip_hits_copy_fused_device": (C.c_int, [vp, vp, C.c_uint64]),
        "spm_hip_hits_stats": (C.c_int, [vp, C.POINTER(ScanStats)]),
        "spm_hip_hits_checksum": (C.c_uint64, [vp]),
        "spm_hip_hits_destroy": (None, [vp]),
        "spm_hip_synth_pattern": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                               C.c_uint32, u8p]),
        "spm_hip_synth_repeat_pattern": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32,
                                                      C.c_uint32, C.c_uint32, C.c_uint32, u8p]),
        "spm_hip_synth_repeat_text": (None, [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, u8p]),
        "spm_hip_mix64": (C.c_uint64, [C.c_uint64]),
        "spm_hip_host_selftest": (C.c_int, [C.c_int, u8p, u32p, C.c_uint32, u16p, C.c_uint32, C.POINTER(C.c_uint64)]),
        "spm_hip_version": (C.c_char_p, []),
        "spm_hip_jst_create": (C.c_int, [vp, vp, C.POINTER(JstAllele), C.c_uint64, u8p, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(vp)]),
        "spm_hip_jst_destroy": (None, [vp]),
        "spm_hip_jst_haplotype_length": (C.c_uint64, [vp, C.c_uint32]),
        "spm_hip_jst_extract": (C.c_int, [vp, C.c_uint32, C.c_uint64, C.c_uint64, u8p]),
        "spm_hip_jst_index": (C.c_int, [vp, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64]),
        "spm_hip_jst_search": (C.c_int, [vp, vp, C.POINTER(ScanOpts), C.POINTER(vp)]),
        "spm_hip_jst_stats": (C.c_int, [vp, C.POINTER(JstStats)]),
        "spm_hip_jst_hits_view": (C.c_int, [vp, C.POINTER(C.POINTER(JstHit)), C.POINTER(C.c_uint64)]),
        "spm_hip_jst_hits_device": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]),
        "spm_hip_jst_hits_copy_device": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(C.c_uint64)]),
        "spm_hip_jst_hits_destroy": (None, [vp]),
        "spm_hip_comm_unique_id": (C.c_int, [vp]),
        "spm_hip_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(vp)]),
        "spm_hip_comm_destroy": (None, [vp]),
        "spm_hip_gatherv_hits": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "spm_hip_gatherv_jst_hits": (C.c_int, [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_uint64)]),
        "spm_hip_gatherv_plan": (C.c_int, [C.POINTER(C.c_uint64), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
        "spm_hip_comm_selftest": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_int)]),
        "spm_hip_jst_synth_variants": (C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32,
                                                 C.POINTER(JstAllele), C.POINTER(C.c_uint64), u8p,
                                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


EXPORTS = [
    "spm_hip_init", "spm_hip_destroy", "spm_hip_last_error", "spm_hip_synchronize", "spm_hip_text_upload",
    "spm_hip_text_wrap", "spm_hip_text_generate", "spm_hip_text_generate_repeats", "spm_hip_text_pack", "spm_hip_text_is_packed",
    "spm_hip_text_download", "spm_hip_text_length",
    "spm_hip_text_device_ptr", "spm_hip_text_destroy", "spm_hip_patterns_create", "spm_hip_patterns_destroy",
    "spm_hip_patterns_window_size", "spm_hip_patterns_filterable", "spm_hip_patterns_build_stats", "spm_hip_patterns_state_stride",
    "spm_hip_patterns_state_init", "spm_hip_scan", "spm_hip_scan_segments", "spm_hip_hits_view", "spm_hip_hits_device",
    "spm_hip_hits_copy_device", "spm_hip_hits_copy_fused", "spm_hip_hits_copy_fused_device", "spm_hip_hits_stats", "spm_hip_hits_checksum", "spm_hip_hits_destroy", "spm_hip_synth_pattern",
    "spm_hip_synth_repeat_pattern", "spm_hip_synth_repeat_text", "spm_hip_mix64", "spm_hip_host_selftest", "spm_hip_version",
    "spm_hip_jst_create", "spm_hip_jst_destroy", "spm_hip_jst_haplotype_length", "spm_hip_jst_extract",
    "spm_hip_jst_index", "spm_hip_jst_search", "spm_hip_jst_stats", "spm_hip_jst_hits_view", "spm_hip_jst_hits_device",
    "spm_hip_jst_hits_copy_device", "spm_hip_jst_hits_destroy", "spm_hip_jst_synth_variants",
    "spm_hip_comm_unique_id", "spm_hip_comm_init", "spm_hip_comm_destroy", "spm_hip_gatherv_hits",
    "spm_hip_gatherv_jst_hits", "spm_hip_gatherv_plan", "spm_hip_comm_selftest",
]
