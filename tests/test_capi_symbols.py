"""CPU: the C-ABI shared library loads and exports every symbol include/spm_hip.h declares; calls that need a
device fail loudly with an error code and message (no silent fallback)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "spm_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(spm_hip_\w+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported(spm):
    names = _declared()
    assert len(names) >= 25
    L = ctypes.CDLL(spm.capi.SO_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/spm_hip.h but not exported by libspm_hip.so"
    assert sorted(spm.capi.EXPORTS) == names, "libspm_amd/capi.py binding list out of sync with the header"


def test_version_and_struct_layout(spm):
    assert b"gfx950" in spm.capi.lib().spm_hip_version()
    assert ctypes.sizeof(spm.capi.Hit) == 16
    assert spm.HIT_DTYPE.itemsize == 16
    assert ctypes.sizeof(spm.capi.ScanOpts) == 32
    assert ctypes.sizeof(spm.capi.ScanStats) == 64
    # journaled-sequence records (static_assert'ed to the same sizes in csrc/jst.hpp)
    assert ctypes.sizeof(spm.capi.JstAllele) == 24 and spm.ALLELE_DTYPE.itemsize == 24
    assert ctypes.sizeof(spm.capi.JstHit) == 24 and spm.JST_HIT_DTYPE.itemsize == 24
    assert ctypes.sizeof(spm.capi.JstStats) == 104


def test_no_device_fails_loudly(spm):
    import torch
    if torch.cuda.is_available():
        return
    try:
        spm.Context(0)
    except spm.SpmError as e:
        assert "spm_hip_init failed" in str(e)
    else:
        raise AssertionError("Context(0) succeeded without a GPU")


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under libspm_amd/ or include/ may reference it."""
    bad = []
    for base in ("libspm_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                    src = open(os.path.join(d, f), errors="ignore").read()
                    if re.search(r"(import\s+oracle|from\s+oracle|spm_oracle\.h|libspm_oracle)", src):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_gatherv_plan_offsets(spm):
    """Host arithmetic of the native gatherv (spm_hip_gatherv_plan): rank r's records land at offsets[r] bytes."""
    import numpy as np
    L = spm.capi.lib()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    counts = np.array([3, 0, 7, 1], dtype=np.uint64)
    offs = np.zeros(5, dtype=np.uint64)
    assert L.spm_hip_gatherv_plan(counts.ctypes.data_as(u64p), 4, 16, offs.ctypes.data_as(u64p)) == 0
    assert offs.tolist() == [0, 48, 48, 160, 176]
    assert L.spm_hip_gatherv_plan(counts.ctypes.data_as(u64p), 4, 24, offs.ctypes.data_as(u64p)) == 0
    assert offs.tolist() == [0, 72, 72, 240, 264]
    huge = np.array([1 << 62, 1 << 62], dtype=np.uint64)
    assert L.spm_hip_gatherv_plan(huge.ctypes.data_as(u64p), 2, 16, offs.ctypes.data_as(u64p)) == -5   # SPM_E_OVERFLOW
    assert L.spm_hip_gatherv_plan(None, 2, 16, offs.ctypes.data_as(u64p)) == -1


def test_native_gatherv_protocol_never_leaves_a_rank_waiting(spm):
    """The native RCCL gatherv (csrc/comm.hpp) as a protocol over an in-process loopback of `world` threads
    (csrc/comm_protocol.hpp, spm_hip_comm_selftest): the root holds every rank's records at the planned offsets; a rank
    with a local error (its scan overflowed), a root that cannot allocate, counts that overflow the offsets -- every rank
    returns an error (the failing one its own, the others SPM_E_PEER) before any send or receive is posted; nothing hangs."""
    L = spm.capi.lib()
    OK, NOMEM, OVERFLOW, PEER = 0, -3, -5, -6
    for world in (1, 2, 3, 4, 8):
        for root in sorted({0, world - 1}):
            for rec in (16, 24):   # spm_hit / spm_jst_hit
                d = (ctypes.c_int * world)()
                assert L.spm_hip_comm_selftest(world, root, 0, 0, rec, 1234 + world, d) == OK and list(d) == [OK] * world
                assert L.spm_hip_comm_selftest(world, root, 4, 0, rec, 1, d) == OK and list(d) == [OK] * world   # nobody has records
            for victim in sorted({0, world // 2, world - 1}):
                d = (ctypes.c_int * world)()
                assert L.spm_hip_comm_selftest(world, root, 1, victim, 16, 99, d) == OK
                assert list(d) == [OVERFLOW if r == victim else PEER for r in range(world)]
            d = (ctypes.c_int * world)()
            assert L.spm_hip_comm_selftest(world, root, 2, 0, 16, 7, d) == OK
            assert list(d) == [NOMEM if r == root else PEER for r in range(world)]
            if world > 1:
                assert L.spm_hip_comm_selftest(world, root, 3, 0, 16, 7, d) == OK and list(d) == [OVERFLOW] * world
    assert L.spm_hip_comm_selftest(0, 0, 0, 0, 16, 1, None) == -1
