"""Child process of tests/test_host_sanitizers.py: loads the ASan + UBSan builds of the host-only product code
(libspm_amd/libspm_host_asan.so: seed index build, match-mask tables, gatherv protocol) and of the CPU oracle
(oracle/libspm_oracle_asan.so) and drives them over the shapes the CPU tests use.  Run with LD_PRELOAD=libasan.so (the
interpreter is not instrumented).  A sanitizer report ends the process with a non-zero code (halt_on_error / no-recover)."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = C.CDLL(os.path.join(ROOT, "libspm_amd", "libspm_host_asan.so"))
O = C.CDLL(os.path.join(ROOT, "oracle", "libspm_oracle_asan.so"))
u8p, u32p, u16p, u64p = (C.POINTER(t) for t in (C.c_uint8, C.c_uint32, C.c_uint16, C.c_uint64))
H.spm_hip_host_selftest.restype = C.c_int
H.spm_hip_host_selftest.argtypes = [C.c_int, u8p, u32p, C.c_uint32, u16p, C.c_uint32, u64p]
H.spm_host_tables_check.restype = C.c_uint64
H.spm_host_tables_check.argtypes = [C.c_int, u8p, u32p, C.c_uint32, C.c_uint32, C.c_uint32]
H.spm_hip_comm_selftest.restype = C.c_int
H.spm_hip_comm_selftest.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_int)]
SHIFTOR, MYERS, PREFIX, HORSPOOL = 0, 1, 2, 3
done = []


def cat(needles):
    offs = np.zeros(len(needles) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(x) for x in needles])
    return np.ascontiguousarray(np.concatenate(needles), dtype=np.uint8), offs


def selftest(name, algo, needles, k, sigma=4, env=None):
    c, offs = cat(needles)
    ks = np.full(len(needles), k, dtype=np.uint16) if np.isscalar(k) else np.asarray(k, dtype=np.uint16)
    st = (C.c_uint64 * 8)()
    for kk, v in (env or {}).items():
        os.environ[kk] = v
    try:
        rc = H.spm_hip_host_selftest(algo, c.ctypes.data_as(u8p), offs.ctypes.data_as(u32p), len(needles),
                                     ks.ctypes.data_as(u16p), sigma, st)
    finally:
        for kk in (env or {}):
            os.environ.pop(kk, None)
    assert rc == 0 and st[4] == 0, (name, rc, list(st))
    wrong = H.spm_host_tables_check(algo, c.ctypes.data_as(u8p), offs.ctypes.data_as(u32p), len(needles), sigma, 4)
    assert wrong == 0, (name, "match-mask tables", wrong)
    done.append({"case": name, "passes": int(st[0]), "keys": int(st[2])})


rng = np.random.default_rng(11)
rnd = lambda m, n: [rng.integers(0, 4, m, dtype=np.uint8) for _ in range(n)]
selftest("c3-shaped", MYERS, rnd(100, 1024), 3)
selftest("c2-shaped", SHIFTOR, rnd(32, 1024), 0)
selftest("horspool", HORSPOOL, rnd(32, 16), 0)
selftest("prefix matcher (tables only: no seed index)", PREFIX, rnd(70, 9), 2)
selftest("sub-batches, sparse", MYERS, rnd(150, 3000), 3, env={"SPM_HIP_FILTER_DENSE": "0", "SPM_HIP_FILTER_MAX_KEYS": "4096"})
selftest("anchored passes", MYERS, rnd(150, 2500), 3,
         env={"SPM_HIP_FILTER_DENSE": "0", "SPM_HIP_FILTER_MAX_KEYS": "2048", "SPM_HIP_FILTER_STRIDE": "1"})
selftest("dense pass", MYERS, rnd(150, 6000), 3, env={"SPM_HIP_FILTER_DENSE": "2"})
selftest("dense pass, short needles", MYERS, rnd(64, 300) + rnd(100, 300), 3, env={"SPM_HIP_FILTER_DENSE": "2"})
selftest("dense pass, repeats", MYERS, rnd(150, 500) + [np.zeros(150, np.uint8), np.resize(np.array([0, 1], np.uint8), 150)], 3,
         env={"SPM_HIP_FILTER_DENSE": "2"})
selftest("dense pass, one thread", MYERS, rnd(120, 700), 2, env={"SPM_HIP_FILTER_DENSE": "2", "SPM_HIP_BUILD_THREADS": "1"})
selftest("mixed lengths and k", MYERS, [rng.integers(0, 4, m, dtype=np.uint8) for m in (64, 100, 150, 300, 1000)], [3, 3, 5, 10, 40])
selftest("long needles", MYERS, rnd(1024, 8) + rnd(2047, 2), 64)
selftest("short seeds", MYERS, rnd(44, 100), 3)
d5 = [rng.choice(np.array([0, 1, 2, 4], dtype=np.uint8), 100) for _ in range(64)]
d5[7][50] = 3
d5[8][[0, 31, 62, 99]] = 3
selftest("dna5 with N", MYERS, d5, 3, sigma=5)
d15 = [np.array([0, 2, 4, 11], dtype=np.uint8)[rng.integers(0, 4, 100)] for _ in range(32)]
d15[3][40] = 8
selftest("dna15", MYERS, d15, 3, sigma=15)
selftest("bloom cascade", MYERS, rnd(100, 512), 3, env={"SPM_HIP_FILTER_HASH": "1"})
for world in (1, 2, 5, 8):
    for sc in range(5):
        det = (C.c_int * world)()
        assert H.spm_hip_comm_selftest(world, world - 1, sc, world // 2, 24, 77 + sc, det) == 0, (world, sc, list(det))
done.append({"case": "gatherv protocol, worlds 1 2 5 8 x 5 scenarios"})

# ---- the oracle: the reference's golden rows and the generators, under the sanitizers ----
sys.path.insert(0, ROOT)
import oracle.oracle as OO  # noqa: E402

OO._SO = os.path.join(ROOT, "oracle", "libspm_oracle_asan.so")
OO._lib = None
G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")))
Hs, Pn = OO.encode(G["haystack"]), OO.encode(G["needle"])
for case in G["cases"]:
    if case["matcher"] == "horspool":
        got = OO.horspool(Hs, Pn).tolist()
    elif case["matcher"] == "shiftor":
        got = OO.shiftor(Hs, Pn).tolist()
    elif "chunk_size" in case:
        st, got = OO.myers_state(len(Pn), case["k"]), []
        for off in range(0, len(Hs), case["chunk_size"]):
            got += OO.myers(Hs[off:off + case["chunk_size"]], Pn, case["k"], state=st, text_offset=off)["pos"].tolist()
    else:
        got = OO.myers(Hs, Pn, case["k"])["pos"].tolist()
    assert got == case["expected"], case["name"]
    done.append({"case": "oracle golden row " + case["name"], "hits": len(got)})
T = OO.text(0x5EED0001, 0, 1 << 16)
for m, k in ((5, 1), (64, 2), (100, 3), (150, 3), (300, 8)):
    nd = OO.pattern(0x5EED0001, 0x5EED0002, 1 << 16, 3, m, k)[0]
    a = OO.myers(T, nd, k)
    b = OO.sellers(T, nd, k)
    assert [(int(x["pos"]), int(x["score"])) for x in a] == [(int(x["pos"]), int(x["score"])) for x in b], (m, k)
nd = OO.pattern(0x5EED0001, 0x5EED0002, 1 << 16, 5, 32, 0)[0]
assert list(OO.shiftor(T, nd)) == list(OO.horspool(T, nd)) == list(OO.naive_exact(T, nd))
R = OO.repeat_text(0x5EED0001, 50000, 0, 1 << 14)
assert len(R) == 1 << 14
done.append({"case": "oracle myers == sellers, shiftor == horspool == naive, generators"})
print(json.dumps({"ok": True, "cases": done}))
