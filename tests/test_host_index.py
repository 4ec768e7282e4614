"""CPU: the host logic of the seed filter -- pigeonhole seeds, stride choice, sub-batching, the perfect-hash
fingerprint table, the Bloom fallback and the exact key table -- checked without a device through
spm_hip_host_selftest (the same code spm_hip_patterns_create runs)."""
import ctypes as C
import os

import numpy as np
import pytest


def _selftest(spm, algo, needles, k, sigma=4):
    offs = np.zeros(len(needles) + 1, dtype=np.uint32)
    offs[1:] = np.cumsum([len(x) for x in needles])
    cat = np.ascontiguousarray(np.concatenate(needles), dtype=np.uint8)
    ks = np.full(len(needles), k, dtype=np.uint16) if np.isscalar(k) else np.asarray(k, dtype=np.uint16)
    stats = (C.c_uint64 * 8)()
    rc = spm.capi.lib().spm_hip_host_selftest(algo, cat.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              offs.ctypes.data_as(C.POINTER(C.c_uint32)), len(needles),
                                              ks.ctypes.data_as(C.POINTER(C.c_uint16)), sigma, stats)
    names = ["passes", "stride", "keys", "checked", "missing", "fp", "trials", "hash_variant"]
    st = dict(zip(names, [int(x) for x in stats]))
    st["anchor_dimers"] = (st["hash_variant"] >> 8) & 0xFF
    st["deep"] = st["hash_variant"] >> 16       # dense pass: needles whose pieces overlap (c > 1)
    st["hash_variant"] &= 0xFF
    st["key_len"] = st["stride"] >> 32
    st["stride"] &= 0xFFFFFFFF
    return rc, st


def test_c3_shaped_set_single_pass_fingerprint_table(spm):
    needles = [spm.synth_pattern(0x5EED0001, 0x5EED0002, 1 << 34, p, 100, 3)[0] for p in range(1024)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, needles, 3)
    assert rc == 0 and st["missing"] == 0
    assert st["passes"] == 1 and st["stride"] == 8 and st["keys"] == 1024 * 4 * 8 == st["checked"]
    assert st["hash_variant"] == 2
    assert st["fp"] / st["trials"] < 1e-4      # ~ load * 2^-16


def test_exact_and_long_needles_and_sub_batches(spm):
    rng = np.random.default_rng(3)
    rc, st = _selftest(spm, spm.ALGO_SHIFTOR, [rng.integers(0, 4, 32, dtype=np.uint8) for _ in range(1024)], 0)
    assert rc == 0 and st["stride"] == 16 and st["passes"] == 1 and st["keys"] == 1024 * 16
    # 6 000 needles of 150, k = 3: 24 000 seeds -> the cost model trades stride for passes (stride 2, one pass)
    big = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(6000)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, big, 3)
    assert rc == 0 and st["missing"] == 0 and st["passes"] == 1 and st["stride"] == 2 and st["keys"] == 6000 * 4 * 2
    assert st["hash_variant"] == 4      # 8 windows of every 16 symbols are looked up: presence bits + L2 buckets as level 1
    # 20 000 needles: 80 000 seeds do not fit one table (57 344 keys) even at stride 1.  Sparse passes: sub-batches; the
    # cost model prefers three passes at stride 2 to two at stride 1.  (By default such a set takes the dense pass, below.)
    big20 = big + [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(14000)]
    c4 = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(100000)]
    os.environ["SPM_HIP_FILTER_DENSE"] = "0"
    try:
        rc, st = _selftest(spm, spm.ALGO_MYERS, big20, 3)
        assert rc == 0 and st["missing"] == 0 and (st["passes"], st["stride"], st["keys"]) == (3, 2, 160000)
        assert st["hash_variant"] == 4      # stride 2: presence bits (stride >= 4 and anchored passes: fingerprint table)
        # a C4-sized set: 400 000 seeds at stride 1, one key per seed.  Anchored: every seed's key begins with the dimer of
        # its pass (the streaming kernel looks up only the text windows that begin with it); a few passes take two dimers
        rc, st = _selftest(spm, spm.ALGO_MYERS, c4, 3)
        assert rc == 0 and st["missing"] == 0 and (st["stride"], st["keys"], st["checked"]) == (1, 400000, 400000)
        assert st["passes"] == 7 and st["hash_variant"] == 2 and 7 <= st["anchor_dimers"] <= 14  # (dimers over all passes)
        os.environ["SPM_HIP_FILTER_ANCHOR"] = "0"
        try:
            rc, st = _selftest(spm, spm.ALGO_MYERS, c4, 3)
        finally:
            del os.environ["SPM_HIP_FILTER_ANCHOR"]
        assert rc == 0 and st["missing"] == 0 and (st["passes"], st["stride"], st["keys"]) == (7, 1, 400000)
        assert st["hash_variant"] == 4 and st["anchor_dimers"] == 0
    finally:
        del os.environ["SPM_HIP_FILTER_DENSE"]
    # mixed lengths and k: the stride follows the shortest seed
    mixed = [rng.integers(0, 4, m, dtype=np.uint8) for m in (64, 100, 150, 300, 1000)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, mixed, [3, 3, 5, 10, 40])
    assert rc == 0 and st["passes"] == 1 and (st["key_len"], st["stride"]) == (15, 2)   # q = 64 / 4 = 16
    # needles with k >= 8 carry k + 2 seeds (two intact seeds per occurrence: candidate merging)
    assert st["keys"] == (4 + 4 + 6 + 12 + 42) * 2 and st["missing"] == 0
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 1024, dtype=np.uint8)], 64)
    assert rc == 0 and st["missing"] == 0 and st["keys"] == 66 * 2 and (st["key_len"], st["stride"]) == (14, 2)
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 100, dtype=np.uint8)], 7)       # q = 12, k + 1 seeds
    assert rc == 0 and st["missing"] == 0 and st["keys"] == 8 and (st["key_len"], st["stride"]) == (12, 1)
    # short seeds: q = 15 (|P| = 60, k = 3 and the C5 shape |P| = 1024, k = 64) -> 14-symbol keys at stride 2
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 60, dtype=np.uint8) for _ in range(100)], 3)
    assert rc == 0 and st["missing"] == 0 and st["passes"] == 1 and (st["key_len"], st["stride"]) == (14, 2)
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 1024, dtype=np.uint8) for _ in range(64)], 64)
    assert rc == 0 and st["missing"] == 0 and (st["key_len"], st["stride"]) == (14, 2)
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 48, dtype=np.uint8)], 3)      # q = 12
    assert rc == 0 and st["missing"] == 0 and (st["key_len"], st["stride"]) == (12, 1)
    # seeds shorter than the shortest key: the filter does not apply
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 32, dtype=np.uint8)], 3)
    assert rc == 0 and st["passes"] == 0
    # seeds of 9 .. 11 symbols: the whole seed is the key, stride 1
    for m, k, q in ((44, 3, 11), (32, 2, 10), (27, 2, 9)):
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, m, dtype=np.uint8) for _ in range(100)], k)
        assert rc == 0 and st["passes"] == 1 and st["missing"] == 0 and st["stride"] == 1 and st["key_len"] == q, (m, k, st)
    # ... unless there are so many of them that most text windows would match one by chance
    rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 27, dtype=np.uint8) for _ in range(20000)], 2)
    assert rc == 0 and st["passes"] == 0


def test_dna5_and_bloom_fallback(spm):
    rng = np.random.default_rng(4)
    needles = [rng.choice(np.array([0, 1, 2, 4], dtype=np.uint8), 100) for _ in range(256)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, needles, 3, sigma=5)
    assert rc == 0 and st["passes"] == 1 and st["missing"] == 0
    # needles with N: their seeds are taken from the N-free stretches (k + 1 disjoint pieces are all the pigeonhole
    # argument needs), so the set stays filterable; the self-test checks that no indexed seed holds an N
    with_n = [x.copy() for x in needles]
    with_n[7][50] = 3
    with_n[8][[0, 31, 62, 99]] = 3
    with_n[9][10:20] = 3
    rc, st = _selftest(spm, spm.ALGO_MYERS, with_n, 3, sigma=5)
    assert rc == 0 and st["passes"] == 1 and st["missing"] == 0
    with_n[10][::8] = 3                                              # no stretch of 12 key symbols left: brute engine
    rc, st = _selftest(spm, spm.ALGO_MYERS, with_n, 3, sigma=5)
    assert rc == 0 and st["passes"] == 0
    # dna15 (A0 C2 G4 T11 are the key symbols): the same, with every other code standing in for N
    d15 = [np.array([0, 2, 4, 11], dtype=np.uint8)[rng.integers(0, 4, 100)] for _ in range(64)]
    d15[3][40] = 8
    d15[4][5] = 14
    rc, st = _selftest(spm, spm.ALGO_MYERS, d15, 3, sigma=15)
    assert rc == 0 and st["passes"] == 1 and st["missing"] == 0
    os.environ["SPM_HIP_FILTER_HASH"] = "1"                          # force the Bloom cascade
    try:
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 100, dtype=np.uint8) for _ in range(1024)], 3)
    finally:
        del os.environ["SPM_HIP_FILTER_HASH"]
    assert rc == 0 and st["hash_variant"] == 1 and st["missing"] == 0
    assert st["fp"] / st["trials"] < 2e-3                            # 4 probes at 12.5 % occupancy


def test_dense_pass_index(spm):
    """Sets that would need several sparse passes (or stride 1) get ONE dense pass: every needle has c k + 1 pieces, no
    position in more than c of them, every piece holds a 16-symbol key that begins with an anchor dimer and is found at
    all three levels (presence bit, fingerprint bucket, directory entry) -- the self-check verifies exactly that."""
    rng = np.random.default_rng(5)
    c4 = [rng.integers(0, 4, 150, dtype=np.uint8) for _ in range(100000)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, c4, 3)
    assert rc == 0 and st["missing"] == 0 and st["passes"] == 1 and st["hash_variant"] == 3
    assert (st["stride"], st["key_len"]) == (1, 16) and 400000 <= st["keys"] <= 420000 and st["checked"] == st["keys"]
    assert 2 <= st["anchor_dimers"] <= 3                     # 1/8 .. 3/16 of the text windows are looked up
    assert st["fp"] / st["trials"] < 2e-4                    # a random window: anchored AND bit set AND fingerprint equal
    # the shape of the reference's own read set (100 000 x 100, k <= 3): less room per piece, more anchors
    r100 = [rng.integers(0, 4, 100, dtype=np.uint8) for _ in range(100000)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, r100, 3)
    assert rc == 0 and st["missing"] == 0 and st["passes"] == 1 and st["hash_variant"] == 3 and 3 <= st["anchor_dimers"] <= 6
    # needles whose windows offer a single dimer (poly-A, (AC)n) decide which anchors are usable at all
    odd = c4[:20000] + [np.zeros(150, np.uint8), np.resize(np.array([0, 1], np.uint8), 150),
                        np.resize(np.array([2, 3, 3], np.uint8), 150)]
    rc, st = _selftest(spm, spm.ALGO_MYERS, odd, 3)
    assert rc == 0 and st["missing"] == 0 and st["passes"] == 1 and st["hash_variant"] == 3 and st["anchor_dimers"] <= 8
    # a small set, forced: needles of exactly (k + 1) x 16 symbols have one layout only (keys at 0, 16, 32, 48): the anchor
    # set grows until it covers them -- every window, then
    os.environ["SPM_HIP_FILTER_DENSE"] = "2"
    try:
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 64, dtype=np.uint8) for _ in range(300)], 3)
        assert rc == 0 and st["missing"] == 0 and st["hash_variant"] == 3 and st["anchor_dimers"] == 16
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 100, dtype=np.uint8) for _ in range(300)], 3)
        assert rc == 0 and st["missing"] == 0 and st["hash_variant"] == 3
        # exact matchers: one piece per needle
        rc, st = _selftest(spm, spm.ALGO_SHIFTOR, [rng.integers(0, 4, 40, dtype=np.uint8) for _ in range(300)], 0)
        assert rc == 0 and st["missing"] == 0 and st["hash_variant"] == 3 and st["keys"] == 300
    finally:
        del os.environ["SPM_HIP_FILTER_DENSE"]
    # not for sets the dense pass cannot hold: needles shorter than (k + 1) x 16, sets with k >= 8
    os.environ["SPM_HIP_FILTER_DENSE"] = "2"
    try:
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 60, dtype=np.uint8) for _ in range(100)], 3)
        assert rc == 0 and st["hash_variant"] == 4 and st["stride"] == 2     # (a sparse pass; stride 2: presence bits)
        rc, st = _selftest(spm, spm.ALGO_MYERS, [rng.integers(0, 4, 1024, dtype=np.uint8) for _ in range(8)], 64)
        assert rc == 0 and st["hash_variant"] == 4 and st["stride"] == 2
    finally:
        del os.environ["SPM_HIP_FILTER_DENSE"]
