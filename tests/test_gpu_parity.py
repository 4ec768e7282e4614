"""GPU parity tests: the HIP engines, called through the C ABI, against the CPU oracle (bit-exact)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hits_list(h):
    return [(int(a), int(b), int(c)) for a, b, c in zip(h["pattern"], h["pos"], h["score"])]


def _oracle_multi(O, algo, text, needles, ks):
    """Oracle hits for a needle set, (pattern, pos, score) sorted -- one pass per needle like the reference."""
    out = []
    for p, nd in enumerate(needles):
        if len(nd) == 0:
            continue
        if algo == "myers":
            r = O.myers(text, nd, ks[p])
            out += [(p, int(a), int(s)) for a, s in zip(r["pos"], r["score"])]
        else:
            r = O.shiftor(text, nd)
            out += [(p, int(a), 0) for a in r]
    return sorted(out)


def test_reference_golden_vectors(spm, ctx, oracle):
    """Every known-answer row the reference's tests hold for this path (tests/golden/reference_vectors.json)."""
    G = json.load(open(os.path.join(GOLD, "reference_vectors.json")))
    H = oracle.encode(G["haystack"])
    P = oracle.encode(G["needle"])
    text = ctx.upload(H)
    for case in G["cases"]:
        algo = {"horspool": spm.ALGO_HORSPOOL, "shiftor": spm.ALGO_SHIFTOR}.get(case["matcher"], spm.ALGO_MYERS)
        ps = ctx.patterns(algo, [P], k=case["k"])
        assert ps.window_size(0) == case["window_size"]
        for engine in (spm.ENGINE_BRUTE, spm.ENGINE_AUTO):
            if "chunk_size" in case:
                # restorable matcher: restore(state) before / capture() after each chunk
                # (test/api/libspm/matcher/myers_matcher_restorable_test.cpp:55-74)
                state = ps.initial_state()
                got = []
                cs = case["chunk_size"]
                for off in range(0, len(H), cs):
                    chunk = ctx.upload(H[off:off + cs])
                    hits, state = spm.scan(ctx, chunk, ps, state_in=state, want_state=True, pos_offset=off)
                    v = hits.view()
                    got += list(zip(v["pos"].tolist(), v["score"].tolist()))
            else:
                v = spm.scan(ctx, text, ps, engine=engine).view()
                got = list(zip(v["pos"].tolist(), v["score"].tolist()))
            assert [g[0] for g in got] == case["expected"], (case["name"], engine)
            if "scores" in case:
                assert [g[1] for g in got] == case["scores"], (case["name"], engine)


@pytest.mark.parametrize("m", [1, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 300, 1024, 2047, 2048])
@pytest.mark.parametrize("k", [0, 1, 3])
def test_myers_brute_block_borders(spm, ctx, oracle, m, k):
    """|P| around every word border, needles of mixed length in one set, planted occurrences with edits."""
    if k >= m:
        pytest.skip("k >= |P|")
    rng = np.random.default_rng(1000 * m + k)
    n = 20000
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles, ks = [], []
    for i in range(70):  # > 64 needles: two lane groups
        mm = m if i % 3 else max(1, m - (i % 7))
        nd = rng.integers(0, 4, mm, dtype=np.uint8)
        at = int(rng.integers(0, n - 2 * mm - 8))
        occ = nd.copy()
        if i % 4 == 1 and mm > 2:
            occ = np.delete(occ, mm // 2)
        elif i % 4 == 2 and mm > 2:
            occ = np.insert(occ, mm // 3, (occ[mm // 3] + 1) & 3)
        elif i % 4 == 3:
            occ[mm // 2] = (occ[mm // 2] + 1) & 3
        T[at:at + len(occ)] = occ
        needles.append(nd)
        ks.append(min(k, mm - 1) if mm > 1 else 0)
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=ks)
    got = _hits_list(spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view())
    want = _oracle_multi(oracle, "myers", T, needles, ks)
    assert got == want


@pytest.mark.parametrize("m", [1, 5, 31, 32, 33, 64, 65, 100, 129])
def test_shiftor_brute(spm, ctx, oracle, m):
    rng = np.random.default_rng(77 + m)
    n = 30000
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = []
    for i in range(66):
        nd = rng.integers(0, 4, m, dtype=np.uint8)
        at = int(rng.integers(0, n - 3 * m - 4))
        T[at:at + m] = nd
        if i % 5 == 0 and m > 2:  # overlapping occurrences: periodic needle
            nd[:] = nd[0]
            T[at:at + 2 * m] = nd[0]
        needles.append(nd)
    text = ctx.upload(T)
    for algo in (spm.ALGO_SHIFTOR, spm.ALGO_HORSPOOL):
        ps = ctx.patterns(algo, needles)
        got = _hits_list(spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view())
        want = _oracle_multi(oracle, "shiftor", T, needles, None)
        assert got == want


def _planted_config(spm, oracle, n_total, n_pat, L, kmax, seed_text=0x5EED0001, seed_pat=0x5EED0002):
    needles = [spm.synth_pattern(seed_text, seed_pat, n_total, p, L, kmax)[0] for p in range(n_pat)]
    return needles


@pytest.mark.parametrize("cfg", [("myers", 100, 3, 256), ("myers", 150, 3, 100), ("shiftor", 32, 0, 256),
                                 ("myers", 64, 1, 64), ("myers", 1024, 10, 8),
                                 # short seeds -> keys of 12..15 symbols: q = 15 (the C5 shape), 16, 13, 12
                                 ("myers", 1024, 64, 8), ("myers", 60, 3, 100), ("myers", 64, 3, 64),
                                 ("myers", 52, 3, 64), ("myers", 48, 3, 64), ("shiftor", 13, 0, 40),
                                 # k >= 8: k + 2 seeds, candidates merged per diagonal band before verification
                                 ("myers", 300, 10, 32), ("myers", 150, 8, 64), ("myers", 400, 20, 16),
                                 ("myers", 2000, 100, 4)])
def test_filter_engine_equals_brute_and_oracle(spm, ctx, oracle, cfg):
    """Seed filter + verification must return exactly the brute-force hit set (and the oracle's)."""
    algo, L, kmax, n_pat = cfg
    n = 1 << 22
    text = ctx.generate(0x5EED0001, 0, n)
    needles = _planted_config(spm, oracle, n, n_pat, L, kmax)
    a = spm.ALGO_MYERS if algo == "myers" else spm.ALGO_SHIFTOR
    ps = ctx.patterns(a, needles, k=kmax)
    assert ps.filterable
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE)
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    assert hf.stats().engine_used == spm.ENGINE_FILTER and hf.stats().fell_back == 0
    vb, vf = hb.view(), hf.view()
    assert len(vb) >= n_pat  # every needle has a planted occurrence
    assert np.array_equal(vb, vf)
    assert hb.checksum() == hf.checksum()
    # oracle on a subset of needles over the whole text
    T = text.download(0, n)
    assert np.array_equal(T[:4096], oracle.text(0x5EED0001, 0, 4096))
    sub = list(range(0, n_pat, max(1, n_pat // 8)))
    want = _oracle_multi(oracle, algo, T, [needles[i] for i in sub], [kmax] * len(sub))
    got = [(sub.index(p), pos, s) for p, pos, s in _hits_list(vf) if p in sub]
    assert sorted(got) == want


@pytest.mark.parametrize("cfg", [(100, 3, 128, 4), (33, 1, 64, 4), (150, 3, 64, 5), (260, 10, 32, 4), (700, 30, 8, 4),
                                 (1024, 64, 8, 4), (2000, 100, 4, 4), (300, 5, 16, 5)])
def test_wave_per_candidate_verification(spm, ctx, oracle, cfg):
    """The systolic verification kernel (one lane per 32-row block) forced for every needle length: group sizes 8, 16,
    32 and 64 lanes, dna4 and dna5, raw and merged candidates.  Same hits as the brute engine and as the lane-per-
    candidate kernel."""
    L, kmax, n_pat, sigma = cfg
    n = 1 << 21
    rng = np.random.default_rng(L * 7 + kmax)
    T = rng.integers(0, 4, n, dtype=np.uint8)
    if sigma == 5:
        T[T == 3] = 4                                   # dna5 ranks: A C G N T
        T[rng.integers(0, n, 200)] = 3
    needles = []
    for i in range(n_pat):
        o = int(rng.integers(0, n - 2 * L)) if i > 1 else (0 if i == 0 else n - L - kmax)
        nd = list(T[o:o + L + kmax])
        for j in range(int(rng.integers(0, kmax + 1))):
            pos = int(rng.integers(0, L))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                nd[pos] = int(rng.choice([0, 1, 2, 4] if sigma == 5 else [0, 1, 2, 3]))
            elif kind == 1:
                del nd[pos]
            else:
                nd.insert(pos, int(rng.choice([0, 1, 2, 4] if sigma == 5 else [0, 1, 2, 3])))
        nd = np.array(nd[:L], dtype=np.uint8)
        if sigma == 5:
            nd[nd == 3] = 0                             # needles with an N are not filterable
        needles.append(nd)
    text = ctx.upload(T, sigma=sigma)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=kmax, sigma=sigma)
    assert ps.filterable
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view()
    os.environ["SPM_HIP_VERIFY_WAVE_MIN_WORDS"] = "1"
    try:
        hw = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22)
        assert hw.stats().fell_back == 0
        hw = hw.view()
        part = spm.scan(ctx, text, ps, 100001, n - 77, engine=spm.ENGINE_FILTER, left_context=True, max_hits=1 << 22).view()
    finally:
        del os.environ["SPM_HIP_VERIFY_WAVE_MIN_WORDS"]
    os.environ["SPM_HIP_VERIFY_WAVE_MIN_WORDS"] = "0"
    try:
        hl = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22).view()
    finally:
        del os.environ["SPM_HIP_VERIFY_WAVE_MIN_WORDS"]
    assert len(hb) >= n_pat // 2 and np.array_equal(hw, hb) and np.array_equal(hl, hb)
    pb = spm.scan(ctx, text, ps, 100001, n - 77, engine=spm.ENGINE_BRUTE, left_context=True, max_hits=1 << 22).view()
    assert np.array_equal(part, pb)
    want = _oracle_multi(oracle, "myers", T, needles[:3], [kmax] * 3) if sigma == 4 else None
    if want is not None:
        assert [h for h in _hits_list(hw) if h[0] < 3] == want


def test_candidate_merging_with_clustered_indels(spm, ctx, oracle):
    """k >= 8: occurrences whose k edits are insertions / deletions bunched together (the intact seeds sit on diagonals
    up to k apart), occurrences at the very start and end of the haystack, a tandem repeat (many seed hits per band)
    and sub-ranges with left context.  Filter (merged bands) == brute == oracle; fewer bands than candidates."""
    rng = np.random.default_rng(99)
    n, L, k = 1 << 20, 360, 12
    T = rng.integers(0, 4, n, dtype=np.uint8)
    unit = rng.integers(0, 4, 45, dtype=np.uint8)
    T[500000:500000 + 45 * 40] = np.tile(unit, 40)
    needles = []
    for i in range(24):
        if i == 0:
            src = T[:L + k].copy()                      # occurrence at the haystack start
        elif i == 1:
            src = T[n - L - k:].copy()                  # ... and at its end
        elif i == 2:
            src = T[500000 + 7:500000 + 7 + L + k].copy()   # inside the repeat
        else:
            o = int(rng.integers(1000, n - 2 * L))
            src = T[o:o + L + k].copy()
        nd = list(src)
        e = int(rng.integers(0, k + 1))
        at = int(rng.integers(10, L - 40))
        for j in range(e):                              # bunched edits: all within ~30 symbols
            kind = (i + j) % 3
            pos = at + int(rng.integers(0, 30))
            if kind == 0:
                nd[pos] = (nd[pos] + 1) & 3
            elif kind == 1:
                del nd[pos]
            else:
                nd.insert(pos, int(rng.integers(0, 4)))
        needles.append(np.array(nd[:L], dtype=np.uint8))
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22)
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22)
    st = hf.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
    assert 0 < st.n_bands < st.n_candidates
    assert np.array_equal(hf.view(), hb.view())
    assert len(set(hb.view()["pattern"].tolist())) == len(needles)
    want = _oracle_multi(oracle, "myers", T, needles[:6], [k] * 6)
    assert [h for h in _hits_list(hf.view()) if h[0] < 6] == want
    # sub-range with left context, and a cold-started sub-range
    for lo, hi, lc in ((123457, 700001, True), (499000, 503000, False)):
        a = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=lc, max_hits=1 << 22)
        b = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=lc, max_hits=1 << 22)
        assert np.array_equal(a.view(), b.view())
    # merging switched off gives the same hits (k + 2 seeds, every candidate verified on its own)
    os.environ["SPM_HIP_FILTER_MERGE"] = "0"
    try:
        hn = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22)
        assert np.array_equal(hn.view(), hb.view())
    finally:
        del os.environ["SPM_HIP_FILTER_MERGE"]


def test_left_context_sharding_equals_whole_scan(spm, ctx, oracle):
    """Shard rule of SURVEY 8(e): shards with left context reproduce the whole-text hit list exactly."""
    n = 1 << 20
    text = ctx.generate(0x5EED0001, 0, n)
    needles = _planted_config(spm, oracle, n, 128, 100, 3)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
    for engine in (spm.ENGINE_BRUTE, spm.ENGINE_FILTER):
        whole = spm.scan(ctx, text, ps, engine=engine).view()
        parts = []
        cuts = [0, 1000, 333333, 700001, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            parts.append(spm.scan(ctx, text, ps, a, b, engine=engine, left_context=True).view())
        merged = np.sort(np.concatenate(parts), order=["pattern", "pos"])
        assert np.array_equal(merged, whole)


def test_restorable_state_chunks_random(spm, ctx, oracle):
    """capture()/restore() across ragged chunks == one sequential scan (Myers |P|=100 and Shift-Or |P|=40)."""
    rng = np.random.default_rng(5)
    n = 5000
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = [rng.integers(0, 4, 100, dtype=np.uint8) for _ in range(3)]
    for i, nd in enumerate(needles):
        T[700 * (i + 1):700 * (i + 1) + 100] = nd
    cuts = [0, 13, 14, 700, 790, 1405, 3000, n]
    for algo, k in ((spm.ALGO_MYERS, 3), (spm.ALGO_SHIFTOR, 0)):
        nds = needles if algo == spm.ALGO_MYERS else [nd[:40] for nd in needles]
        ps = ctx.patterns(algo, nds, k=k)
        whole = spm.scan(ctx, ctx.upload(T), ps, engine=spm.ENGINE_BRUTE).view()
        state = ps.initial_state()
        parts = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            hits, state = spm.scan(ctx, ctx.upload(T[a:b]), ps, state_in=state, want_state=True, pos_offset=a)
            parts.append(hits.view())
        merged = np.sort(np.concatenate(parts), order=["pattern", "pos"])
        assert np.array_equal(merged, whole)
        assert len(whole) >= 3


def test_edge_cases(spm, ctx, oracle):
    T = oracle.encode("ACGTACGTAC")
    text = ctx.upload(T)
    # empty needle set, empty needle, needle longer than the text, empty range
    assert len(spm.scan(ctx, text, ctx.patterns(spm.ALGO_MYERS, [], k=0)).view()) == 0
    ps = ctx.patterns(spm.ALGO_MYERS, [np.zeros(0, np.uint8), oracle.encode("ACGTACGTACGTACGT")], k=[0, 1])
    assert ps.window_size(0) == 0 and ps.window_size(1) == 17
    assert len(spm.scan(ctx, text, ps).view()) == 0
    ps = ctx.patterns(spm.ALGO_SHIFTOR, [oracle.encode("AC")])
    assert spm.scan(ctx, text, ps).view()["pos"].tolist() == [0, 4, 8]
    assert len(spm.scan(ctx, text, ps, 3, 3).view()) == 0
    # sub-range without left context is a haystack of its own
    assert spm.scan(ctx, text, ps, 1, 10).view()["pos"].tolist() == [4, 8]
    # dna5 haystack (ranks A0 C1 G2 N3 T4) through the brute engine
    T5 = oracle.encode("ACGNTACGNTAACGT", 5)
    P5 = oracle.encode("ACGNT", 5)
    ps5 = ctx.patterns(spm.ALGO_MYERS, [P5], k=1, sigma=5)
    got = spm.scan(ctx, ctx.upload(T5, sigma=5), ps5).view()
    want = oracle.myers(T5, P5, 1, sigma=5)
    assert got["pos"].tolist() == want["pos"].tolist() and got["score"].tolist() == want["score"].tolist()


def test_prefix_matcher(spm, ctx, oracle):
    """restorable_myers_prefix_matcher: global start, scan bounded to |P|+k+1 symbols
    (matcher/myers_prefix_matcher_restorable.hpp:47-61).  PARITY UNPINNED by the reference; pinned on Sellers."""
    rng = np.random.default_rng(9)
    for m, k in ((5, 1), (40, 3), (100, 3), (130, 5)):
        P = rng.integers(0, 4, m, dtype=np.uint8)
        T = np.concatenate([np.delete(P, m // 2), rng.integers(0, 4, 50, dtype=np.uint8)])
        bound = min(len(T), m + k + 1)
        ps = ctx.patterns(spm.ALGO_MYERS_PREFIX, [P], k=k)
        got = spm.scan(ctx, ctx.upload(T), ps, 0, bound).view()
        want = oracle.sellers(T[:bound], P, k, mode=oracle.PREFIX)
        assert got["pos"].tolist() == want["pos"].tolist()
        assert got["score"].tolist() == want["score"].tolist()
        assert len(want) >= 1


@pytest.mark.parametrize("m,k", [(33, 1), (64, 3), (100, 3), (100, 40), (129, 5), (300, 10), (1024, 64)])
def test_brute_cutoff_kernel_equals_full_kernel_and_oracle(spm, ctx, oracle, m, k):
    """Ukkonen cut-off kernel (band of active 32-row words per lane) vs the full-width kernel vs the oracle, on a
    text built to make the band grow and shrink constantly: long near-matches of needle prefixes."""
    rng = np.random.default_rng(m * 131 + k)
    n = 30000
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = []
    for i in range(70):
        mm = m - (i % 5)
        nd = rng.integers(0, 4, mm, dtype=np.uint8)
        needles.append(nd)
        # prefixes of many lengths (band grows, then dies), a few full occurrences with edits
        for rep in range(6):
            L = int(rng.integers(8, mm))
            at = int(rng.integers(0, n - mm - 8))
            piece = nd[:L].copy()
            if L > 4 and rep % 2:
                piece[L // 2] = (piece[L // 2] + 1) & 3
            T[at:at + L] = piece
        at = int(rng.integers(0, n - mm - 8))
        occ = np.delete(nd, mm // 3) if i % 2 else nd
        T[at:at + len(occ)] = occ
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    os.environ["SPM_HIP_BRUTE_CUTOFF"] = "0"
    full = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view()
    os.environ["SPM_HIP_BRUTE_CUTOFF"] = "1"
    cut = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view()
    del os.environ["SPM_HIP_BRUTE_CUTOFF"]
    assert np.array_equal(full, cut)
    sub = list(range(0, 70, 9))
    want = _oracle_multi(oracle, "myers", T, [needles[i] for i in sub], [k] * len(sub))
    got = [(sub.index(p), pos, s) for p, pos, s in _hits_list(cut) if p in sub]
    assert sorted(got) == want
    if m <= 300:  # for |P| = 1024 the plantings overwrite each other in this short text
        assert len(want) >= len(sub)


def test_filter_sub_batches_equal_single_pass(spm, ctx, oracle):
    """Needle sets whose keys outgrow one LDS table are split into sub-batches (one filter pass each, one
    verification): same hits as the brute engine, whatever the split and stride."""
    n = 1 << 22
    text = ctx.generate(0x5EED0001, 0, n)
    needles = _planted_config(spm, oracle, n, 300, 150, 3)
    want = None
    for max_keys, stride in ((None, None), (4096, None), (1500, 4), (1024, 1)):
        os.environ["SPM_HIP_FILTER_DENSE"] = "0"    # (the sparse passes are the subject; the dense pass: test_gpu_dense.py)
        if max_keys:
            os.environ["SPM_HIP_FILTER_MAX_KEYS"] = str(max_keys)
        if stride:
            os.environ["SPM_HIP_FILTER_STRIDE"] = str(stride)
        try:
            ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
        finally:
            os.environ.pop("SPM_HIP_FILTER_MAX_KEYS", None)
            os.environ.pop("SPM_HIP_FILTER_STRIDE", None)
            os.environ.pop("SPM_HIP_FILTER_DENSE", None)
        assert ps.filterable
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
        st = h.stats()
        assert st.fell_back == 0
        if stride:  # forced stride: the key count exceeds the cap, so several passes are needed
            assert st.main_launches > 1
        got = h.view()
        if want is None:
            want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
            assert len(want) >= 300
        assert np.array_equal(got, want), (max_keys, stride)


@pytest.mark.parametrize("algo,L,k", [("myers", 100, 3), ("myers", 40, 2), ("shiftor", 32, 0), ("myers", 20, 1)])
def test_segmented_haystacks_equal_per_segment_scans(spm, ctx, oracle, algo, L, k):
    """spm_hip_scan_segments: many independent haystacks in one buffer, one launch == scanning each on its own.
    Needles are planted inside segments AND across segment borders (those must NOT be reported)."""
    rng = np.random.default_rng(L * 7 + k)
    lens = [0, 5, 300, 17, 4096, 1, 999, 150, 0, 70000, 33, 2500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(offs[-1])
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = [rng.integers(0, 4, L, dtype=np.uint8) for _ in range(70)]
    for i, nd in enumerate(needles):
        s = [2, 4, 6, 9, 11][i % 5]
        sb, se = int(offs[s]), int(offs[s + 1])
        if se - sb > L + 4:
            at = sb + int(rng.integers(0, se - sb - L - 2))
            T[at:at + L] = nd
        # straddle the border between segments 9 and 10
        if i % 7 == 0:
            b = int(offs[10])
            T[b - L // 2:b - L // 2 + L] = nd
    a = spm.ALGO_MYERS if algo == "myers" else spm.ALGO_SHIFTOR
    text = ctx.upload(T)
    ps = ctx.patterns(a, needles, k=k)
    want = []
    for s in range(len(lens)):
        sb, se = int(offs[s]), int(offs[s + 1])
        if se > sb:
            v = spm.scan(ctx, text, ps, sb, se, engine=spm.ENGINE_BRUTE).view()  # haystack of its own
            want.append(v)
    want = np.sort(np.concatenate(want), order=["pattern", "pos"])
    assert len(want) >= 30
    engines = [spm.ENGINE_BRUTE] + ([spm.ENGINE_FILTER] if ps.filterable else [])
    for engine in engines:
        got = spm.scan_segments(ctx, text, ps, offs, engine=engine).view()
        assert np.array_equal(got, want), engine
    # oracle on two segments
    for s in (4, 9):
        sb, se = int(offs[s]), int(offs[s + 1])
        o = _oracle_multi(oracle, algo, T[sb:se], needles[:8], [k] * 8)
        g = [(p, pos - sb, sc) for p, pos, sc in _hits_list(want) if p < 8 and sb <= pos <= se and
             (pos - (0 if algo == "myers" else 0)) >= sb]
        if algo == "myers":
            g = [(p, pos, sc) for p, pos, sc in g if 0 < pos <= se - sb]
        # positions of other segments can alias only at the borders; compare as sets restricted to this segment
        assert set(o) <= set(g)


def test_filter_overflow_falls_back_to_brute_exactly(spm, ctx, oracle):
    """Pathological input: a low-complexity text in which every sampled window is a seed match.  The candidate
    buffer fills up, the spans that met it full are scanned again by the brute-force kernel (hits deduplicated against
    what the verification already reported); the result is still exact."""
    n = 1 << 18
    T = np.zeros(n, dtype=np.uint8)                    # AAAA...
    T[5000:5100] = oracle.encode("ACGT" * 25)          # plus a little structure
    needles = [np.zeros(100, np.uint8), np.zeros(64, np.uint8), oracle.encode("ACGT" * 25)]
    needles[1][40] = 1
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=[3, 1, 3])
    assert ps.filterable
    os.environ["SPM_HIP_FILTER_CAND_CAP"] = "4096"
    try:
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_AUTO, max_hits=1 << 22)
        st = h.stats()
        got = h.view()
    finally:
        del os.environ["SPM_HIP_FILTER_CAND_CAP"]
    assert st.fell_back == 0 and st.fallback_spans > 0 and st.engine_used == spm.ENGINE_FILTER
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view()
    assert np.array_equal(got, want)
    o = _oracle_multi(oracle, "myers", T, needles, [3, 1, 3])
    assert _hits_list(got) == o
    assert len(o) > 100000


def test_filter_on_repetitive_text_without_overflow(spm, ctx, oracle):
    """Tandem repeats: many true seed matches per needle, several seeds per occurrence (dedupe set), default caps."""
    rng = np.random.default_rng(11)
    unit = rng.integers(0, 4, 37, dtype=np.uint8)
    T = np.tile(unit, 3000)
    T[rng.integers(0, len(T), 400)] ^= 1               # sprinkle substitutions
    needles = [np.tile(unit, 4)[i:i + 100].copy() for i in range(0, 37, 5)]
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_AUTO, max_hits=1 << 23)
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 23)
    assert np.array_equal(hf.view(), hb.view())
    assert len(hb.view()) > 10000
    sub = [0, 3]
    want = _oracle_multi(oracle, "myers", T, [needles[i] for i in sub], [3, 3])
    got = [(sub.index(p), pos, s) for p, pos, s in _hits_list(hf.view()) if p in sub]
    assert sorted(got) == want


def test_fuzz_engines_against_oracle(spm, ctx, oracle):
    """Randomised configurations (needle length, k, set size, alphabet, text length, sub-ranges): brute engine ==
    oracle always; filter engine == brute whenever the set admits it."""
    rng = np.random.default_rng(20260104)
    n_filter = 0
    for it in range(60):
        sigma = int(rng.choice([4, 4, 4, 5, 15]))
        m = int(rng.choice([1, 3, 8, 17, 32, 33, 50, 64, 65, 90, 100, 128, 160, 257]))
        k = int(rng.integers(0, min(m, 6)))
        n = int(rng.choice([0, 1, 7, 255, 256, 257, 1023, 4097, 20011]))
        n_needles = int(rng.choice([1, 2, 63, 64, 65, 130]))
        algo = "myers" if it % 3 else "shiftor"
        T = rng.integers(0, sigma, n, dtype=np.uint8)
        needles, ks = [], []
        for i in range(n_needles):
            mm = max(1, m - int(rng.integers(0, 3)))
            nd = rng.integers(0, sigma, mm, dtype=np.uint8)
            if n > 2 * mm + 2 and i % 2 == 0:
                at = int(rng.integers(0, n - mm))
                occ = nd.copy()
                if k and mm > 3 and i % 4 == 0:
                    occ[mm // 2] = (occ[mm // 2] + 1) % sigma
                T[at:at + mm] = occ
            needles.append(nd)
            ks.append(min(k, mm - 1))
        a = spm.ALGO_MYERS if algo == "myers" else spm.ALGO_SHIFTOR
        text = ctx.upload(T, sigma=sigma)
        ps = ctx.patterns(a, needles, k=ks if algo == "myers" else 0, sigma=sigma)
        lo = int(rng.integers(0, n + 1)) if it % 5 == 0 else 0
        hi = int(rng.integers(lo, n + 1)) if it % 5 == 0 else n
        got = _hits_list(spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view())
        sub = list(range(0, n_needles, max(1, n_needles // 6)))
        want = []
        for j, p in enumerate(sub):
            nd = needles[p]
            if algo == "myers":
                r = oracle.myers(T[lo:hi], nd, ks[p], sigma=sigma)
                want += [(p, int(x) + lo, int(s)) for x, s in zip(r["pos"], r["score"])]
            else:
                want += [(p, int(x) + lo, 0) for x in oracle.shiftor(T[lo:hi], nd, sigma)]
        assert sorted(x for x in got if x[0] in sub) == sorted(want), (it, sigma, m, k, n, n_needles, algo, lo, hi)
        if ps.filterable and hi > lo:
            n_filter += 1
            gf = _hits_list(spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, max_hits=1 << 22).view())
            assert gf == got, (it, "filter", sigma, m, k, n, n_needles, algo, lo, hi)
    assert n_filter >= 5


def test_fuzz_long_needles_many_errors(spm, ctx, oracle):
    """Randomised long needles with many errors (k + 2 seeds, band merging, wave-per-candidate verification, mixed
    lengths and k in one set, short keys, sub-ranges with and without left context): filter == brute; a sample of
    needles also against the oracle."""
    rng = np.random.default_rng(int(os.environ.get("SPM_FUZZ_SEED", "777")))     # soak: SPM_FUZZ_ITERS=500 SPM_FUZZ_SEED=..
    for it in range(int(os.environ.get("SPM_FUZZ_ITERS", "24"))):
        n = int(rng.choice([1 << 16, 1 << 18, (1 << 20) + 77]))
        T = rng.integers(0, 4, n, dtype=np.uint8)
        if it % 6 == 0:                                  # low-complexity stretch: many seed hits per band
            unit = rng.integers(0, 4, int(rng.integers(3, 40)), dtype=np.uint8)
            reps = 4000 // len(unit)
            T[n // 2:n // 2 + len(unit) * reps] = np.tile(unit, reps)
        n_pat = int(rng.choice([1, 3, 8, 20]))
        needles, ks = [], []
        for i in range(n_pat):
            m = int(rng.choice([120, 200, 256, 257, 400, 777, 1024, 1500, 2047]))
            m = min(m, n // 4)
            kmax = max(1, m // 13 - 1)                   # keeps floor(m / (k + 1)) >= 12: filterable
            k = int(rng.integers(1, kmax + 1)) if it % 4 else kmax
            o = int(rng.integers(0, n - m - k - 1)) if i else int(rng.choice([0, n - m - k - 1]))
            nd = list(T[o:o + m + k])
            for _ in range(int(rng.integers(0, k + 1))):
                pos = int(rng.integers(0, m))
                kind = int(rng.integers(0, 3))
                if kind == 0:
                    nd[pos] = int(rng.integers(0, 4))
                elif kind == 1:
                    del nd[pos]
                else:
                    nd.insert(pos, int(rng.integers(0, 4)))
            needles.append(np.array(nd[:m], dtype=np.uint8))
            ks.append(k)
        text = ctx.upload(T)
        ps = ctx.patterns(spm.ALGO_MYERS, needles, k=ks)
        assert ps.filterable, (it, [len(x) for x in needles], ks)
        lo, hi, lc = 0, n, False
        if it % 3 == 1:
            lo = int(rng.integers(1, n // 2))
            hi = int(rng.integers(lo + 1, n + 1))
            lc = bool(it % 2)
        hb = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=lc, max_hits=1 << 23).view()
        hf = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=lc, max_hits=1 << 23)
        assert hf.stats().fell_back == 0
        assert np.array_equal(hf.view(), hb), (it, n, [len(x) for x in needles], ks, lo, hi, lc)
        if it % 4 == 0 and not lc:
            p = int(rng.integers(0, n_pat))
            r = oracle.myers(T[lo:hi], needles[p], ks[p])
            want = [(p, int(x) + lo, int(sc)) for x, sc in zip(r["pos"], r["score"])]
            assert [h for h in _hits_list(hb) if h[0] == p] == want
        text.close()
        ps.close()


def test_dna5_haystack_through_the_seed_filter(spm, ctx, oracle):
    """dna5 text (ranks A0 C1 G2 N3 T4) with runs of N and isolated Ns: the seed filter drops windows containing an
    N, verification runs on 5 Peq rows.  filter == brute == oracle.  Needles with Ns take their seeds from their N-free
    stretches."""
    rng = np.random.default_rng(55)
    n = 1 << 20
    T = rng.choice(np.array([0, 1, 2, 4], dtype=np.uint8), n)
    for at in rng.integers(0, n - 3000, 40):
        T[at:at + int(rng.integers(1, 2000))] = 3          # N runs (assembly gaps)
    T[rng.integers(0, n, 2000)] = 3                          # isolated Ns
    needles = []
    for i in range(96):
        at = int(rng.integers(0, n - 400))
        nd = T[at:at + 100].copy()
        nd[nd == 3] = 0                                      # needles are N-free
        if i % 3 == 1:
            nd[50] = [0, 1, 2, 4][(int(nd[50]) + 1) % 4 if nd[50] < 3 else 0]
        if i % 3 == 2:
            nd = np.delete(nd, 30)
        needles.append(nd)
        if i % 4 == 0:                                       # plant a clean copy next to an N so windows straddle it
            p = int(rng.integers(0, n - 300))
            T[p:p + len(nd)] = nd
            T[p - 1] = 3
    text = ctx.upload(T, sigma=5)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3, sigma=5)
    assert ps.filterable
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    assert hf.stats().engine_used == spm.ENGINE_FILTER and hf.stats().fell_back == 0
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE)
    assert np.array_equal(hf.view(), hb.view())
    assert len(hb.view()) >= 96
    sub = list(range(0, 96, 12))
    want = []
    for j, p in enumerate(sub):
        r = oracle.myers(T, needles[p], 3, sigma=5)
        want += [(j, int(x), int(s)) for x, s in zip(r["pos"], r["score"])]
    got = [(sub.index(p), pos, s) for p, pos, s in _hits_list(hf.view()) if p in sub]
    assert sorted(got) == sorted(want)
    # needles WITH Ns stay on the filter: their seeds come from the N-free stretches.  An N in a needle matches an N in
    # the text (rank equality, as in the reference), so plant copies whose Ns sit on text Ns, too
    with_n = [nd.copy() for nd in needles[:24]]
    for i, nd in enumerate(with_n):
        nd[[10, 47, 80][i % 3]] = 3
        if i % 4 == 0:
            nd[55:58] = 3
        p = 50_000 + 2_000 * i
        T2 = nd.copy()
        if i % 2:
            T2[20] = [0, 1, 2, 4][(int(T2[20]) + 1) % 4 if T2[20] < 3 else 0]   # one substitution elsewhere
        T[p:p + len(T2)] = T2
    text = ctx.upload(T, sigma=5)
    pn = ctx.patterns(spm.ALGO_MYERS, with_n, k=3, sigma=5)
    assert pn.filterable
    hn = spm.scan(ctx, text, pn, engine=spm.ENGINE_FILTER)
    assert hn.stats().engine_used == spm.ENGINE_FILTER and hn.stats().fell_back == 0
    assert np.array_equal(hn.view(), spm.scan(ctx, text, pn, engine=spm.ENGINE_BRUTE).view())
    assert len(np.unique(hn.view()["pattern"])) == len(with_n)
    want = []
    for j in range(0, 24, 5):
        r = oracle.myers(T, with_n[j], 3, sigma=5)
        want += [(j, int(x), int(s)) for x, s in zip(r["pos"], r["score"])]
    assert sorted(x for x in _hits_list(hn.view()) if x[0] % 5 == 0 and x[0] < 24) == sorted(want)
    # exact matcher on dna5
    pe = ctx.patterns(spm.ALGO_SHIFTOR, [nd[:40] for nd in needles[:64]], sigma=5)
    assert pe.filterable
    assert np.array_equal(spm.scan(ctx, text, pe, engine=spm.ENGINE_FILTER).view(),
                          spm.scan(ctx, text, pe, engine=spm.ENGINE_BRUTE).view())


@pytest.mark.parametrize("cfg", [("myers", 100, 3, 256), ("myers", 150, 3, 64), ("shiftor", 32, 0, 200),
                                 ("myers", 64, 3, 64), ("myers", 40, 1, 32)])
def test_packed_text_shadow_gives_identical_hits(spm, ctx, oracle, cfg):
    """spm_hip_text_pack: seed-filter scans over the 2-bit shadow == scans over the 1-byte text (== brute), for every
    stride the filter picks, whole text, ragged sub-ranges with left context, and segmented haystacks."""
    algo, L, kmax, n_pat = cfg
    n = (1 << 22) + 12345
    text = ctx.generate(0x5EED0001, 0, n)
    needles = _planted_config(spm, oracle, n, n_pat, L, kmax)
    a = spm.ALGO_MYERS if algo == "myers" else spm.ALGO_SHIFTOR
    ps = ctx.patterns(a, needles, k=kmax)
    assert ps.filterable and not text.packed
    plain = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER).view()
    brute = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert np.array_equal(plain, brute)
    text.pack()
    assert text.packed
    packed = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER).view()
    assert np.array_equal(packed, plain)
    again = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.capi.SCAN_IGNORE_PACKED).view()
    assert np.array_equal(again, plain)
    for lo, hi in ((0, 5000), (4097, 70001), (n - 9000, n), (123457, 123457 + 4096)):
        want = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=True).view()
        got = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=True).view()
        assert np.array_equal(got, want), (lo, hi)
        want = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE).view()
        got = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER).view()
        assert np.array_equal(got, want), (lo, hi, "own haystack")
    offs = np.array([0, 10, 5000, 5000, 300000, 300100, n], dtype=np.uint64)
    assert np.array_equal(spm.scan_segments(ctx, text, ps, offs, engine=spm.ENGINE_FILTER).view(),
                          spm.scan_segments(ctx, text, ps, offs, engine=spm.ENGINE_BRUTE).view())


def test_pack_rejects_non_dna4(spm, ctx, oracle):
    t5 = ctx.upload(np.array([0, 1, 2, 4, 3] * 100, dtype=np.uint8), sigma=5)
    with pytest.raises(spm.SpmError):
        t5.pack()
    import torch
    # a borrowed buffer is validated when it is wrapped (a stray byte would corrupt the filter's 2-bit packing)
    bad = torch.zeros((1 << 20,), dtype=torch.uint8, device="cuda")
    ok = ctx.wrap(bad.data_ptr(), bad.numel(), sigma=4, keepalive=bad)
    assert len(ok) == bad.numel()
    bad[777_777] = 7
    torch.cuda.synchronize()
    with pytest.raises(spm.SpmError, match="not ranks"):
        ctx.wrap(bad.data_ptr(), bad.numel(), sigma=4, keepalive=bad)
    with pytest.raises(spm.SpmError, match="not ranks"):
        ctx.wrap(bad.data_ptr(), bad.numel(), sigma=5, keepalive=bad)
    assert len(ctx.wrap(bad.data_ptr(), bad.numel(), sigma=15, keepalive=bad)) == bad.numel()


def test_c_abi_error_paths(spm, ctx, oracle):
    """Bad arguments come back as error codes with a message -- nothing throws across the ABI, nothing is silent."""
    import ctypes as C
    L = spm.capi.lib()
    text = ctx.upload(oracle.encode("ACGTACGT"))
    ps = ctx.patterns(spm.ALGO_MYERS, [oracle.encode("ACG")], k=1)
    h = C.c_void_p()
    assert L.spm_hip_scan(ctx._h, text._h, 5, 3, ps._h, None, None, None, C.byref(h)) == -1          # begin > end
    assert L.spm_hip_scan(ctx._h, text._h, 0, 99, ps._h, None, None, None, C.byref(h)) == -1         # end > n
    assert b"invalid" in L.spm_hip_last_error(ctx._h)
    with pytest.raises(spm.SpmError):                                                                # rank >= sigma
        ctx.upload(np.array([0, 1, 7], dtype=np.uint8), sigma=4)
    with pytest.raises(spm.SpmError):                                                                # needle too long
        ctx.patterns(spm.ALGO_MYERS, [np.zeros(spm.capi.MAX_NEEDLE + 1, np.uint8)], k=0)
    p5 = ctx.patterns(spm.ALGO_MYERS, [oracle.encode("ACG", 5)], k=0, sigma=5)
    with pytest.raises(spm.SpmError):                                                                # sigma mismatch
        spm.scan(ctx, text, p5)
    short = ctx.patterns(spm.ALGO_MYERS, [oracle.encode("ACGTA")], k=1)                              # q < 16
    assert not short.filterable
    with pytest.raises(spm.SpmError):
        spm.scan(ctx, text, short, engine=spm.ENGINE_FILTER)
    big = ctx.patterns(spm.ALGO_MYERS, [oracle.encode("A")], k=0)
    many = ctx.upload(np.zeros(100000, np.uint8))
    hh = spm.scan(ctx, many, big, max_hits=10)                                                       # hit overflow
    with pytest.raises(spm.SpmError, match="raise spm_scan_opts.max_hits"):
        hh.view()
    with pytest.raises(spm.SpmError):                                                                # unaligned wrap
        ctx.wrap(text.device_ptr + 1, 4)


def test_c1_horspool_plumbing(spm, ctx, oracle):
    """BASELINE.json configs[0]: Horspool exact, 1 needle |P|=32 over 1 MiB random DNA -- the CPU restatement
    (Horspool with skip table) against naive search, and the device's exact engine through SPM_ALGO_HORSPOOL."""
    n = 1 << 20
    T = oracle.text(0x5EED0001, 0, n)
    P, at = oracle.pattern(0x5EED0001, 0x5EED0002, n, 0, 32, 0)
    want = oracle.naive_exact(T, P).tolist()
    assert at in want
    assert oracle.horspool(T, P).tolist() == want
    text = ctx.generate(0x5EED0001, 0, n)
    got = spm.scan(ctx, text, ctx.patterns(spm.ALGO_HORSPOOL, [P])).view()
    assert got["pos"].tolist() == want and set(got["score"].tolist()) == {0}


@pytest.mark.parametrize("algo,L,k", [("myers", 100, 3), ("shiftor", 40, 0), ("myers", 64, 1)])
def test_restorable_chunks_through_the_seed_filter(spm, ctx, oracle, algo, L, k):
    """Restorable matchers on the HBM-bound engine: chunks >= 1 MiB continue from the restored state
    -- the brute-force kernel takes the first window - 1 symbols of a chunk, the seed filter the rest -- and hand back a
    state.  Hits == one sequential scan == oracle; every state blob == the brute-force path's."""
    n = (3 << 20) + 12345
    T = oracle.text(0x5EED0001, 0, n)
    rng = np.random.default_rng(77)
    needles = [spm.synth_pattern(0x5EED0001, 0x5EED0002, n, p, L, k)[0] for p in range(40)]
    cuts = [0, 1 << 20, (1 << 20) + 300000, (2 << 20) + 7, n]
    for c in cuts[1:-1]:                     # occurrences straddling every chunk border, at several phases
        for j, d in enumerate((-L + 1, -L // 2, -3, 0)):
            nd = needles[4 * cuts.index(c) + j].copy()
            if k and j % 2:
                nd[L // 3] ^= 1
            T[c + d:c + d + L] = needles[4 * cuts.index(c) + j]
            needles[4 * cuts.index(c) + j] = nd
    s_algo = spm.ALGO_MYERS if algo == "myers" else spm.ALGO_SHIFTOR
    ps = ctx.patterns(s_algo, needles, k=k)
    assert ps.filterable
    whole = spm.scan(ctx, ctx.upload(T), ps, engine=spm.ENGINE_BRUTE).view()
    want = _oracle_multi(oracle, algo, T, needles, [k] * len(needles))
    assert _hits_list(whole) == want
    sf, sb = ps.initial_state(), ps.initial_state()
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        chunk = ctx.upload(T[a:b])
        hf, sf = spm.scan(ctx, chunk, ps, state_in=sf, want_state=True, pos_offset=a, engine=spm.ENGINE_AUTO)
        hb, sb = spm.scan(ctx, chunk, ps, state_in=sb, want_state=True, pos_offset=a, engine=spm.ENGINE_BRUTE)
        assert hf.stats().engine_used == spm.ENGINE_FILTER and hb.stats().engine_used == spm.ENGINE_BRUTE
        assert np.array_equal(sf, sb), "state after the chunk differs between the engines"
        assert np.array_equal(hf.view(), hb.view())
        parts.append(hf.view())
    merged = np.sort(np.concatenate(parts), order=["pattern", "pos"])
    assert np.array_equal(merged, whole)
    # want_state without a state to continue from: a fresh matcher's first chunk
    h0, s0 = spm.scan(ctx, ctx.upload(T[:cuts[1]]), ps, want_state=True, engine=spm.ENGINE_FILTER)
    hb0, sb0 = spm.scan(ctx, ctx.upload(T[:cuts[1]]), ps, want_state=True, engine=spm.ENGINE_BRUTE)
    assert h0.stats().engine_used == spm.ENGINE_FILTER
    assert np.array_equal(s0, sb0) and np.array_equal(h0.view(), hb0.view())


def test_dna15_haystack_through_the_seed_filter(spm, ctx, oracle):
    """dna15 (seqan3 ranks A0 B1 C2 D3 G4 H5 K6 M7 N8 R9 S10 T11 V12 W13 Y14): A, C, G, T are the key symbols, every
    ambiguity code -- in the text or in a needle -- is treated like dna5's N (rank equality decides a match, as in the
    reference).  filter == brute == oracle."""
    rng = np.random.default_rng(1515)
    n = 1 << 20
    T = np.array([0, 2, 4, 11], dtype=np.uint8)[rng.integers(0, 4, n)]
    amb = np.array([1, 3, 5, 6, 7, 8, 9, 10, 12, 13, 14], dtype=np.uint8)
    T[rng.integers(0, n, 3000)] = amb[rng.integers(0, len(amb), 3000)]   # isolated ambiguity codes
    for at in rng.integers(0, n - 2000, 20):
        T[at:at + int(rng.integers(1, 1500))] = 8                         # runs of N
    needles = []
    for i in range(80):
        at = int(rng.integers(0, n - 400))
        nd = T[at:at + 100].copy()
        if i % 2 == 0:
            nd[np.isin(nd, amb)] = 0                                      # half of the needles are ACGT only
        if i % 3 == 1:
            nd[60] = 2 if nd[60] != 2 else 4
        if i % 3 == 2:
            nd = np.delete(nd, 25)
        needles.append(nd)
        p = 10_000 + 5_000 * i
        T[p:p + len(nd)] = nd
    text = ctx.upload(T, sigma=15)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3, sigma=15)
    assert ps.filterable
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    assert hf.stats().engine_used == spm.ENGINE_FILTER and hf.stats().fell_back == 0
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE)
    assert np.array_equal(hf.view(), hb.view())
    assert len(np.unique(hf.view()["pattern"])) == len(needles)
    want = []
    for j in range(0, 80, 9):
        r = oracle.myers(T, needles[j], 3, sigma=15)
        want += [(j, int(x), int(s)) for x, s in zip(r["pos"], r["score"])]
    assert sorted(x for x in _hits_list(hf.view()) if x[0] % 9 == 0) == sorted(want)
    pe = ctx.patterns(spm.ALGO_SHIFTOR, [nd[:40] for nd in needles[::2]], sigma=15)
    assert pe.filterable
    assert np.array_equal(spm.scan(ctx, text, pe, engine=spm.ENGINE_FILTER).view(),
                          spm.scan(ctx, text, pe, engine=spm.ENGINE_BRUTE).view())


@pytest.mark.parametrize("m,k", [(32, 2), (44, 3), (27, 2), (30, 1), (20, 1)])
def test_short_seeds_through_the_seed_filter(spm, ctx, oracle, m, k):
    """Seeds of 9 .. 15 symbols (|P| = 32, k = 2 and the like): the whole seed is the key, thousands of chance matches
    per megabase are resolved after the streaming pass.  filter == brute == oracle."""
    rng = np.random.default_rng(100 * m + k)
    n = 1 << 21
    T = oracle.text(0x5EED0001, 0, n)
    needles = []
    for i in range(200):
        at = int(rng.integers(0, n - 2 * m))
        nd = T[at:at + m].copy()
        for e in range(i % (k + 1)):
            nd[(7 * e + 3 + i) % m] ^= 1 + (i & 1)
        needles.append(nd)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    text = ctx.upload(T)
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22)
    st = hf.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0 and st.fallback_spans == 0
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22)
    assert np.array_equal(hf.view(), hb.view())
    assert len(np.unique(hf.view()["pattern"])) == len(needles)
    sub = list(range(0, 200, 23))
    want = _oracle_multi(oracle, "myers", T, [needles[i] for i in sub], [k] * len(sub))
    got = sorted((sub.index(p), pos, s) for p, pos, s in _hits_list(hf.view()) if p in sub)
    assert got == want


def test_native_rccl_gatherv_world_of_one(spm, ctx, oracle):
    """The C-ABI exchange entry (spm_hip_comm_* / spm_hip_gatherv_hits): librccl.so is opened on demand, a communicator
    of one rank is made, and the gatherv hands back this rank's records unchanged -- the count all-gather and the root's
    own copy run; the send/recv legs need a multi-GPU box (the offsets they use are checked on the CPU)."""
    import ctypes as C
    L = spm.capi.lib()
    uid = (C.c_char * 128)()
    assert L.spm_hip_comm_unique_id(uid) == 0
    comm = C.c_void_p()
    rc = L.spm_hip_comm_init(ctx._h, uid, 0, 1, C.byref(comm))
    assert rc == 0, L.spm_hip_last_error(ctx._h)
    try:
        n = 1 << 20
        text = ctx.generate(0x5EED0001, 0, n)
        needles = [spm.synth_pattern(0x5EED0001, 0x5EED0002, n, p, 100, 3)[0] for p in range(64)]
        ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
        h = spm.scan(ctx, text, ps)
        want = h.view()
        rec, tot = C.c_void_p(), C.c_uint64()
        counts = (C.c_uint64 * 1)()
        assert L.spm_hip_gatherv_hits(comm, h._h, 0, C.byref(rec), C.byref(tot), counts) == 0
        assert tot.value == len(want) == counts[0] and len(want) >= 64
        got = np.empty(tot.value, dtype=spm.HIT_DTYPE)
        buf = (C.c_char * (16 * tot.value)).from_buffer(got)
        hip = C.CDLL("libamdhip64.so")
        assert hip.hipMemcpy(buf, rec, 16 * tot.value, 2) == 0          # hipMemcpyDeviceToHost
        assert np.array_equal(np.sort(got, order=["pattern", "pos"]), want)
        assert L.spm_hip_gatherv_hits(comm, h._h, 3, C.byref(rec), C.byref(tot), counts) == -1   # no such root
    finally:
        L.spm_hip_comm_destroy(comm)


def test_hits_copy_fused_and_copy_to(spm, ctx, oracle):
    """spm_hip_hits_copy_device / _copy_fused: the records (and the [count | records] layout the fused all-gather ships)
    land in a caller-owned device buffer, truncated to its capacity, the count always the true one."""
    import torch
    rng = np.random.default_rng(5)
    T = rng.integers(0, 4, 1 << 20, dtype=np.uint8)
    needles = [T[o:o + 64].copy() for o in rng.integers(0, (1 << 20) - 64, 200)]
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=1)
    h = spm.scan(ctx, text, ps)
    want = h.view()
    n_true = len(want)
    assert n_true >= 600                                   # every needle: its end position and the two next to it
    key = lambda a: np.sort(a.view(np.uint8).reshape(-1, 16), axis=0).tobytes()  # (arrival order is not defined)
    for cap in (n_true + 10, 50):
        buf = torch.full((cap + 1, 2), -1, dtype=torch.int64, device="cuda")
        n = h.copy_fused(buf.data_ptr(), cap)
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
        assert n == n_true and host[0, 0] == n_true and host[0, 1] == 0
        got = host[1:1 + min(n, cap)].copy().view(np.uint8).reshape(-1, 16)
        if cap >= n_true:
            assert key(got) == key(want.view(np.uint8).reshape(-1, 16))
            assert (host[1 + n_true:] == -1).all()         # nothing written past the records
        buf2 = torch.full((cap, 2), -1, dtype=torch.int64, device="cuda")
        assert h.copy_to(buf2.data_ptr(), cap) == n_true
        torch.cuda.synchronize()
        assert np.array_equal(buf2.cpu().numpy()[:min(n, cap)], host[1:1 + min(n, cap)])
