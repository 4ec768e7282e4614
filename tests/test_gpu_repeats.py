"""Repeat-rich texts: the seed filter must stay exact AND stay the engine -- a span whose seed hits exceed its budget
is scanned again by the brute-force kernel on its own, the rest of the scan keeps the filter's result
(VERDICT r01 "what's weak" 1).  Everything is checked against the CPU oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED_TEXT, SEED_PAT = 0x5EED0001, 0x5EED0002


def _hits_list(h):
    return [(int(a), int(b), int(c)) for a, b, c in zip(h["pattern"], h["pos"], h["score"])]


def _oracle(O, T, needles, k):
    out = []
    for p, nd in enumerate(needles):
        r = O.myers(T, nd, k)
        out += [(p, int(a), int(s)) for a, s in zip(r["pos"], r["score"])]
    return sorted(out)


def _setup(spm, ctx, oracle, n, ppm, n_pat, L=100, k=3):
    text = ctx.generate_repeats(SEED_TEXT, 0, n, ppm)
    T = text.download(0, n)
    assert np.array_equal(T, oracle.repeat_text(SEED_TEXT, ppm, 0, n)), "device generator != oracle generator"
    needles = [spm.synth_repeat_pattern(SEED_TEXT, SEED_PAT, n, p, L, k, ppm)[0] for p in range(n_pat)]
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    return text, T, needles, ps


@pytest.mark.parametrize("ppm", [10000, 50000, 120000])
def test_repeat_rich_text_filter_equals_brute_and_oracle(spm, ctx, oracle, ppm):
    """c3r-shaped: needles cut across tandem-repeat / low-complexity stretches; default budgets -> no fallback at all."""
    n = 1 << 22
    text, T, needles, ps = _setup(spm, ctx, oracle, n, ppm, 64)
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
    st = hf.stats()
    hb = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 24)
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
    assert np.array_equal(hf.view(), hb.view())
    assert _hits_list(hf.view()) == _oracle(oracle, T, needles, 3)
    assert len(np.unique(hf.view()["pattern"])) == len(needles)


@pytest.mark.parametrize("budget", [1, 3, 8])
def test_partial_fallback_is_exact(spm, ctx, oracle, budget):
    """A tiny per-span budget forces spans to give up; only those are brute-scanned, hits stay bit-exact and unique."""
    n = 1 << 22
    text, T, needles, ps = _setup(spm, ctx, oracle, n, 50000, 64)
    os.environ["SPM_HIP_FILTER_SPAN_BUDGET"] = str(budget)
    try:
        hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
        st = hf.stats()
        got = hf.view()
    finally:
        del os.environ["SPM_HIP_FILTER_SPAN_BUDGET"]
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
    assert st.fallback_spans > 0 and 0 < st.fallback_symbols
    if budget >= 3:
        assert st.fallback_symbols < n  # partial: the quiet spans kept the filter's result
    want = _oracle(oracle, T, needles, 3)
    assert _hits_list(got) == want
    assert len(set(_hits_list(got))) == len(got)  # nothing reported twice (filter + re-scan share the dedupe set)


def test_partial_fallback_with_left_context_and_offset(spm, ctx, oracle):
    """Shard semantics survive the fallback: sub-range with left context, pos_offset, hits owned by last symbol."""
    n = 1 << 21
    text, T, needles, ps = _setup(spm, ctx, oracle, n, 80000, 32)
    b, e = 300001, n - 7777
    os.environ["SPM_HIP_FILTER_SPAN_BUDGET"] = "2"
    try:
        hf = spm.scan(ctx, text, ps, b, e, engine=spm.ENGINE_FILTER, left_context=True, pos_offset=1000, max_hits=1 << 24)
        st = hf.stats()
        got = _hits_list(hf.view())
    finally:
        del os.environ["SPM_HIP_FILTER_SPAN_BUDGET"]
    assert st.fallback_spans > 0
    want = [(p, pos + 1000, s) for p, pos, s in _oracle(oracle, T, needles, 3) if b < pos <= e]
    assert got == want


def test_periodic_needles_merged_index_entries(spm, ctx, oracle):
    """Homopolymer / short-period needles: every shift of every seed has the same key.  The index keeps one entry per
    (key, needle) with a diagonal range; verification covers all the diagonals it stands for."""
    rng = np.random.default_rng(5)
    n = 1 << 20
    T = rng.integers(0, 4, n, dtype=np.uint8)
    for at in range(5000, n - 1000, 9973):                 # runs of A, (CA)n, (GAT)n of varying length
        ln = 60 + (at % 240)
        unit = [[0], [1, 0], [2, 0, 3]][at % 3]
        T[at:at + ln] = np.resize(np.array(unit, np.uint8), ln)
    T[rng.integers(0, n, 2000)] ^= 2                       # impurities
    needles = [np.zeros(100, np.uint8), np.resize(np.array([1, 0], np.uint8), 100),
               np.resize(np.array([2, 0, 3], np.uint8), 100),
               np.concatenate([rng.integers(0, 4, 60, dtype=np.uint8), np.zeros(40, np.uint8)]),  # unique + poly-A tail
               np.concatenate([np.resize(np.array([0, 1], np.uint8), 50), rng.integers(0, 4, 50, dtype=np.uint8)])]
    for nd in needles[3:]:
        T[777777:777777 + 100] = nd                       # (the second overwrites the first: one planted at least)
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
    hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
    st = hf.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
    want = _oracle(oracle, T, needles, 3)
    assert _hits_list(hf.view()) == want
    assert len(want) > 1000
    # the same with the merged entries switched off: more candidates, same hits
    os.environ["SPM_HIP_FILTER_DEDUPE"] = "0"
    try:
        ps0 = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
    finally:
        del os.environ["SPM_HIP_FILTER_DEDUPE"]
    h0 = spm.scan(ctx, text, ps0, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
    assert np.array_equal(h0.view(), hf.view())
    assert h0.stats().n_candidates > 4 * st.n_candidates


def test_hit_overflow_is_reported_not_rescanned(spm, ctx, oracle):
    """More hits than max_hits: SPM_E_OVERFLOW from the view with the count so far, no brute-force re-run of the scan."""
    n = 1 << 20
    T = np.zeros(n, dtype=np.uint8)
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, [np.zeros(100, np.uint8)], k=3)
    h = spm.scan(ctx, text, ps, engine=spm.ENGINE_AUTO, max_hits=1000)
    st = h.stats()
    assert st.n_hits > 1000 and st.fell_back == 0
    with pytest.raises(spm.SpmError, match="raise spm_scan_opts.max_hits"):
        h.view()
    h2 = spm.scan(ctx, text, ps, engine=spm.ENGINE_AUTO, max_hits=1 << 21)
    assert len(h2.view()) == n - 100 + 3 + 1


@pytest.mark.parametrize("algo", ["shiftor", "horspool"])
def test_exact_sets_with_and_without_repeat_needles(spm, ctx, oracle, algo):
    """Exact sets (k = 0, every needle its own single seed): the resolve kernel reports the hits itself -- unless a needle
    IS a repeat (its index entries are merged into a diagonal range and skip the whole-seed check), then the set goes
    through bands and verification.  Either way: == the brute-force engine == the naive definition."""
    rng = np.random.default_rng(17)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    T[1000:1400] = 0                                               # poly-A
    T[9000:9600] = np.resize(np.array([0, 1], np.uint8), 600)      # (AC)n
    L = 32
    plain = [T[o:o + L].copy() for o in rng.integers(20000, n - L, 300)]
    repeats = [np.zeros(L, np.uint8), np.resize(np.array([0, 1], np.uint8), L), np.resize(np.array([1, 0], np.uint8), L)]
    a = spm.ALGO_SHIFTOR if algo == "shiftor" else spm.ALGO_HORSPOOL
    text = ctx.upload(T)
    for needles, with_runs in ((plain, False), (plain + repeats, True)):
        ps = ctx.patterns(a, needles, k=0)
        assert ps.filterable
        hf = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 20)
        st = hf.stats()
        assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
        assert (st.n_bands > 0) == with_runs          # no bands at all when the resolve kernel reports the hits
        got = hf.view()
        want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 20).view()
        assert np.array_equal(got, want) and len(want) >= 300
        # the naive definition for a few needles (exact matchers report the begin position)
        for p in (0, 7, len(needles) - 1):
            nd = needles[p]
            win = np.lib.stride_tricks.sliding_window_view(T, L)
            begins = np.flatnonzero((win == nd).all(axis=1))
            mine = np.sort(want["pos"][want["pattern"] == p])
            assert np.array_equal(mine, begins.astype(np.uint64))
        # a sub-range with left context and an offset: ownership by the last symbol, reported by the resolve kernel too
        lo, hi = 500, 12000
        h1 = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=True, pos_offset=77).view()
        h2 = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=True, pos_offset=77).view()
        assert np.array_equal(h1, h2)


@pytest.mark.parametrize("budget", [2, 24])
def test_segmented_scan_falls_back_span_by_span(spm, ctx, oracle, budget):
    """Independent haystacks stored back to back (spm_hip_scan_segments; the journaled-sequence search is one such scan):
    spans whose survivors exceed a tiny budget give up and are scanned again by the brute-force kernel, with tiles that follow
    the segment table -- not the whole scan.  Hits == per-segment brute-force scans == oracle; nothing spans two haystacks."""
    rng = np.random.default_rng(31 + budget)
    lens = [0, 7, 5000, 150, 149, 70000, 1, 33, 260000, 0, 2500, 1 << 20, 99, 400000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    n = int(offs[-1])
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 100, 3
    needles = []
    for i in range(96):
        s = int(rng.integers(0, len(lens)))
        while lens[s] < 2 * L:
            s = int(rng.integers(0, len(lens)))
        at = int(offs[s]) + int(rng.integers(0, lens[s] - L))
        nd = T[at:at + L].copy()
        for e in range(i % (k + 1)):
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
        needles.append(nd)
    for j, s in enumerate((2, 5, 8)):       # occurrences across a border between two haystacks: must NOT be reported
        b = int(offs[s + 1])
        T[b - 50:b + 50] = needles[j]
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    ref = []
    for s in range(len(lens)):
        b, e = int(offs[s]), int(offs[s + 1])
        if e > b:
            ref.append(spm.scan(ctx, text, ps, b, e, engine=spm.ENGINE_BRUTE).view())
    ref = np.concatenate(ref)
    ref = ref[np.lexsort((ref["pos"], ref["pattern"]))]
    os.environ["SPM_HIP_FILTER_SPAN_BUDGET"] = str(budget)
    try:
        h = spm.scan_segments(ctx, text, ps, offs, engine=spm.ENGINE_FILTER)
        st = h.stats()
        got = h.view()
    finally:
        del os.environ["SPM_HIP_FILTER_SPAN_BUDGET"]
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0 and st.fallback_spans > 0
    assert len(ref) >= 90 and np.array_equal(got, ref)
    for p in (0, 5, 50):
        want = []
        for s in range(len(lens)):
            b, e = int(offs[s]), int(offs[s + 1])
            if e - b >= 1:
                r = oracle.myers(T[b:e], needles[p], k)
                want += [(int(x) + b, int(sc)) for x, sc in zip(r["pos"], r["score"])]
        mine = got[got["pattern"] == p]
        assert sorted((int(a), int(c)) for a, c in zip(mine["pos"], mine["score"])) == sorted(want)


@pytest.mark.parametrize("ppm,every", [(50000, 8), (120000, 4)])
def test_band_runs_verify_adjacent_bands_once(spm, ctx, oracle, ppm, every):
    """Runs of adjacent bands (a repeat stretch that a needle nearly matches is hit on a hundred diagonals in a row) are
    verified by their head with ONE cold start (filter.hpp: band_runs_kernel).  Forced here for a short band list; a
    5 %-shaped 64 MiB text (and a 12 % one): hits == the scan without runs == the CPU oracle, nothing reported twice."""
    n = 1 << 26
    text = ctx.generate_repeats(SEED_TEXT, 0, n, ppm)
    needles = [spm.synth_repeat_pattern(SEED_TEXT, SEED_PAT, n, p, 100, 3, ppm, every)[0] for p in range(64)]
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=3)
    os.environ["SPM_HIP_VERIFY_RUNS"] = "0"
    try:
        plain = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
        st0 = plain.stats()
        want = plain.view()
    finally:
        del os.environ["SPM_HIP_VERIFY_RUNS"]
    os.environ["SPM_HIP_VERIFY_RUNS_MIN_BANDS"] = "0"
    try:
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 24)
        st = h.stats()
        got = h.view()
    finally:
        del os.environ["SPM_HIP_VERIFY_RUNS_MIN_BANDS"]
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0 and st0.fell_back == 0
    assert len(want) > 100 and np.array_equal(got, want)
    assert st.n_bands < st0.n_bands                      # followers are not verified on their own
    assert len(set(_hits_list(got))) == len(got)
    # the oracle, on the first 8 MiB (one sequential pass per needle) and for three needles on all of it
    T8 = text.download(0, 1 << 23)
    ref = oracle.scan_multi(oracle.MYERS, T8, needles, k=3, threads=8)
    assert np.array_equal(got[got["pos"] <= (1 << 23)], ref)
    T = text.download(0, n)
    for p in (0, every, 63):
        r = oracle.myers(T, needles[p], 3)
        mine = got[got["pattern"] == p]
        assert np.array_equal(mine["pos"], r["pos"]) and np.array_equal(mine["score"], r["score"])
