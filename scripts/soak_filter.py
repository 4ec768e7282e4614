"""Soak: large random needle sets (strides 1 and 2, sub-batches, short keys, queue/chunk paths) -- filter engine against
the brute-force engine.  Not part of the test suite; run on a GPU box: python scripts/soak_filter.py [iters] [seed]."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libspm_amd as S  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = S.Context(0)
for it in range(iters):
    n = int(rng.choice([1 << 20, (1 << 22) + 123, 1 << 23]))
    sigma = int(rng.choice([4, 4, 5]))
    T = rng.integers(0, 4, n, dtype=np.uint8)
    if sigma == 5:
        T[T == 3] = 4
        T[rng.integers(0, n, 100)] = 3
    if it % 4 == 1:  # repeat stretches: microsatellites and low-complexity runs, ~1 % of the text
        for _ in range(n // 13000):
            o = int(rng.integers(0, n - 300)); ln = int(rng.integers(16, 257))
            unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
            if sigma == 5:
                unit[unit == 3] = 4
            T[o:o + ln] = np.resize(unit, ln)
    n_pat = int(rng.choice([200, 3000, 9000, 16000, 40000]))
    m = int(rng.choice([48, 52, 60, 64, 80, 100, 150]))
    k = int(rng.integers(0, min(4, m // 12)))
    offs = rng.integers(0, n - m - 8, n_pat)
    needles = []
    for i, o in enumerate(offs):
        nd = T[o:o + m].copy()
        if sigma == 5:
            nd[nd == 3] = 0
        for _ in range(int(rng.integers(0, k + 1))):
            nd[rng.integers(0, m)] = rng.choice([0, 1, 2, 4] if sigma == 5 else [0, 1, 2, 3])
        needles.append(nd)
    text = ctx.upload(T, sigma=sigma)
    algo = S.ALGO_SHIFTOR if (k == 0 and it % 2 == 0) else S.ALGO_MYERS  # (exact sets: hits reported by the resolve kernel)
    ps = ctx.patterns(algo, needles, k=k, sigma=sigma)
    if not ps.filterable:
        print(it, "not filterable", m, k)
        continue
    lo, hi, lc = 0, n, False
    if it % 3 == 2:
        lo = int(rng.integers(1, n // 2)); hi = int(rng.integers(lo + 1, n + 1)); lc = bool(it % 2)
    hb = S.scan(ctx, text, ps, lo, hi, engine=S.ENGINE_BRUTE, left_context=lc, max_hits=1 << 23).view()
    hf = S.scan(ctx, text, ps, lo, hi, engine=S.ENGINE_FILTER, left_context=lc, max_hits=1 << 23)
    st = hf.stats()
    assert st.fell_back == 0, (it, "fell back")
    assert np.array_equal(hf.view(), hb), (it, n, sigma, n_pat, m, k, lo, hi, lc, len(hb))
    print(it, "ok", n, sigma, n_pat, m, k, "launches", st.main_launches, "cand", st.n_candidates, "hits", len(hb), flush=True)
    text.close(); ps.close()
print("soak ok")
