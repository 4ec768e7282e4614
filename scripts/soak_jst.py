"""Soak: random journaled sequence trees through the device index against per-haplotype scans (not part of the test
suite; run on a GPU box: python scripts/soak_jst.py [iterations] [seed])."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libspm_amd as S  # noqa: E402
from test_gpu_jst import _apply, _expected, _got, _needles_from, _random_alleles  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = S.Context(0)
for it in range(iters):
    n_ref = int(rng.integers(5_000, 60_000))
    n_hap = int(rng.choice([1, 2, 7, 33, 64, 65, 130]))
    max_len = int(rng.choice([1, 5, 30, 200]))
    n_var = int(rng.integers(1, max(2, (n_ref - max_len - 4) // (max_len + 2) // 2)))
    algo_name = "myers" if it % 3 else "shiftor"
    L = int(rng.choice([16, 24, 40, 64, 150, 300]))
    k = 0 if algo_name == "shiftor" else int(rng.integers(0, max(1, L // 13)))
    block = int(rng.choice([0, 64, 100, 256, 1000, 5000]))
    ref = rng.integers(0, 4, n_ref, dtype=np.uint8)
    ref_text = ctx.upload(ref)
    alleles, pool, cov = _random_alleles(rng, n_ref, n_hap, n_var, max_len)
    jst = S.Jst(ctx, ref_text, alleles, pool, cov, n_hap)
    haps = [_apply(ref, alleles, pool, cov, h) for h in range(n_hap)]
    if min(len(h) for h in haps) <= L + 2:
        continue
    needles = _needles_from(rng, haps, int(rng.integers(1, 20)), L, k)
    ps = ctx.patterns(S.ALGO_MYERS if algo_name == "myers" else S.ALGO_SHIFTOR, needles, k=k)
    window = max(ps.window_size(p) for p in range(len(needles)))
    exp = _expected(S, ctx, haps, ps, S.ENGINE_BRUTE)
    st = jst.index(window + int(rng.integers(0, 3)), block)
    assert st.haplotype_symbols == sum(len(h) for h in haps), (it, "symbols")
    got = _got(jst.search(ps, max_hits=1 << 22))
    assert got == exp, (it, n_ref, n_hap, n_var, max_len, algo_name, L, k, block, len(got), len(exp))
    jst.close()
    ps.close()
    ref_text.close()
    print(it, "ok", n_ref, n_hap, n_var, max_len, algo_name, L, k, block, len(exp), st.unique_contexts, st.contexts, flush=True)
print("soak ok")
