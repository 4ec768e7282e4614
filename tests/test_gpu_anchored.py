"""GPU: anchored keys (stride-1 needle sets of several passes, libspm_amd/csrc/spm_hip.hip build_filter_index).  Every seed
has ONE key, chosen so that it begins with the anchor dimer of its pass; the streaming kernel looks up only the text
windows that begin with it.  Whatever the needles look like -- uniform, low-complexity seeds that offer a single dimer,
needles that are repeats -- the hits equal the brute-force engine's, which the other tests pin to the oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _patterns(ctx, spm, needles, k, max_keys):
    os.environ["SPM_HIP_FILTER_MAX_KEYS"] = str(max_keys)
    os.environ["SPM_HIP_FILTER_STRIDE"] = "1"
    os.environ["SPM_HIP_FILTER_DENSE"] = "0"    # (sets of several stride-1 passes go to the dense pass by default: test_gpu_dense.py)
    try:
        return ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    finally:
        os.environ.pop("SPM_HIP_FILTER_MAX_KEYS", None)
        os.environ.pop("SPM_HIP_FILTER_STRIDE", None)
        os.environ.pop("SPM_HIP_FILTER_DENSE", None)


@pytest.mark.parametrize("kind", ["uniform", "low_complexity", "many_passes"])
def test_anchored_passes_equal_brute_force(spm, ctx, oracle, kind):
    rng = np.random.default_rng({"uniform": 1, "low_complexity": 2, "many_passes": 3}[kind])
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 150, 3
    needles = []
    n_needles = 600 if kind != "many_passes" else 1500
    for i in range(n_needles):
        at = int(rng.integers(0, n - 2 * L))
        nd = T[at:at + L].copy()
        if kind == "low_complexity" and i % 3 == 0:
            # a seed (37 symbols) that offers one dimer only, or two: poly-A, (AC)n, (AAC)n ...
            unit = [[0], [3], [0, 1], [2, 3], [0, 0, 1], [1, 2, 3, 3]][i % 6]
            s = 37 * int(rng.integers(0, 4))
            rep = np.resize(np.array(unit, np.uint8), 37)
            nd[s:s + 37] = rep
            T[at:at + L] = nd           # the text carries it too
            if i % 9 == 0:              # and a stretch of the same repeat elsewhere (seed hits without an occurrence)
                o = int(rng.integers(0, n - 400))
                T[o:o + 300] = np.resize(np.array(unit, np.uint8), 300)
        for e in range(i % (k + 1)):    # planted edits
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
        needles.append(nd)
    if kind == "low_complexity":        # needles that ARE repeats
        needles.append(np.zeros(L, np.uint8))
        needles.append(np.resize(np.array([0, 1], np.uint8), L))
        T[1000:1400] = 0
        T[5000:5600] = np.resize(np.array([0, 1], np.uint8), 600)
    text = ctx.upload(T)
    ps = _patterns(ctx, spm, needles, k, 1024 if kind != "many_passes" else 600)
    assert ps.filterable
    h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    st = h.stats()
    assert st.fell_back == 0 and st.main_launches >= 3      # 2400+ seeds, <= 1024 per pass
    got = h.view()
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert len(want) >= n_needles
    assert np.array_equal(got, want)
    # the same set unanchored: identical hits
    os.environ["SPM_HIP_FILTER_ANCHOR"] = "0"
    try:
        ps0 = _patterns(ctx, spm, needles, k, 1024 if kind != "many_passes" else 600)
    finally:
        os.environ.pop("SPM_HIP_FILTER_ANCHOR", None)
    assert np.array_equal(spm.scan(ctx, text, ps0, engine=spm.ENGINE_FILTER).view(), want)
    # and a few needles against the oracle
    for p in (0, 7, len(needles) - 1):
        o = oracle.myers(T[:1 << 20], needles[p], k)
        mine = want[(want["pattern"] == p) & (want["pos"] <= (1 << 20))]
        assert sorted((int(x["pos"]), int(x["score"])) for x in mine) == sorted((int(x["pos"]), int(x["score"])) for x in o)
