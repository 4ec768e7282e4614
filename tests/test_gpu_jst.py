"""Journaled-sequence search on the device (C ABI spm_hip_jst_*) against what SURVEY 8(f)-2 defines it to be:
the union over haplotypes of a linear scan of each materialised haplotype, in haplotype coordinates.

The materialised haplotypes come from an independent numpy application of the alleles (checked against
spm_hip_jst_extract); the per-haplotype scans run through the ordinary C-ABI scan, one of them also through the oracle.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED_TEXT = 0x5EED0001
SEED_VAR = 0x5EED0003


def _apply(ref, alleles, pool, cov, h):
    """Haplotype h spelled out on the host: walk the reference, splice in the alleles h carries."""
    out, r = [], 0
    for i, a in enumerate(alleles):
        if not (int(cov[i, h >> 6]) >> (h & 63)) & 1:
            continue
        p, rl, al, ao = int(a["pos"]), int(a["ref_len"]), int(a["alt_len"]), int(a["alt_off"])
        out.append(ref[r:p])
        out.append(pool[ao:ao + al])
        r = min(len(ref), p + rl)
    out.append(ref[r:])
    return np.concatenate(out).astype(np.uint8)


def _random_alleles(rng, n_ref, n_hap, n_var, max_len):
    """Non-overlapping random alleles (SNPs, insertions, deletions, replacements), some multi-allelic sites."""
    cw = (n_hap + 63) // 64
    pos = np.sort(rng.choice(np.arange(1, n_ref - max_len - 2, max_len + 2), size=n_var, replace=False))
    rows, pool, cov = [], [], []

    def add(p, rl, alt, c):
        rows.append((p, rl, len(alt), sum(len(x) for x in pool)))
        pool.append(np.asarray(alt, dtype=np.uint8))
        cov.append(c)

    def rand_cov():
        bits = rng.random(n_hap) < rng.choice([0.05, 0.3, 0.5, 0.9])
        if not bits.any():
            bits[rng.integers(0, n_hap)] = True
        return bits

    def words(bits):
        w = np.zeros(cw, dtype=np.uint64)
        for h in np.nonzero(bits)[0]:
            w[h >> 6] |= np.uint64(1) << np.uint64(h & 63)
        return w

    for p in pos:
        kind = rng.integers(0, 5)
        c = rand_cov()
        if kind == 0:      # SNP
            add(int(p), 1, rng.integers(0, 4, 1), words(c))
        elif kind == 1:    # insertion
            add(int(p), 0, rng.integers(0, 4, rng.integers(1, max_len + 1)), words(c))
        elif kind == 2:    # deletion
            add(int(p), int(rng.integers(1, max_len + 1)), [], words(c))
        elif kind == 3:    # replacement
            add(int(p), int(rng.integers(1, max_len + 1)), rng.integers(0, 4, rng.integers(1, max_len + 1)), words(c))
        else:              # multi-allelic site: two alleles at one position, disjoint haplotypes
            c2 = rand_cov() & ~c
            add(int(p), 1, rng.integers(0, 4, 1), words(c))
            if c2.any():
                add(int(p), int(rng.integers(0, 3)), rng.integers(0, 4, rng.integers(0, 4)), words(c2))
    al = np.array(rows, dtype=[("pos", "<u8"), ("ref_len", "<u4"), ("alt_len", "<u4"), ("alt_off", "<u8")])
    return al, (np.concatenate(pool) if pool else np.zeros(0, np.uint8)), np.array(cov, dtype=np.uint64).reshape(len(rows), cw)


def _needles_from(rng, haps, n, L, k):
    out = []
    for _ in range(n):
        hp = haps[rng.integers(0, len(haps))]
        o = int(rng.integers(0, len(hp) - L))
        nd = hp[o:o + L].copy()
        for _e in range(int(rng.integers(0, k + 1))):
            nd[rng.integers(0, L)] = rng.integers(0, 4)
        out.append(nd)
    return out


def _expected(spm, ctx, haps, ps, engine):
    exp = []
    for h, hp in enumerate(haps):
        t = ctx.upload(hp)
        r = spm.scan(ctx, t, ps, engine=engine, max_hits=1 << 20)
        v = r.view()
        exp += [(h, int(a), int(b), int(c)) for a, b, c in zip(v["pos"], v["pattern"], v["score"])]
        r.close()
        t.close()
    return sorted(exp)


def _got(rec):
    return sorted(zip(rec["haplotype"].tolist(), rec["pos"].tolist(), rec["pattern"].tolist(), rec["score"].tolist()))


@pytest.mark.parametrize("cfg", [
    # n_ref, n_hap, n_var, max allele length, algo, |P|, k, block_len
    (60_000, 13, 300, 12, "myers", 40, 2, 256),
    (60_000, 100, 400, 40, "myers", 64, 3, 512),     # two coverage words, alleles longer than half a window
    (30_000, 64, 80, 300, "myers", 50, 2, 256),       # deletions / insertions longer than a block
    (40_000, 7, 1500, 3, "shiftor", 24, 0, 128),      # dense variants, exact matcher (begin positions)
    (50_000, 33, 200, 20, "myers", 200, 8, 0),        # default block length
    (12_000, 1500, 60, 8, "myers", 32, 1, 256),       # two haplotype groups (contexts shared within groups of 1024)
])
def test_device_jst_equals_per_haplotype_scans(spm, ctx, oracle, cfg):
    n_ref, n_hap, n_var, max_len, algo_name, L, k, block = cfg
    rng = np.random.default_rng(n_ref + 7 * n_hap + n_var)
    ref_text = ctx.generate(SEED_TEXT, 0, n_ref)
    ref = ref_text.download(0, n_ref)
    alleles, pool, cov = _random_alleles(rng, n_ref, n_hap, n_var, max_len)
    jst = spm.Jst(ctx, ref_text, alleles, pool, cov, n_hap)
    haps = [_apply(ref, alleles, pool, cov, h) for h in range(n_hap)]
    for h in (0, n_hap // 2, n_hap - 1):
        assert jst.haplotype_length(h) == len(haps[h])
        assert np.array_equal(jst.extract(h, 0, len(haps[h])), haps[h])
        b = len(haps[h]) // 3
        assert np.array_equal(jst.extract(h, b, 777), haps[h][b:b + 777])

    algo = spm.ALGO_MYERS if algo_name == "myers" else spm.ALGO_SHIFTOR
    needles = _needles_from(rng, haps, 24, L, k)
    ps = ctx.patterns(algo, needles, k=k)
    window = max(ps.window_size(p) for p in range(len(needles)))
    exp = _expected(spm, ctx, haps, ps, spm.ENGINE_BRUTE)
    assert len(exp) >= len(needles)
    # anchor one haplotype on the oracle as well
    O = oracle
    ref_hits = O.scan_multi(O.MYERS if algo_name == "myers" else O.SHIFTOR, haps[0], needles, k=k, threads=8)
    assert sorted((0, int(a), int(b), int(c)) for a, b, c in zip(ref_hits["pos"], ref_hits["pattern"], ref_hits["score"])) \
        == [e for e in exp if e[0] == 0]

    st = jst.index(window, block)
    assert st.unique_contexts <= st.contexts and st.context_symbols > 0
    assert st.haplotype_symbols == sum(len(h) for h in haps)
    for engine in (spm.ENGINE_AUTO, spm.ENGINE_BRUTE):
        assert _got(jst.search(ps, engine=engine, max_hits=1 << 20)) == exp

    # a larger indexed window than the needles need is still exact; so is any block length
    jst.index(window + 37, 1000)
    assert _got(jst.search(ps, max_hits=1 << 20)) == exp

    # block shards: the union of the shards' hits is the whole
    n_blocks = jst.index(window, 256).n_blocks
    cuts = [0, n_blocks // 3, n_blocks // 3 + 1, n_blocks]
    parts = []
    for b0, b1 in zip(cuts[:-1], cuts[1:]):
        st = jst.index(window, 256, b0, b1)
        assert st.n_blocks == b1 - b0
        parts += _got(jst.search(ps, max_hits=1 << 20))
    assert sorted(parts) == exp
    jst.close()
    ps.close()
    ref_text.close()


def test_device_jst_synthetic_c5_shape(spm, ctx, oracle):
    """The C5 generator at a small size: |P| = 1024, k = 64 needles over 64 haplotypes."""
    n_ref, n_hap, L, k = 400_000, 64, 1024, 64
    rng = np.random.default_rng(5)
    ref_text = ctx.generate(SEED_TEXT, 0, n_ref)
    ref = ref_text.download(0, n_ref)
    alleles, pool, cov = spm.synth_variants(SEED_TEXT, SEED_VAR, 0, n_ref, n_hap)
    assert 400 <= len(alleles) <= 440                       # one SNP per 1000 bases + one indel per 10 000
    assert np.all(np.diff(alleles["pos"].astype(np.int64)) > 0)
    cov2 = cov.reshape(-1, 1)
    jst = spm.Jst(ctx, ref_text, alleles, pool, cov2, n_hap)
    haps = [_apply(ref, alleles, pool, cov2, h) for h in range(n_hap)]
    needles = _needles_from(rng, haps, 6, L, k)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    exp = _expected(spm, ctx, haps, ps, spm.ENGINE_AUTO)
    st = jst.index(L + k, 1024)
    assert st.context_symbols < st.haplotype_symbols          # sharing pays: fewer symbols than 64 haplotypes
    got = _got(jst.search(ps, max_hits=1 << 22))
    assert got == exp
    assert jst.stats().engine_used == spm.ENGINE_FILTER
    jst.close()


def test_device_jst_two_filter_passes_with_merging(spm, ctx, oracle):
    """900 needles |P| = 1024, k = 64: 59 400 seeds -> two seed-filter passes feeding one band-merging + wave-verification
    stage over segmented contexts.  Equals per-haplotype brute-force scans."""
    n_ref, n_hap, L, k = 150_000, 12, 1024, 64
    rng = np.random.default_rng(23)
    ref_text = ctx.generate(SEED_TEXT, 0, n_ref)
    ref = ref_text.download(0, n_ref)
    alleles, pool, cov = spm.synth_variants(SEED_TEXT, SEED_VAR, 0, n_ref, n_hap)
    cov2 = cov.reshape(-1, 1)
    jst = spm.Jst(ctx, ref_text, alleles, pool, cov2, n_hap)
    haps = [_apply(ref, alleles, pool, cov2, h) for h in range(n_hap)]
    needles = _needles_from(rng, haps, 900, L, k)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    assert ps.filterable
    exp = _expected(spm, ctx, haps, ps, spm.ENGINE_BRUTE)
    jst.index(L + k, 512)
    h = jst.search_device(ps, max_hits=1 << 23)
    got = _got(h.view())
    h.close()
    st = jst.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.main_launches == 2 and st.fell_back == 0
    assert 0 < st.bands < st.candidates
    assert got == exp and len(exp) >= 900
    jst.close()


def test_device_jst_dna5_reference(spm, ctx, oracle):
    """A reference with N (dna5 ranks A C G N T): contexts inherit the alphabet, the seed filter masks windows with N."""
    rng = np.random.default_rng(17)
    n_ref, n_hap, L, k = 80_000, 20, 60, 2
    ref = rng.integers(0, 4, n_ref, dtype=np.uint8)
    ref[ref == 3] = 4
    ref[rng.integers(0, n_ref, 60)] = 3
    ref_text = ctx.upload(ref, sigma=5)
    alleles, pool, cov = _random_alleles(rng, n_ref, n_hap, 300, 10)
    pool[pool == 3] = 4
    jst = spm.Jst(ctx, ref_text, alleles, pool, cov, n_hap)
    haps = [_apply(ref, alleles, pool, cov, h) for h in range(n_hap)]
    needles = []
    while len(needles) < 16:
        nd = _needles_from(rng, haps, 1, L, k)[0]
        nd[nd == 3] = 0
        needles.append(nd)
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k, sigma=5)
    assert ps.filterable
    exp = []
    for h, hp in enumerate(haps):
        t = ctx.upload(hp, sigma=5)
        v = spm.scan(ctx, t, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 20).view()
        exp += [(h, int(a), int(b), int(c)) for a, b, c in zip(v["pos"], v["pattern"], v["score"])]
        t.close()
    jst.index(L + k, 256)
    assert _got(jst.search(ps, max_hits=1 << 20)) == sorted(exp) and len(exp) > 0
    assert jst.stats().engine_used == spm.ENGINE_FILTER
    jst.close()


def test_device_jst_rejects_what_it_cannot_index(spm, ctx, oracle):
    ref_text = ctx.upload(np.zeros(1000, dtype=np.uint8))
    al = np.array([(10, 5, 0, 0), (12, 1, 1, 0)], dtype=spm.ALLELE_DTYPE)      # SNP inside a deletion, shared haplotype
    with pytest.raises(spm.SpmError):
        spm.Jst(ctx, ref_text, al, np.array([1], np.uint8), np.array([[3], [1]], np.uint64), 2)
    ok = spm.Jst(ctx, ref_text, al, np.array([1], np.uint8), np.array([[2], [1]], np.uint64), 2)   # disjoint: fine
    assert ok.haplotype_length(0) == 1000 and ok.haplotype_length(1) == 995
    ps = ctx.patterns(spm.ALGO_MYERS, [np.zeros(30, np.uint8)], k=1)
    with pytest.raises(spm.SpmError):
        ok.search(ps)                                       # not indexed yet
    ok.index(20)
    with pytest.raises(spm.SpmError):
        ok.search(ps)                                       # indexed window smaller than the needle's
    ok.index(31)
    rec = ok.search(ps)
    assert len(rec) > 0 and set(rec["haplotype"].tolist()) == {0, 1}
    with pytest.raises(spm.SpmError):                       # out of order
        spm.Jst(ctx, ref_text, al[::-1].copy(), np.array([1], np.uint8), np.array([[1], [2]], np.uint64), 2)
    ok.close()
