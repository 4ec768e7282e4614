"""GPU parity at BASELINE.json's full sizes (C2 1 GiB, C3 16 GiB, C4's per-GPU 8 GiB), where the oracle cannot scan
the whole text: size-independent properties + the oracle on sub-ranges and around every reported hit.

  * every planted needle is found where the generator planted it (source offset o, <= p mod (k+1) edits)
  * hit list sorted by (pattern, pos), no duplicates
  * checksum of checksums: the hits of ragged text shards scanned with their left context add up to the
    whole scan's order-independent checksum, count for count
  * the 2-bit shadow gives the same checksum
  * soundness: every reported hit (a sample of 2000 for C4) re-derived by the oracle on the |P|+k symbols before it
  * completeness: the oracle's hit list on sub-ranges (around planted sites and at random places, read with their left
    context) equals the slice of the GPU's list
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED_TEXT = 0x5EED0001
SEED_PAT = 0x5EED0002
M64 = (1 << 64) - 1

CASES = {
    # name: (algo, |P|, k_max, needles, text symbols, oracle sub-range length, hits re-derived)
    "c2": ("shiftor", 32, 0, 1024, 1 << 30, 1 << 18, None),
    "c3": ("myers", 100, 3, 1024, 1 << 34, 1 << 18, None),
    "c4": ("myers", 150, 3, 100_000, 1 << 33, 1 << 12, 2000),
}


@pytest.mark.parametrize("name", ["c2", "c3", "c4"])
def test_full_size_properties(spm, ctx, oracle, name):
    O = oracle
    algo_name, L, kmax, n_pat, N, sub_len, n_verify = CASES[name]
    algo = spm.ALGO_MYERS if algo_name == "myers" else spm.ALGO_SHIFTOR
    o_algo = O.MYERS if algo_name == "myers" else O.SHIFTOR
    made = [spm.synth_pattern(SEED_TEXT, SEED_PAT, N, p, L, kmax) for p in range(n_pat)]
    needles = [m[0] for m in made]
    planted = np.array([m[1] for m in made], dtype=np.int64)
    # the device generator and the oracle's generator are the same function (checked bytewise elsewhere); needles too
    nd0, o0 = O.pattern(SEED_TEXT, SEED_PAT, N, n_pat - 1, L, kmax)
    assert o0 == planted[-1] and np.array_equal(nd0, needles[-1])

    text = ctx.generate(SEED_TEXT, 0, N)
    ps = ctx.patterns(algo, needles, k=kmax)
    assert ps.filterable
    window = ps.window_size(0)
    assert window == L + kmax
    max_hits = 1 << 22

    whole = spm.scan(ctx, text, ps, max_hits=max_hits)
    st = whole.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0
    H = whole.view().copy()
    total_sum = whole.checksum()
    assert O.checksum(H) == total_sum                       # device-side and oracle-side checksum agree on the list

    # sorted, no duplicates
    key = H["pattern"].astype(np.uint64) << np.uint64(40) | H["pos"]
    assert np.all(np.diff(key.astype(np.int64)) > 0)

    # planted occurrences: needle p was cut at `planted[p]` and carries e = p mod (k+1) edits
    pos_of = {}
    for p, pos, sc in zip(H["pattern"], H["pos"], H["score"]):
        pos_of.setdefault(int(p), []).append((int(pos), int(sc)))
    report_end = algo_name == "myers"
    for p in range(n_pat):
        e = p % (kmax + 1)
        want = int(planted[p]) + (L if report_end else 0)
        assert any(abs(pos - want) <= kmax and sc <= e for pos, sc in pos_of.get(p, [])), (p, want, pos_of.get(p))

    # checksum of checksums over ragged shards with left context
    cuts = [0, 12345, N // 3 + 7, N // 2 - 1, N - (N // 5) - 13, N]
    acc, cnt = 0, 0
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        h = spm.scan(ctx, text, ps, lo, hi, left_context=lo > 0, max_hits=max_hits)
        acc = (acc + h.checksum()) & M64
        cnt += len(h.view())
        h.close()
    assert cnt == len(H) and acc == total_sum

    # 2-bit shadow
    text.pack()
    assert text.packed
    hp = spm.scan(ctx, text, ps, max_hits=max_hits)
    assert hp.checksum() == total_sum and len(hp.view()) == len(H)
    hp.close()

    # soundness: the oracle re-derives each hit from the window before it
    rng = np.random.default_rng(7)
    idx = np.arange(len(H)) if n_verify is None or len(H) <= n_verify else np.sort(rng.choice(len(H), n_verify, replace=False))
    for i in idx:
        p, pos, sc = int(H["pattern"][i]), int(H["pos"][i]), int(H["score"][i])
        if report_end:
            lo = max(0, pos - window)
            frag = text.download(lo, pos - lo)
            r = O.myers(frag, needles[p], kmax)
            assert len(r["pos"]) and int(r["pos"][-1]) == pos - lo and int(r["score"][-1]) == sc, (p, pos, sc)
        else:
            assert np.array_equal(text.download(pos, L), needles[p]) and sc == 0

    # completeness on sub-ranges: three around planted sites, three anywhere
    starts = [max(0, int(planted[p]) - sub_len // 2) for p in (0, n_pat // 2, n_pat - 1)]
    starts += [int(x) for x in rng.integers(0, N - sub_len, 3)]
    for s in starts:
        s = min(s, N - sub_len)
        a = max(0, s - (window - 1))
        frag = text.download(a, s + sub_len - a)
        ref = O.scan_multi(o_algo, frag, needles, k=kmax, threads=16, cap=1 << 20)
        ref_pos = ref["pos"].astype(np.int64) + a
        # the oracle reports ends (Myers) / begins (exact) like the device; keep what this sub-range owns
        owned_lo, owned_hi = (s + 1, s + sub_len + 1) if report_end else (s, s + sub_len - L + 1)
        keep = (ref_pos >= owned_lo) & (ref_pos < owned_hi)
        want = sorted(zip(ref["pattern"][keep].tolist(), ref_pos[keep].tolist(), ref["score"][keep].tolist()))
        m = (H["pos"].astype(np.int64) >= owned_lo) & (H["pos"].astype(np.int64) < owned_hi)
        got = sorted(zip(H["pattern"][m].tolist(), H["pos"][m].astype(np.int64).tolist(), H["score"][m].tolist()))
        assert got == want, (name, s, len(got), len(want))

    whole.close()
    ps.close()
    text.close()


def _edit_needle(spm, src, L, e, seed):
    """bench.py's needle maker: e edits at pseudo-random places of a haplotype window (substitute / delete / insert)."""
    mix = spm.capi.lib().spm_hip_mix64
    out = [int(x) for x in src[:L + e]]
    for j in range(e):
        r = mix(seed + j + 1)
        at, kind = r % L, (r >> 32) % 3
        if kind == 0:
            out[at] = (out[at] + 1 + (r >> 40) % 3) & 3
        elif kind == 1:
            del out[at]
        else:
            out.insert(at, (r >> 40) & 3)
    return np.array(out[:L], dtype=np.uint8)


def test_full_size_c5_journaled_pan_genome(spm, ctx, oracle):
    """C5 at its bench size -- 2^27-base reference x 64 haplotypes (8.56 Gbases of haplotype sequence), 256 needles
    |P| = 1024, k <= 64 -- where per-haplotype scans of everything are out of reach:

      * every planted needle is reported on the haplotype it was cut from, where it was cut, with <= its planted edits
      * records sorted by (haplotype, pos, pattern), no duplicates
      * soundness + completeness on coordinate sub-ranges of three haplotypes: the device search's records there ==
        an ordinary scan of the extracted haplotype stretch == the oracle's hit list (one pass per needle)
      * the union of two block shards of the tree (one GPU each in the N > 1 run) == the whole search
    """
    O = oracle
    SEED_VAR = 0x5EED0003
    L, kmax, n_pat, n_hap = 1024, 64, 256, 64
    ref_len = (1 << 27) // 640000 * 640000
    mix = spm.capi.lib().spm_hip_mix64
    ref = ctx.generate(SEED_TEXT, 0, ref_len)
    alleles, pool, cov = spm.synth_variants(SEED_TEXT, SEED_VAR, 0, ref_len, n_hap)
    jst = spm.Jst(ctx, ref, alleles, pool, cov.reshape(-1, 1), n_hap)
    window = L + kmax
    needles, planted = [], []
    for p in range(n_pat):
        r = mix(SEED_PAT + 7919 * p)
        h = r % n_hap
        o = (r >> 8) % (jst.haplotype_length(h) - 2 * window)
        e = p % (kmax + 1)
        needles.append(_edit_needle(spm, jst.extract(h, o, window), L, e, SEED_PAT ^ (p << 20)))
        planted.append((h, o, e))
    ps = ctx.patterns(spm.ALGO_MYERS, needles, k=kmax)
    assert ps.filterable
    st = jst.index(window, 1024)
    assert st.haplotype_symbols > 8.5e9 and st.context_symbols < st.haplotype_symbols / 4
    rec = jst.search(ps, max_hits=1 << 23)
    js = jst.stats()
    assert js.engine_used == spm.ENGINE_FILTER and js.fell_back == 0
    assert len(rec) > 500_000

    # sorted, unique
    key = np.stack([rec["haplotype"].astype(np.int64), rec["pos"].astype(np.int64), rec["pattern"].astype(np.int64)], 1)
    d = np.diff(key, axis=0)
    lex_up = (d[:, 0] > 0) | ((d[:, 0] == 0) & ((d[:, 1] > 0) | ((d[:, 1] == 0) & (d[:, 2] > 0))))
    assert lex_up.all()

    # planted needles
    for p, (h, o, e) in enumerate(planted):
        m = (rec["pattern"] == p) & (rec["haplotype"] == h)
        assert np.any((np.abs(rec["pos"][m].astype(np.int64) - (o + L)) <= kmax) & (rec["score"][m] <= e)), (p, h, o, e)

    # sub-ranges of three haplotypes: around two planted sites each + one anywhere
    rng = np.random.default_rng(11)
    sub = 1 << 16
    for h in (0, 17, 63):
        hl = jst.haplotype_length(h)
        sites = [o for (hh, o, e) in planted if hh == h][:2] + [int(rng.integers(0, hl - sub))]
        for s in sites:
            s = max(0, min(int(s) - sub // 2, hl - sub))
            a = max(0, s - (window - 1))
            frag = jst.extract(h, a, s + sub - a)
            owned_lo, owned_hi = s + 1, s + sub + 1          # exclusive end positions owned by [s, s + sub)
            m = (rec["haplotype"] == h) & (rec["pos"].astype(np.int64) >= owned_lo) & (rec["pos"].astype(np.int64) < owned_hi)
            got = sorted(zip(rec["pattern"][m].tolist(), rec["pos"][m].astype(np.int64).tolist(), rec["score"][m].tolist()))
            t = ctx.upload(frag)
            v = spm.scan(ctx, t, ps, max_hits=1 << 20).view()
            t.close()
            pos = v["pos"].astype(np.int64) + a
            keep = (pos >= owned_lo) & (pos < owned_hi)
            direct = sorted(zip(v["pattern"][keep].tolist(), pos[keep].tolist(), v["score"][keep].tolist()))
            assert got == direct, (h, s, len(got), len(direct))
            o_ref = O.scan_multi(O.MYERS, frag, needles, k=kmax, threads=16, cap=1 << 20)
            opos = o_ref["pos"].astype(np.int64) + a
            ok = (opos >= owned_lo) & (opos < owned_hi)
            want = sorted(zip(o_ref["pattern"][ok].tolist(), opos[ok].tolist(), o_ref["score"][ok].tolist()))
            assert got == want, (h, s, len(got), len(want))

    # block shards: two halves of the tree == the whole
    nb = int(st.n_blocks)
    parts = []
    for b0, b1 in ((0, nb // 2), (nb // 2, nb)):
        jst.index(window, 1024, b0, b1)
        parts.append(jst.search(ps, max_hits=1 << 23))
    both = np.sort(np.concatenate(parts), order=["haplotype", "pos", "pattern", "score"])
    assert np.array_equal(both, np.sort(rec, order=["haplotype", "pos", "pattern", "score"]))
    jst.close()
    ps.close()
    ref.close()
