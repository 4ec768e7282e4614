"""CPU: the oracle against every known-answer vector the reference's tests hold for the path, and against the
O(nm) Sellers DP / naive search for everything those vectors leave unpinned."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")
G = json.load(open(os.path.join(GOLD, "reference_vectors.json")))


def test_reference_known_answers(oracle):
    O = oracle
    H, P = O.encode(G["haystack"]), O.encode(G["needle"])
    for case in G["cases"]:
        if case["matcher"] == "horspool":
            got = O.horspool(H, P).tolist()
        elif case["matcher"] == "shiftor":
            got = O.shiftor(H, P).tolist()
        elif "chunk_size" in case:
            st, got, sc = O.myers_state(len(P), case["k"]), [], []
            for off in range(0, len(H), case["chunk_size"]):
                r = O.myers(H[off:off + case["chunk_size"]], P, case["k"], state=st, text_offset=off)
                got += r["pos"].tolist()
                sc += r["score"].tolist()
            assert sc == case["scores"]
        else:
            r = O.myers(H, P, case["k"])
            got = r["pos"].tolist()
            assert r["score"].tolist() == case["scores"]
        assert got == case["expected"], case["name"]


def test_pigeonhole_reference_seed_hits_are_exact_occurrences(oracle):
    """pigeonhole_matcher_test.cpp:30-33 with error rate 0: seed hits == exact occurrences of each needle."""
    O = oracle
    H = O.encode(G["haystack"])
    pg = G["pigeonhole"]
    assert O.naive_exact(H, O.encode(pg["needles"][0])).tolist() == pg["single_expected_begin"]
    both = sorted(O.naive_exact(H, O.encode(pg["needles"][0])).tolist()
                  + O.naive_exact(H, O.encode(pg["needles"][1])[:5]).tolist()
                  + O.naive_exact(H, O.encode(pg["needles"][1])[5:]).tolist())
    assert both == pg["multi_expected_begin"]


def _fasta(path):
    recs, name = [], None
    for line in open(path):
        line = line.strip()
        if line.startswith(">"):
            recs.append([line[1:], ""])
        elif line and recs:
            recs[-1][1] += line
    return recs


def test_reference_fasta_fixture_reads_occur_in_their_source(oracle):
    O = oracle
    refs = _fasta(os.path.join(GOLD, "sim_refx5.fasta"))
    assert len(refs) == 5
    for i in range(5):
        ref = O.encode(refs[i][1])
        for _, read in _fasta(os.path.join(GOLD, f"sim_reads_ref{i + 1}x10.fa")):
            r = O.encode(read)
            occ = O.shiftor(ref, r)
            assert len(occ) >= 1
            assert occ.tolist() == O.horspool(ref, r).tolist() == O.naive_exact(ref, r).tolist()
            m = O.myers(ref, r, 0)
            assert (m["pos"] - len(r)).tolist() == occ.tolist()


@pytest.mark.parametrize("m", [1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 150, 257, 1024])
def test_myers_variants_equal_sellers(oracle, m):
    O = oracle
    rng = np.random.default_rng(m)
    for k in (0, 1, 3, 17, 64):
        if k >= m:
            continue
        T = rng.integers(0, 4, 4000, dtype=np.uint8)
        P = rng.integers(0, 4, m, dtype=np.uint8)
        T[100:100 + m] = P
        if m > 3:
            T[2000:2000 + m - 1] = np.delete(P, m // 2)
        ref = O.sellers(T, P, k)
        assert len(ref) >= 1
        for v in ([0] if m <= 64 else []) + [1, 2]:
            assert np.array_equal(ref, O.myers(T, P, k, variant=v)), (m, k, v)
        refp = O.sellers(T[100:100 + m + k + 1], P, k, mode=O.PREFIX)
        for v in ([0] if m <= 64 else []) + [1, 2]:
            assert np.array_equal(refp, O.myers(T[100:100 + m + k + 1], P, k, variant=v, mode=O.PREFIX))


def test_myers_chunked_state_equals_sequential(oracle):
    O = oracle
    rng = np.random.default_rng(3)
    T = rng.integers(0, 4, 3000, dtype=np.uint8)
    for m, k in ((5, 1), (64, 2), (100, 3), (200, 8)):
        P = rng.integers(0, 4, m, dtype=np.uint8)
        T[500:500 + m] = P
        for v in ([0] if m <= 64 else [1, 2]):
            whole = O.myers(T, P, k, variant=v)
            st = O.myers_state(m, k, cutoff=(v == 2))
            parts = [O.myers(T[a:a + 77], P, k, variant=v, state=st, text_offset=a) for a in range(0, len(T), 77)]
            assert np.array_equal(np.concatenate(parts), whole)


def test_exact_matchers_equal_naive_incl_overlaps_and_dna5(oracle):
    O = oracle
    rng = np.random.default_rng(4)
    for sigma in (4, 5, 15):
        for m in (1, 2, 7, 32, 33, 70):
            T = rng.integers(0, sigma, 5000, dtype=np.uint8)
            P = rng.integers(0, sigma, m, dtype=np.uint8)
            T[10:10 + m] = P
            T[4000:4000 + 3 * m] = np.tile(P[:1], 3 * m)  # run of one symbol
            for pat in (P, np.tile(P[:1], m)):
                want = O.naive_exact(T, pat).tolist()
                assert O.horspool(T, pat, sigma).tolist() == want
                assert O.shiftor(T, pat, sigma).tolist() == want
                st = O.shiftor_state(m)
                got = []
                for a in range(0, len(T), 61):
                    got += O.shiftor(T[a:a + 61], pat, sigma, state=st, text_offset=a).tolist()
                assert got == want
    assert len(O.horspool(O.encode("ACG"), O.encode("ACGT"))) == 0
    assert len(O.shiftor(O.encode("ACG"), np.zeros(0, np.uint8))) == 0


def test_alphabet_rank_tables(oracle):
    """seqan3 rank orders as adapted by seqan/alphabet.hpp:68-77 (dna4 ACGT, dna5 ACGNT, dna15 ABCDGHKMNRSTVWY)."""
    O = oracle
    assert O.encode("ACGT").tolist() == [0, 1, 2, 3]
    assert O.encode("ACGNT", 5).tolist() == [0, 1, 2, 3, 4]
    assert O.decode(range(15), 15) == "ABCDGHKMNRSTVWY"
    assert O.encode("acgu").tolist() == [0, 1, 2, 3]
    assert O.encode("X", 5).tolist() == [3]


def test_synthetic_generator_matches_product_copy(oracle, spm):
    """libspm_hip's host-side needle generator (C ABI) against the oracle's restatement; no GPU needed."""
    O = oracle
    L = spm.capi.lib()
    for z in (0, 1, 0x5EED0001, 2**63 + 12345):
        assert L.spm_hip_mix64(z) == O.lib().spm_oracle_mix64(z)
    for (Lp, kmax) in ((32, 0), (100, 3), (150, 3), (1024, 64)):
        for p in range(0, 40):
            a, oa = spm.synth_pattern(0x5EED0001, 0x5EED0002, 1 << 34, p, Lp, kmax)
            b, ob = O.pattern(0x5EED0001, 0x5EED0002, 1 << 34, p, Lp, kmax)
            assert oa == ob and np.array_equal(a, b)
            # the planted occurrence is within e = p mod (kmax+1) edits of the source text
            src = O.text(0x5EED0001, oa, Lp + kmax + 8)
            r = O.sellers(src, a, p % (kmax + 1), mode=O.PREFIX)
            assert len(r) >= 1


def test_checksum_is_order_independent(oracle):
    O = oracle
    h = np.zeros(5, dtype=O.HIT_DTYPE)
    h["pos"] = [5, 1 << 35, 7, 9, 11]
    h["pattern"] = [0, 1023, 5, 5, 99999]
    h["score"] = [0, 3, 1, 2, 0]
    assert O.checksum(h) == O.checksum(h[::-1].copy())
    assert O.checksum(h) != O.checksum(h[:4])


def test_two_block_register_loop_equals_block_variant_and_sellers(oracle):
    """The loop the CPU baseline times (spm_oracle_myers2_fast) == variant 2 (blocks + cut-off) == Sellers, hit for hit:
    |P| in 65..128, k up to 70, texts with planted near-occurrences (the band grows and shrinks) and low-complexity runs."""
    O = oracle
    rng = np.random.default_rng(3)
    for m, k in [(65, 0), (65, 3), (100, 3), (100, 1), (127, 5), (128, 3), (128, 12), (96, 40), (100, 70)]:
        n = 6000
        T = rng.integers(0, 4, n, dtype=np.uint8)
        P = rng.integers(0, 4, m, dtype=np.uint8)
        for at, edits in ((500, 0), (1500, min(k, 2)), (2500, k), (3500, k + 1)):
            occ = P.copy()
            for e in range(edits):
                occ[(7 * e + 3) % m] ^= 1
            T[at:at + m] = occ
        T[4000:4300] = 0
        T[4300:4300 + min(m, 100)] = P[:100]
        got = O.myers2_fast(T, P, k)
        ref = O.myers(T, P, k, variant=2)
        sel = O.sellers(T, P, k)
        assert np.array_equal(got["pos"], ref["pos"]) and np.array_equal(got["score"], ref["score"]), (m, k)
        assert np.array_equal(got["pos"], sel["pos"]) and np.array_equal(got["score"], sel["score"]), (m, k)
        assert len(got) >= 1
    # the multi-needle driver takes this loop for two-block needles: same list as one variant-2 pass per needle
    T = O.text(0x5EED0001, 0, 1 << 16)
    needles = [O.pattern(0x5EED0001, 0x5EED0002, 1 << 16, p, 100, 3)[0] for p in range(12)]
    multi = O.scan_multi(O.MYERS, T, needles, k=3, threads=3)
    one = []
    for p, nd in enumerate(needles):
        r = O.myers(T, nd, 3, variant=2)
        one += [(p, int(a), int(b)) for a, b in zip(r["pos"], r["score"])]
    assert sorted(zip(multi["pattern"].tolist(), multi["pos"].tolist(), multi["score"].tolist())) == sorted(one)
