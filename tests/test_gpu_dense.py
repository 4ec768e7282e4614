"""GPU: the dense pass (libspm_amd/csrc/index_build.hpp build_dense_index, filter.hpp seed_filter_dense_kernel) -- ONE pass
over the text for a needle set of any size: anchored windows -> presence bits in LDS -> fingerprint buckets in L2 ->
survivors.  Forced here for small sets (SPM_HIP_FILTER_DENSE=2); whatever the needles look like, the hits equal the
brute-force engine's and the CPU oracle's."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dense(ctx, spm, algo, needles, k, **env):
    env = {"SPM_HIP_FILTER_DENSE": "2", **{k_: str(v) for k_, v in env.items()}}
    os.environ.update(env)
    try:
        ps = ctx.patterns(algo, needles, k=k)
    finally:
        for k_ in env:
            os.environ.pop(k_, None)
    bs = ps.build_stats()
    assert ps.filterable and bs.dense == 1 and bs.passes == 1, (bs.dense, bs.passes)
    return ps, bs


def _oracle_hits(O, T, needles, k, which):
    out = []
    for p in which:
        r = O.myers(T, needles[p], k)
        out += [(p, int(a), int(s)) for a, s in zip(r["pos"], r["score"])]
    return sorted(out)


def _as_list(h, which=None):
    return sorted((int(a), int(b), int(c)) for a, b, c in zip(h["pattern"], h["pos"], h["score"])
                  if which is None or int(a) in which)


@pytest.mark.parametrize("L,k,n_needles", [(150, 3, 700), (100, 3, 500), (64, 3, 200), (40, 1, 300), (200, 5, 300)])
def test_dense_pass_equals_brute_force_and_oracle(spm, ctx, oracle, L, k, n_needles):
    rng = np.random.default_rng(L * 10 + k)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = []
    for i in range(n_needles):
        at = int(rng.integers(0, n - 2 * L))
        nd = T[at:at + L].copy()
        for e in range(i % (k + 1)):    # planted substitutions, and an indel now and then
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
        if i % 7 == 3 and k >= 2:
            j = int(rng.integers(10, L - 10))
            nd = np.concatenate([nd[:j], nd[j + 1:], rng.integers(0, 4, 1, dtype=np.uint8)])
        needles.append(nd)
    text = ctx.upload(T)
    ps, bs = _dense(ctx, spm, spm.ALGO_MYERS, needles, k)
    assert 1 <= bs.anchor_sixteenths <= 16 and bs.keys >= n_needles * (k + 1)
    h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    st = h.stats()
    assert st.engine_used == spm.ENGINE_FILTER and st.fell_back == 0 and st.main_launches == 1
    got = h.view()
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert len(want) >= n_needles * 0.9
    assert np.array_equal(got, want)
    which = (0, 3, 10, n_needles - 1)
    assert _as_list(got[got["pos"] <= (1 << 20)], which) == _oracle_hits(oracle, T[:1 << 20], needles, k, which)


def test_dense_low_complexity_and_repeat_needles(spm, ctx, oracle):
    """Needles whose windows offer one or two dimers only (poly-A, (AC)n, (GTT)n) decide which anchors are usable; needles
    that ARE repeats put the same key at many offsets; the text carries long stretches of the same repeats."""
    rng = np.random.default_rng(77)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 150, 3
    needles = []
    for i in range(400):
        at = int(rng.integers(0, n - 2 * L))
        nd = T[at:at + L].copy()
        if i % 3 == 0:
            unit = [[0], [3], [0, 1], [2, 3], [0, 0, 1], [1, 2, 3, 3]][i % 6]
            s = 37 * int(rng.integers(0, 4))
            nd[s:s + 37] = np.resize(np.array(unit, np.uint8), 37)
            T[at:at + L] = nd
            if i % 9 == 0:
                o = int(rng.integers(0, n - 400))
                T[o:o + 300] = np.resize(np.array(unit, np.uint8), 300)
        for e in range(i % (k + 1)):
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
        needles.append(nd)
    needles.append(np.zeros(L, np.uint8))
    needles.append(np.resize(np.array([0, 1], np.uint8), L))
    needles.append(np.resize(np.array([2, 3, 3], np.uint8), L))
    T[1000:1400] = 0
    T[5000:5600] = np.resize(np.array([0, 1], np.uint8), 600)
    T[9000:9500] = np.resize(np.array([2, 3, 3], np.uint8), 500)
    text = ctx.upload(T)
    ps, bs = _dense(ctx, spm, spm.ALGO_MYERS, needles, k)
    h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, max_hits=1 << 22)
    assert h.stats().fell_back == 0
    got = h.view()
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 22).view()
    assert np.array_equal(got, want)
    which = (0, len(needles) - 3, len(needles) - 2, len(needles) - 1)
    assert _as_list(got[got["pos"] <= (1 << 18)], which) == _oracle_hits(oracle, T[:1 << 18], needles, k, which)


@pytest.mark.parametrize("density", [2, 4, 8, 16])
def test_dense_pass_at_forced_anchor_densities(spm, ctx, density):
    """Two .. sixteen sixteenths of the dimers as anchors (1, 2 or 3 patterns in the kernel): identical hits."""
    rng = np.random.default_rng(5)
    n = 1 << 21
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 120, 2
    needles = [T[a:a + L].copy() for a in rng.integers(0, n - L, 300)]
    for i, nd in enumerate(needles):
        for e in range(i % (k + 1)):
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
    text = ctx.upload(T)
    ps, bs = _dense(ctx, spm, spm.ALGO_MYERS, needles, k, SPM_HIP_FILTER_DENSE_MIN_DENSITY=density)
    assert bs.anchor_sixteenths >= density
    got = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER).view()
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert len(want) >= 300 and np.array_equal(got, want)


def test_dense_exact_matchers_subrange_and_segments(spm, ctx, oracle):
    rng = np.random.default_rng(11)
    n = 1 << 21
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = [T[a:a + 40].copy() for a in rng.integers(0, n - 40, 256)]
    text = ctx.upload(T)
    for algo in (spm.ALGO_SHIFTOR, spm.ALGO_HORSPOOL):
        ps, _ = _dense(ctx, spm, algo, needles, 0)
        got = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER).view()
        want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
        assert len(want) >= 256 and np.array_equal(got, want)
        for p in (0, 100, 255):
            o = oracle.naive_exact(T, needles[p])
            assert sorted(int(x) for x in got[got["pattern"] == p]["pos"]) == sorted(int(x) for x in o)
    # Myers: a shard with left context and a position offset == the same part of the whole scan
    L, k = 150, 3
    nd = [T[a:a + L].copy() for a in rng.integers(0, n - L, 300)]
    ps, _ = _dense(ctx, spm, spm.ALGO_MYERS, nd, k)
    whole = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER).view()
    lo, hi = 300_000 + 7, 1_500_000 + 3
    part = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=True, pos_offset=1 << 40).view()
    sel = whole[(whole["pos"] > lo) & (whole["pos"] <= hi)].copy()
    sel["pos"] += 1 << 40
    assert len(sel) > 100 and np.array_equal(part, sel)
    # independent haystacks stored back to back: == per-segment scans
    offs = np.array([0, 1000, 1000, 250_000, 250_100, 1_000_003, n], dtype=np.uint64)
    seg = spm.scan_segments(ctx, text, ps, offs, engine=spm.ENGINE_FILTER).view()
    ref = []
    for s in range(len(offs) - 1):
        b, e = int(offs[s]), int(offs[s + 1])
        if e > b:
            ref.append(spm.scan(ctx, text, ps, b, e, engine=spm.ENGINE_BRUTE).view())
    ref = np.concatenate(ref)
    ref = ref[np.lexsort((ref["pos"], ref["pattern"]))]
    assert len(ref) > 100 and np.array_equal(seg, ref)


@pytest.mark.parametrize("budget", [2, 16])
def test_dense_span_fallback_is_exact(spm, ctx, budget):
    """Spans whose survivors exceed a tiny budget give up and are scanned again by the brute-force kernel: same hits, once."""
    rng = np.random.default_rng(9)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 150, 3
    needles = [T[a:a + L].copy() for a in rng.integers(0, n - L, 400)]
    text = ctx.upload(T)
    ps, _ = _dense(ctx, spm, spm.ALGO_MYERS, needles, k)
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    os.environ["SPM_HIP_FILTER_SPAN_BUDGET"] = str(budget)
    try:
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
        st = h.stats()
        got = h.view()
    finally:
        del os.environ["SPM_HIP_FILTER_SPAN_BUDGET"]
    assert st.fell_back == 0 and st.fallback_spans > 0
    assert np.array_equal(got, want)


def test_dense_restorable_chunks(spm, ctx):
    """capture/restore across chunks with the dense pass taking the bulk of every chunk: hits and states equal the brute path's."""
    rng = np.random.default_rng(21)
    n = 3 << 19
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 150, 3
    needles = [T[a:a + L].copy() for a in rng.integers(0, n - L, 200)]
    text = ctx.upload(T)
    ps, _ = _dense(ctx, spm, spm.ALGO_MYERS, needles, k)
    cuts = [0, 1 << 19, (1 << 19) + 300_001, n]
    res = {}
    for engine in (spm.ENGINE_FILTER, spm.ENGINE_BRUTE):
        st = ps.initial_state()
        hits = []
        for b, e in zip(cuts[:-1], cuts[1:]):
            h, st = spm.scan(ctx, text, ps, b, e, engine=engine, state_in=st, want_state=True)
            hits.append(h.view())
        allh = np.concatenate(hits)
        res[engine] = (allh[np.lexsort((allh["pos"], allh["pattern"]))], st.copy())
    assert len(res[spm.ENGINE_BRUTE][0]) >= 200
    assert np.array_equal(res[spm.ENGINE_FILTER][0], res[spm.ENGINE_BRUTE][0])
    assert np.array_equal(res[spm.ENGINE_FILTER][1], res[spm.ENGINE_BRUTE][1])
    whole = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert np.array_equal(res[spm.ENGINE_BRUTE][0], whole)


def test_deferred_completion_of_exact_scans(spm, ctx):
    """SPM_SCAN_DEFER: an exact-set scan returns once its kernels are enqueued; the counters are read at the first accessor.
    The device-side fused copy writes {count, status | records} without the host.  A scan that needs attention (spans gave
    up) says so in the status word and is repeated, synchronously, when the host looks at it."""
    import torch
    rng = np.random.default_rng(41)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    needles = [T[a:a + 32].copy() for a in rng.integers(0, n - 32, 512)]
    text = ctx.upload(T)
    ps = ctx.patterns(spm.ALGO_SHIFTOR, needles, k=0)
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert len(want) >= 512
    cap = 4096
    buf = torch.zeros((cap + 1, 2), dtype=torch.int64, device="cuda")
    stream = torch.cuda.current_stream()
    for _ in range(3):
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER)
        h.copy_fused_device(buf.data_ptr(), cap)
        ctx.synchronize()
        stream.synchronize()
        head = buf[0].cpu().tolist()
        assert head == [len(want), 0]
        rec = buf[1:1 + len(want)].cpu().numpy().view(np.uint8).reshape(-1, 16)
        got = np.sort(np.frombuffer(rec.tobytes(), dtype=spm.HIT_DTYPE), order=["pattern", "pos"])
        assert np.array_equal(got, want)
        assert np.array_equal(h.view(), want) and h.stats().fell_back == 0     # (the accessor completes the scan)
        h.close()
    h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER)
    h.close()                                                                   # never looked at: completed on destroy
    # the fused-copy kernel delivers the counters to the host block and clears the device's for the scan that reuses the
    # buffers: steps back to back without a host access in between (what bench.py's C2 loop does), a second device-side
    # copy of one result, and an ordinary (Myers) scan on the recycled block
    buf2 = torch.zeros((cap + 1, 2), dtype=torch.int64, device="cuda")
    prev = None
    for _ in range(4):
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER)
        h.copy_fused_device(buf.data_ptr(), cap)
        if prev is not None:
            assert prev.stats().n_hits == len(want)
            prev.close()
        prev = h
    prev.copy_fused_device(buf2.data_ptr(), cap)                                # again: completed on the host first
    ctx.synchronize()
    assert buf[0].cpu().tolist() == [len(want), 0] and torch.equal(buf[:1 + len(want)], buf2[:1 + len(want)])
    prev.close()
    nd0 = [T[a:a + 64].copy() for a in rng.integers(0, n - 64, 32)]
    pm0 = ctx.patterns(spm.ALGO_MYERS, nd0, k=1)
    assert np.array_equal(spm.scan(ctx, text, pm0, engine=spm.ENGINE_FILTER).view(), spm.scan(ctx, text, pm0, engine=spm.ENGINE_BRUTE).view())
    # a scan that needs its host: spans give up -> status 1 in the header; view() repeats the scan the ordinary way
    os.environ["SPM_HIP_FILTER_SPAN_BUDGET"] = "1"
    try:
        h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER)
        h.copy_fused_device(buf.data_ptr(), cap)
        ctx.synchronize()
        assert buf[0, 1].item() == 1
        got = h.view()
        st = h.stats()
    finally:
        del os.environ["SPM_HIP_FILTER_SPAN_BUDGET"]
    assert np.array_equal(got, want) and st.fallback_spans > 0
    h.copy_fused_device(buf.data_ptr(), cap)                                    # completed: final records, status 0
    ctx.synchronize()
    assert buf[0].cpu().tolist() == [len(want), 0]
    # Myers sets ignore the flag (their verification may need a second attempt): same result, completed on return
    nd = [T[a:a + 100].copy() for a in rng.integers(0, n - 100, 64)]
    pm = ctx.patterns(spm.ALGO_MYERS, nd, k=2)
    a = spm.scan(ctx, text, pm, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER).view()
    b = spm.scan(ctx, text, pm, engine=spm.ENGINE_FILTER).view()
    assert np.array_equal(a, b) and len(a) >= 64


def _fused(buf, n_max):
    head = buf[0].cpu().tolist()
    n = min(head[0], n_max)
    rec = buf[1:1 + n].cpu().numpy().view(np.uint8).reshape(-1, 16)
    return head, np.sort(np.frombuffer(rec.tobytes(), dtype=[("pos", "<u8"), ("pattern", "<u4"), ("score", "<i4")]),
                         order=["pattern", "pos"])


@pytest.mark.parametrize("dense", [False, True])
def test_deferred_completion_of_myers_scans_and_the_poisoned_band_table(spm, ctx, dense):
    """SPM_SCAN_DEFER for sets that go through the band table (Myers, k > 0).  Clean scans: back to back without the host,
    results from the device-side fused copy.  A scan whose band list overflows leaves slots in the table the context's scans
    share; scans launched behind it -- before the host has looked -- must not trust that table: they find the device-side
    poison flag, declare themselves void (status 1) and are repeated when the host completes them."""
    import torch
    rng = np.random.default_rng(97 + dense)
    n = 1 << 22
    T = rng.integers(0, 4, n, dtype=np.uint8)
    L, k = 100, 3
    needles = []
    for i in range(300):
        at = int(rng.integers(0, n - L))
        nd = T[at:at + L].copy()
        for _ in range(i % (k + 1)):
            nd[int(rng.integers(0, L))] = rng.integers(0, 4)
        needles.append(nd)
    text = ctx.upload(T)
    if dense:
        ps, _ = _dense(ctx, spm, spm.ALGO_MYERS, needles, k)
    else:
        ps = ctx.patterns(spm.ALGO_MYERS, needles, k=k)
    want = spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE).view()
    assert len(want) >= 300
    cap = 1 << 14
    bufs = [torch.zeros((cap + 1, 2), dtype=torch.int64, device="cuda") for _ in range(4)]

    def deferred(buf, **env):
        os.environ.update({k_: str(v) for k_, v in env.items()})
        try:
            h = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER, flags=spm.SCAN_DEFER)
        finally:
            for k_ in env:
                os.environ.pop(k_, None)
        h.copy_fused_device(buf.data_ptr(), cap)
        return h

    # clean: four scans enqueued back to back, nothing read on the host in between
    hs = [deferred(bufs[i]) for i in range(4)]
    ctx.synchronize()
    for i, h in enumerate(hs):
        head, got = _fused(bufs[i], cap)
        assert head == [len(want), 0] and np.array_equal(got, want)
        assert np.array_equal(h.view(), want) and h.stats().fell_back == 0
        h.close()
    # a band list of 64 entries overflows (300 planted needles); the two scans behind it run on a table with leftovers
    a = deferred(bufs[0], SPM_HIP_FILTER_BAND_CAP=64)
    b = deferred(bufs[1])
    c = deferred(bufs[2])
    ctx.synchronize()
    assert [bufs[i][0, 1].item() for i in range(3)] == [1, 1, 1]          # all three say "ask the host"
    assert np.array_equal(c.view(), want)                                 # completed out of order: each repeats itself
    assert np.array_equal(a.view(), want) and np.array_equal(b.view(), want)
    for h in (a, b, c):
        assert h.stats().fell_back == 0
        h.close()
    d = deferred(bufs[3])                                                 # the table is empty again
    ctx.synchronize()
    head, got = _fused(bufs[3], cap)
    assert head == [len(want), 0] and np.array_equal(got, want)
    d.close()
    # an ordinary scan right behind a poisoned one that nobody has completed yet: heals itself (one repeated attempt)
    a = deferred(bufs[0], SPM_HIP_FILTER_BAND_CAP=64)
    plain = spm.scan(ctx, text, ps, engine=spm.ENGINE_FILTER)
    assert np.array_equal(plain.view(), want) and plain.stats().fell_back == 0
    assert np.array_equal(a.view(), want)
    a.close()
    plain.close()


def test_needles_as_a_matrix_equal_needles_as_a_list(spm, ctx):
    """Context.patterns takes reads of one length as a 2-D array (one row each): same set, same hits."""
    rng = np.random.default_rng(5)
    n = 1 << 20
    T = rng.integers(0, 4, n, dtype=np.uint8)
    rows = np.stack([T[a:a + 64] for a in rng.integers(0, n - 64, 300)])
    text = ctx.upload(T)
    a = ctx.patterns(spm.ALGO_MYERS, rows, k=2)
    b = ctx.patterns(spm.ALGO_MYERS, [r.copy() for r in rows], k=2)
    assert a.n == b.n == 300 and a.build_stats().keys == b.build_stats().keys
    ha, hb = spm.scan(ctx, text, a).view(), spm.scan(ctx, text, b).view()
    assert len(ha) >= 300 and np.array_equal(ha, hb)
    assert ctx.patterns(spm.ALGO_SHIFTOR, np.zeros((0, 32), np.uint8)).n == 0
