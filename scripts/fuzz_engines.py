#!/usr/bin/env python3
"""Differential fuzz on the GPU: random needle sets (sizes, mixed lengths, per-needle k, low-complexity needles, needles
with N, dna4 / dna5 texts, repeat stretches in the text, sub-ranges, segments) through the seed-filter engine -- sparse
passes, anchored passes, the dense pass, whichever the set takes or is forced to take -- against the brute-force engine
(one lane per needle, the kernel shape the parity tests pin on the oracle).  Any difference is printed with its seed and
the run exits non-zero.

    python scripts/fuzz_engines.py [--seconds 240] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def make_case(rng, spm, big=0.0):
    n = int(rng.choice([1 << 18, 1 << 20, 1 << 21, 3 * (1 << 19) + 12345]))
    sigma = int(rng.choice([4, 4, 4, 5]))
    T = rng.integers(0, 4, n, dtype=np.uint8)
    # repeat stretches / low complexity in the text
    for _ in range(int(rng.integers(0, 40))):
        at = int(rng.integers(0, n - 600))
        ln = int(rng.integers(16, 512))
        unit = rng.integers(0, 4, int(rng.integers(1, 7)), dtype=np.uint8)
        T[at:at + ln] = np.resize(unit, ln)      # (before the dna5 mapping below)
    key = np.array([0, 1, 2, 3], dtype=np.uint8)
    if sigma == 5:      # seqan3 dna5 ranks: A C G N T -- N is rank 3
        key = np.array([0, 1, 2, 4], dtype=np.uint8)
        T = key[T]
        for _ in range(int(rng.integers(1, 30))):
            at = int(rng.integers(0, n - 40))
            T[at:at + int(rng.integers(1, 30))] = 3
    algo = int(rng.choice([spm.ALGO_MYERS, spm.ALGO_MYERS, spm.ALGO_MYERS, spm.ALGO_SHIFTOR, spm.ALGO_HORSPOOL]))
    n_needles = int(rng.choice([3, 64, 65, 500, 3000, 9000]))
    if rng.random() < big:      # sets that take the dense pass by themselves (more keys than one fingerprint table holds)
        n_needles = int(rng.choice([16000, 40000, 100000]))
    kmax = 0 if algo != spm.ALGO_MYERS else int(rng.choice([0, 1, 2, 3, 5]))
    Lmin = int(rng.choice([24, 40, 64, 100]))
    Lmax = Lmin if rng.random() < 0.5 else Lmin + int(rng.integers(1, 120))
    if algo != spm.ALGO_MYERS:
        Lmax = min(Lmax, 64)
        Lmin = min(Lmin, Lmax)
    needles, ks = [], []
    for i in range(n_needles):
        L = int(rng.integers(Lmin, Lmax + 1))
        at = int(rng.integers(0, n - L - 8))
        nd = T[at:at + L].copy()
        k = kmax if rng.random() < 0.7 else int(rng.integers(0, kmax + 1))
        if (L // (k + 1)) < 12:
            k = max(0, L // 12 - 1)
        r = rng.random()
        if r < 0.08:       # a needle that is a repeat
            unit = key[rng.integers(0, 4, int(rng.integers(1, 5)))]
            nd = np.resize(unit, L).astype(np.uint8)
        elif r < 0.12 and sigma == 5:
            nd[int(rng.integers(0, L))] = 3
        for _ in range(int(rng.integers(0, k + 1))):      # edits within the budget
            nd[int(rng.integers(0, L))] = key[int(rng.integers(0, 4))]
        if k >= 2 and rng.random() < 0.15 and L > 30:
            j = int(rng.integers(10, L - 10))
            nd = np.concatenate([nd[:j], nd[j + 1:], key[rng.integers(0, 4, 1)]])
        needles.append(np.ascontiguousarray(nd, dtype=np.uint8))
        ks.append(k)
    env = {}
    mode = rng.random()
    if mode < 0.35:
        env["SPM_HIP_FILTER_DENSE"] = "2"
        if rng.random() < 0.4:
            env["SPM_HIP_FILTER_DENSE_MIN_DENSITY"] = str(int(rng.choice([2, 4, 8, 12, 16])))
    elif mode < 0.5:
        env["SPM_HIP_FILTER_DENSE"] = "0"
    if rng.random() < 0.15:
        env["SPM_HIP_FILTER_SPAN_BUDGET"] = str(int(rng.choice([1, 4, 16])))
    if rng.random() < 0.1:
        env["SPM_HIP_FILTER_BITS"] = "0"
    return dict(n=n, sigma=sigma, T=T, algo=algo, needles=needles, ks=np.asarray(ks, dtype=np.uint16), env=env)


def _sorted(h):
    return h[np.lexsort((h["score"], h["pos"], h["pattern"]))]


def check_chunks(spm, ctx, rng, c, text, ps, seed):
    """Restorable scanning: the text in 2..5 chunks, the state carried from chunk to chunk (capture / restore); hits and
    final state of the filter route == the brute-force route's == one whole scan."""
    n = c["n"]
    cuts = sorted({0, n, *[int(x) for x in rng.integers(1, n, int(rng.integers(1, 5)))]})
    res = {}
    for engine in (spm.ENGINE_AUTO, spm.ENGINE_BRUTE):
        st = ps.initial_state()
        hits = []
        for b, e in zip(cuts[:-1], cuts[1:]):
            h, st = spm.scan(ctx, text, ps, b, e, engine=engine, state_in=st, want_state=True, max_hits=1 << 24)
            hits.append(h.view().copy())
            h.close()
        res[engine] = (_sorted(np.concatenate(hits)), st.copy())
    whole = _sorted(spm.scan(ctx, text, ps, engine=spm.ENGINE_BRUTE, max_hits=1 << 24).view().copy())
    ok = (np.array_equal(res[spm.ENGINE_AUTO][0], res[spm.ENGINE_BRUTE][0]) and
          np.array_equal(res[spm.ENGINE_AUTO][1], res[spm.ENGINE_BRUTE][1]) and np.array_equal(res[spm.ENGINE_BRUTE][0], whole))
    if not ok:
        print(f"MISMATCH (chunks) seed {seed}: cuts {cuts} env {c['env']} algo {c['algo']} sigma {c['sigma']}", flush=True)
    return 0 if ok else 1


def check_segments(spm, ctx, rng, c, text, ps, seed):
    """A batch of haystacks back to back: the segmented scan == every segment scanned on its own by the brute-force engine."""
    n = c["n"]
    offs = sorted({0, n, *[int(x) for x in rng.integers(1, n, int(rng.integers(1, 12)))]})
    got = _sorted(spm.scan_segments(ctx, text, ps, offs, engine=spm.ENGINE_FILTER, max_hits=1 << 24).view().copy())
    parts = []
    for b, e in zip(offs[:-1], offs[1:]):
        parts.append(spm.scan(ctx, text, ps, b, e, engine=spm.ENGINE_BRUTE, max_hits=1 << 24).view().copy())
    want = _sorted(np.concatenate(parts))
    ok = np.array_equal(got, want)
    if not ok:
        print(f"MISMATCH (segments) seed {seed}: {len(got)} vs {len(want)} hits, offsets {offs} env {c['env']} algo {c['algo']} "
              f"sigma {c['sigma']}", flush=True)
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--big", type=float, default=0.0, help="share of cases with 16 000 .. 100 000 needles")
    args = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")      # torch's HIP runtime first (tests/conftest.py says why)
    import libspm_amd as spm
    ctx = spm.Context(0)
    t_end = time.time() + args.seconds
    seed = args.seed
    bad = 0
    done = 0
    kinds = {}
    while time.time() < t_end:
        rng = np.random.default_rng(seed)
        c = make_case(rng, spm, args.big)
        os.environ.update(c["env"])
        try:
            text = ctx.upload(c["T"], sigma=c["sigma"])
            ps = ctx.patterns(c["algo"], c["needles"], k=c["ks"], sigma=c["sigma"])
            if not ps.filterable:
                kinds["not filterable"] = kinds.get("not filterable", 0) + 1
                seed += 1
                continue
            bs = ps.build_stats()
            variant = rng.random()
            if variant < 0.12 and c["algo"] != spm.ALGO_HORSPOOL:
                bad += check_chunks(spm, ctx, rng, c, text, ps, seed)
                kinds["chunks"] = kinds.get("chunks", 0) + 1
                done += 1
                seed += 1
                continue
            if variant < 0.24:
                bad += check_segments(spm, ctx, rng, c, text, ps, seed)
                kinds["segments"] = kinds.get("segments", 0) + 1
                done += 1
                seed += 1
                continue
            lo = int(rng.integers(0, c["n"] // 3)) if rng.random() < 0.4 else 0
            hi = int(rng.integers(2 * c["n"] // 3, c["n"])) if rng.random() < 0.4 else c["n"]
            lc = bool(rng.random() < 0.5)
            # (a third of the scans return before their kernels have finished -- SPM_SCAN_DEFER -- and are completed by the
            # first accessor; two of them in a row now and then, so that one runs behind an unfinished other)
            fl = spm.SCAN_DEFER if rng.random() < 0.33 else 0
            if fl and rng.random() < 0.5:
                spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=lc, max_hits=1 << 24, flags=fl).close()
            got_h = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_FILTER, left_context=lc, max_hits=1 << 24, flags=fl)
            st = got_h.stats()
            got = got_h.view().copy()
            want = spm.scan(ctx, text, ps, lo, hi, engine=spm.ENGINE_BRUTE, left_context=lc, max_hits=1 << 24).view().copy()
            kind = ("dense" if bs.dense else f"sparse x{bs.passes}") + (" fallback" if st.fallback_spans else "") + (" deferred" if fl else "")
            kinds[kind] = kinds.get(kind, 0) + 1
            if not np.array_equal(got, want):
                bad += 1
                print(f"MISMATCH seed {seed}: {len(got)} vs {len(want)} hits; algo {c['algo']} sigma {c['sigma']} n {c['n']} "
                      f"needles {len(c['needles'])} range [{lo},{hi}) left_context {lc} env {c['env']} kind {kind}", flush=True)
            got_h.close()
        finally:
            for k in c["env"]:
                os.environ.pop(k, None)
        done += 1
        seed += 1
        if done % 20 == 0:
            print(f"{done} cases, {bad} mismatches, kinds {kinds}", flush=True)
    print(f"done: {done} cases, {bad} mismatches, kinds {kinds}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
