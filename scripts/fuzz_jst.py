#!/usr/bin/env python3
"""Differential fuzz of the journaled-sequence (pan-genome) search on the GPU: random reference / alleles (SNPs,
insertions, deletions, replacements, multi-allelic sites) / coverage / needles / window and block lengths; the device
search (context buffer + segmented scan + fan-out, spm_hip_jst_*) against what SURVEY 8(f)-2 defines it to be -- the union
over haplotypes of a linear scan of each materialised haplotype (numpy application of the alleles, brute-force engine).

    python scripts/fuzz_jst.py [--seconds 240] [--seed 1]
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def helpers():
    spec = importlib.util.spec_from_file_location("jst_helpers", os.path.join(ROOT, "tests", "test_gpu_jst.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def one_case(spm, ctx, H, seed):
    rng = np.random.default_rng(seed)
    n_ref = int(rng.choice([6_000, 20_000, 50_000]))
    n_hap = int(rng.choice([1, 2, 7, 33, 64, 65, 130]))
    max_len = int(rng.choice([1, 3, 12, 40, 200]))
    n_var = int(min(rng.choice([0, 5, 60, 400, 1500]), (n_ref - max_len - 4) // (max_len + 2) - 1))
    algo_name = "myers" if rng.random() < 0.8 else "shiftor"
    L = int(rng.choice([24, 40, 64, 100, 200]))
    k = 0 if algo_name == "shiftor" else int(min(rng.choice([0, 1, 2, 3, 8]), L // 12 - 1))
    block = int(rng.choice([0, 128, 256, 1000]))
    ref = rng.integers(0, 4, n_ref, dtype=np.uint8)
    for _ in range(int(rng.integers(0, 6))):       # repeat stretches in the reference
        at = int(rng.integers(0, n_ref - 300))
        ln = int(rng.integers(16, 256))
        ref[at:at + ln] = np.resize(rng.integers(0, 4, int(rng.integers(1, 5)), dtype=np.uint8), ln)
    ref_text = ctx.upload(ref)
    if n_var > 0:
        alleles, pool, cov = H._random_alleles(rng, n_ref, n_hap, n_var, max_len)
    else:
        alleles = np.zeros(0, dtype=[("pos", "<u8"), ("ref_len", "<u4"), ("alt_len", "<u4"), ("alt_off", "<u8")])
        pool, cov = np.zeros(0, np.uint8), np.zeros((0, (n_hap + 63) // 64), dtype=np.uint64)
    jst = spm.Jst(ctx, ref_text, alleles, pool, cov, n_hap)
    haps = [H._apply(ref, alleles, pool, cov, h) for h in range(n_hap)]
    if min(len(h) for h in haps) <= L + 8:
        return "skipped"
    needles = H._needles_from(rng, haps, int(rng.choice([1, 8, 40])), L, k)
    algo = spm.ALGO_MYERS if algo_name == "myers" else spm.ALGO_SHIFTOR
    ps = ctx.patterns(algo, needles, k=k)
    window = max(ps.window_size(p) for p in range(len(needles))) + int(rng.choice([0, 0, 5, 40]))
    exp = H._expected(spm, ctx, haps, ps, spm.ENGINE_BRUTE)
    jst.index(window, block)
    got = H._got(jst.search(ps, engine=spm.ENGINE_AUTO, max_hits=1 << 22))
    ok = got == exp
    if not ok:
        print(f"MISMATCH seed {seed}: {len(got)} vs {len(exp)} records; n_ref {n_ref} haplotypes {n_hap} alleles {len(alleles)} "
              f"max_len {max_len} {algo_name} |P| {L} k {k} window {window} block {block}", flush=True)
    jst.close()
    ps.close()
    ref_text.close()
    one_case.records += len(exp)
    return "ok" if ok else "bad"


one_case.records = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    torch.zeros(1, device="cuda")
    import libspm_amd as spm
    ctx = spm.Context(0)
    H = helpers()
    t_end = time.time() + args.seconds
    seed, n, bad, skipped = args.seed, 0, 0, 0
    while time.time() < t_end:
        r = one_case(spm, ctx, H, seed)
        bad += r == "bad"
        skipped += r == "skipped"
        n += 1
        seed += 1
        if n % 20 == 0:
            print(f"{n} cases, {bad} mismatches, {skipped} skipped, {one_case.records} records compared", flush=True)
    print(f"done: {n} cases, {bad} mismatches, {skipped} skipped, {one_case.records} records compared")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
