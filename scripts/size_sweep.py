"""Scan time vs haystack size for the C3 needle set (1024 x |P|=100, k<=3): seed filter on the 1-byte text, on the
2-bit shadow, and the brute-force engine (small sizes only)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import libspm_amd as S

ctx = S.Context(0)
N = 1 << 34
text = ctx.generate(0x5EED0001, 0, N)
needles = [S.synth_pattern(0x5EED0001, 0x5EED0002, N, p, 100, 3)[0] for p in range(1024)]
ps = ctx.patterns(S.ALGO_MYERS, needles, k=3)
rows = []
def timed(n, reps, **kw):
    for _ in range(2):
        S.scan(ctx, text, ps, 0, n, **kw).close()
    ctx.synchronize()
    t0 = time.perf_counter()
    ms_k = 0.0
    for _ in range(reps):
        h = S.scan(ctx, text, ps, 0, n, **kw)
        st = h.stats()
        ms_k += st.ms_main
        h.close()
    return (time.perf_counter() - t0) / reps * 1e3, ms_k / reps, int(st.n_hits)
for lg in (20, 22, 24, 26, 28, 30, 32, 34):
    n = 1 << lg
    wall, kern, hits = timed(n, 20 if lg < 32 else 10, engine=S.ENGINE_FILTER, flags=S.capi.SCAN_IGNORE_PACKED)
    row = {"log2_n": lg, "filter_ms": wall, "filter_kernel_ms": kern, "filter_Gbases_s": n / wall / 1e6, "hits": hits}
    if lg <= 28:
        bw, bk, bh = timed(n, 3, engine=S.ENGINE_BRUTE)
        row.update({"brute_ms": bw, "brute_Gbases_s": n / bw / 1e6, "brute_hits_equal": bh == hits})
    rows.append(row)
text.pack()
for row in rows:
    n = 1 << row["log2_n"]
    wall, kern, hits = timed(n, 20 if row["log2_n"] < 32 else 10, engine=S.ENGINE_FILTER)
    row.update({"packed_ms": wall, "packed_kernel_ms": kern, "packed_Gbases_s": n / wall / 1e6, "packed_hits_equal": hits == row["hits"]})
    print(json.dumps(row))
