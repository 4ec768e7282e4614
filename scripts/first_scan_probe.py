"""First scan of a fresh needle set vs the later ones: host wall clock and the device-side split (events)."""
import time, sys, numpy as np, torch
torch.zeros(1, device="cuda")
sys.path.insert(0, ".")
import libspm_amd as S
ctx = S.Context(0)
n = 1 << 30
text = ctx.generate(0x5EED0001, 0, n)
for trial in range(3):
    needles = np.stack([S.synth_pattern(0x5EED0001, 0x5EED0002 + trial, n, p, 32, 0)[0] for p in range(1024)])
    ps = ctx.patterns(S.ALGO_SHIFTOR, needles, k=0)
    ctx.synchronize()
    ts = []
    for i in range(4):
        t0 = time.perf_counter()
        h = S.scan(ctx, text, ps, max_hits=1 << 20)
        n_hits = len(h.view())
        ts.append((time.perf_counter() - t0) * 1e3)
        st = h.stats()
        gs = (round(st.ms_total, 3), round(st.ms_main, 3), round(st.ms_verify, 3))
        ts.append(gs)
        h.close()
    print("needle set", trial, "host ms / (gpu total, main, verify):", [x if isinstance(x, tuple) else round(x, 3) for x in ts], n_hits)
