"""Throughput of the seed filter on a dna5 haystack (A0 C1 G2 N3 T4, 0.1 % N) wrapped from a torch tensor."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libspm_amd as S
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
n = int(gib * 2**30)
dev = torch.device("cuda", 0)
lut = torch.tensor([0, 1, 2, 4], dtype=torch.uint8, device=dev)
t = torch.empty(n, dtype=torch.uint8, device=dev)
step = 1 << 28
for a in range(0, n, step):
    b = min(n, a + step)
    r = torch.randint(0, 4, (b - a,), dtype=torch.uint8, device=dev)
    x = lut[r.long()]
    x[torch.rand(b - a, device=dev) < 1e-3] = 3
    t[a:b] = x
torch.cuda.synchronize()
ctx = S.Context(0)
text = ctx.wrap(t.data_ptr(), n, sigma=5, keepalive=t)
rng = np.random.default_rng(1)
needles = []
for i in range(1024):
    at = int(rng.integers(0, n - 200))
    nd = t[at:at + 100].cpu().numpy().copy()
    nd[nd == 3] = 0
    needles.append(nd)
ps = ctx.patterns(S.ALGO_MYERS, needles, k=3, sigma=5)
assert ps.filterable
ms = []
for i in range(8):
    h = S.scan(ctx, text, ps)
    st = h.stats()
    if i >= 2:
        ms.append(st.ms_main)
    hits = st.n_hits
print(json.dumps({"dna5_text_gib": gib, "kernel_ms": float(np.mean(ms)), "GBps": n / np.mean(ms) / 1e6, "hits": int(hits)}))
