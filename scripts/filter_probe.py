"""Filter-kernel timing for a C5-shaped needle set (|P| = 1024, k = 64) over a plain random text, under a few settings."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libspm_amd as S  # noqa: E402

ctx = S.Context(0)
n = int(float(sys.argv[1]) * 2**30) if len(sys.argv) > 1 else 3 << 29
text = ctx.generate(0x5EED0001, 0, n)
needles = [S.synth_pattern(0x5EED0001, 0x5EED0002, n, p, 1024, 64)[0] for p in range(256)]
for env in ({}, {"SPM_HIP_FILTER_QUEUE": "0"}, {"SPM_HIP_FILTER_KEYLEN": "15"}, {"SPM_HIP_FILTER_KEYLEN": "13"},
            {"SPM_HIP_FILTER_THREADS": "256"}, {"SPM_HIP_FILTER_SPANS_PER_WAVE": "8"}):
    os.environ.update(env)
    ps = ctx.patterns(S.ALGO_MYERS, needles, k=64)
    best = None
    for _ in range(4):
        h = S.scan(ctx, text, ps, engine=S.ENGINE_FILTER, max_hits=1 << 22)
        st = h.stats()
        if best is None or st.ms_main < best[0]:
            best = (st.ms_main, st.ms_verify, st.n_candidates, st.n_bands, len(h.view()))
        h.close()
    print(json.dumps({"env": env, "GiB": n / 2**30, "ms_main": round(best[0], 3), "TBps": round(n / best[0] / 1e9, 3),
                      "ms_verify": round(best[1], 3), "cand": int(best[2]), "bands": int(best[3]), "hits": best[4]}))
    ps.close()
    for k in env:
        del os.environ[k]
