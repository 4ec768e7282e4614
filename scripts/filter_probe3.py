"""Masked-key (H < 16) against full-key filter kernels on the same needle set and stride (1.5 GiB random text)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libspm_amd as S  # noqa: E402

ctx = S.Context(0)
n = 3 << 29
text = ctx.generate(0x5EED0001, 0, n)
for n_pat, k in ((256, 30),):
    needles = [S.synth_pattern(0x5EED0001, 0x5EED0002, n, p, 1024, k)[0] for p in range(n_pat)]
    for env in ({"SPM_HIP_FILTER_STRIDE": "2"}, {"SPM_HIP_FILTER_STRIDE": "2", "SPM_HIP_FILTER_FORCE_MASKED": "1"},
                {"SPM_HIP_FILTER_STRIDE": "2", "SPM_HIP_FILTER_KEYLEN": "15"},
                {"SPM_HIP_FILTER_STRIDE": "2", "SPM_HIP_FILTER_KEYLEN": "14"}):
        os.environ.update(env)
        ps = ctx.patterns(S.ALGO_MYERS, needles, k=k)
        best = None
        for _ in range(4):
            h = S.scan(ctx, text, ps, engine=S.ENGINE_FILTER, max_hits=1 << 22)
            st = h.stats()
            if best is None or st.ms_main < best[0]:
                best = (st.ms_main, st.n_candidates, st.main_launches)
            h.close()
        print(json.dumps({"needles": n_pat, "env": env, "ms_main": round(best[0], 3), "TBps": round(n / best[0] / 1e9, 3),
                          "cand": int(best[1]), "launches": int(best[2])}))
        ps.close()
        for kk in env:
            del os.environ[kk]
