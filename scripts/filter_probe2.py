"""Filter-kernel time against stride for 16-symbol keys (|P| = 1024, k = 30 -> q = 33) on a 1.5 GiB random text."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libspm_amd as S  # noqa: E402

ctx = S.Context(0)
n = 3 << 29
text = ctx.generate(0x5EED0001, 0, n)
for n_pat in (256, 16):
    needles = [S.synth_pattern(0x5EED0001, 0x5EED0002, n, p, 1024, 30)[0] for p in range(n_pat)]
    for stride in (16, 8, 4, 2, 1):
        os.environ["SPM_HIP_FILTER_STRIDE"] = str(stride)
        ps = ctx.patterns(S.ALGO_MYERS, needles, k=30)
        best = None
        for _ in range(4):
            h = S.scan(ctx, text, ps, engine=S.ENGINE_FILTER, max_hits=1 << 22)
            st = h.stats()
            if best is None or st.ms_main < best[0]:
                best = (st.ms_main, st.ms_verify, st.n_candidates, st.n_bands, st.main_launches)
            h.close()
        print(json.dumps({"needles": n_pat, "stride": stride, "ms_main": round(best[0], 3), "TBps": round(n / best[0] / 1e9, 3),
                          "ms_verify": round(best[1], 3), "cand": int(best[2]), "bands": int(best[3]), "launches": int(best[4])}))
        ps.close()
        del os.environ["SPM_HIP_FILTER_STRIDE"]
