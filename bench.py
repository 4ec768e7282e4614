#!/usr/bin/env python3
"""bench.py -- Gbases/s scanned by the matcher hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input: one scan of the rank's text shard against the
whole needle set (seqan_pattern_base::operator() for every matcher of the set), hits compacted in HBM, and -- for
N > 1 -- the RCCL gatherv of the hit records to rank 0.  The text is resident in HBM before the timed region starts.

Default workload = BASELINE.json configs[2], the one the metric is quoted on:
    Myers bit-vector k<=3, 1 024 needles |P|=100, 16 GiB of uniform dna4 text per MI355X.
N > 1 is weak scaling: the global text is N x 16 GiB, rank g scans shard g (plus window_size-1 symbols of left
context); the needles are planted anywhere in the global text.

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c3r|c2|c4|c5|reads100] [--text-gib G] [--engine auto|brute|filter]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED_TEXT = 0x5EED0001
SEED_PAT = 0x5EED0002
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)
HBM_STREAM_PROBE_GBS = 7030.0  # best pure streaming read of a 16 GiB buffer on this GPU model (tools/hbm_read_probe,
#                                profiles/r01/hbm_read_probe_16GiB.log); reported beside `frac`, never instead of it

WORKLOADS = {
    # name: (algo, |P|, kmax, needles, text GiB per GPU, description)
    "c3": ("myers", 100, 3, 1024, 16.0, "Myers k<=3, 1024 needles |P|=100, 16 GiB dna4 text per GPU"),
    "c2": ("shiftor", 32, 0, 1024, 1.0, "Shift-Or exact, 1024 needles |P|=32, 1 GiB dna4 text per GPU"),
    "c4": ("myers", 150, 3, 100000, 8.0, "Myers k<=3, 100k needles |P|=150, 8 GiB dna4 text per GPU (64 GiB on 8)"),
    # the read set the reference's authors set up for their own benchmark (test/data/datasources.cmake:181-183: 100 k
    # simulated reads of 100 nt, 3 errors); same per-GPU text as C4
    "reads100": ("myers", 100, 3, 100000, 8.0, "Myers k<=3, 100k needles |P|=100 (the reference's simulated read set shape), "
                                               "8 GiB dna4 text per GPU"),
    # C3 on a repeat-rich text: --repeat-frac of the bases inside tandem-repeat / low-complexity stretches, every 8th
    # needle cut across one (what the q-gram filter meets on real genomes; VERDICT r01 "next" 1)
    "c3r": ("myers", 100, 3, 1024, 16.0, "Myers k<=3, 1024 needles |P|=100, 16 GiB dna4 text per GPU with repeat "
                                         "stretches"),
    # journaled-sequence pan-genome: text GiB = REFERENCE bases per GPU (2^27; 2^30 on 8), 64 haplotypes over it
    "c5": ("myers", 1024, 64, 256, 0.125, "multi-word Myers |P|=1024 k<=64, 256 needles, journaled pan-genome: "
                                          "64 haplotypes over a 2^27-base reference per GPU (2^30 on 8)"),
}
SEED_VAR = 0x5EED0003


VALU_PEAK_LANE_OPS = 6.5e13   # measured: v_add_u32 / v_bitop3_b32 chains, 8 waves per SIMD (tools/valu_probe.hip)
VALU_NOMINAL_LANE_OPS = 256 * 128 * 2.4e9


def live_stream_probe():
    """Best pure streaming read of a 16 GiB buffer on THIS GPU, measured now by tools/hbm_read_probe (a standalone HIP
    program built by build(); run as a child process once every buffer of the bench is released).  None if absent."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "hbm_read_probe")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=120).stdout
        best = 0.0
        for line in out.splitlines():
            f = line.split()
            if len(f) >= 8 and f[0] in ("span", "inter"):
                best = max(best, float(f[-1]))
        return best or None
    except Exception:
        return None


def valu_roofline(workload, n_pat, lane_steps_per_s):
    """SURVEY 8(d)(ii) for the brute-force engine: lane-ops/s against the measured integer-VALU issue peak.

    VALU instructions per lane-step come from a PMC pass over the same kernel (SQ_INSTS_VALU / wave-steps,
    profiles/brute_valu.json, written by scripts/collect_profiles.py); None if that pass was never collected."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "brute_valu.json")))[workload]
        if pm["needles"] != n_pat:
            return None
    except Exception:
        return None
    ops = lane_steps_per_s * pm["valu_per_lane_step"]
    return {"kernel": pm["kernel"], "valu_per_lane_step": pm["valu_per_lane_step"], "lane_ops_per_s": ops,
            "peak_measured": VALU_PEAK_LANE_OPS, "frac_of_measured": ops / VALU_PEAK_LANE_OPS,
            "peak_nominal": VALU_NOMINAL_LANE_OPS}


def edit_needle(src, L, e, seed):
    """Needle of length L from src (>= L + e symbols): e edits at pseudo-random places (substitute / delete / insert),
    trimmed back to L -- the scheme of SURVEY 8(d), applied to a haplotype window."""
    import libspm_amd as S
    out = [int(x) for x in src[:L + e]]
    for j in range(e):
        r = S.capi.lib().spm_hip_mix64(seed + j + 1)
        at = r % L
        kind = (r >> 32) % 3
        if kind == 0:
            out[at] = (out[at] + 1 + (r >> 40) % 3) & 3
        elif kind == 1:
            del out[at]
        else:
            out.insert(at, (r >> 40) & 3)
    return np.array(out[:L], dtype=np.uint8)


def run_c5(args, S, sdist, torch, dist, rank, world, dev, ctx):
    """Config C5: needles against every haplotype of a journaled sequence tree.  One reference
    chromosome per GPU (weak scaling), its context index built once on the device; a step = one search of the whole
    needle set over all haplotypes (segment scan of the context buffer + verification + fan-out) + the gatherv."""
    algo, L, kmax, n_pat, gib, desc = WORKLOADS["c5"]
    if args.text_gib is not None:
        gib = args.text_gib
    if args.needles is not None:
        n_pat = args.needles
    n_hap = 64
    # chromosome length: whole variant blocks (10 000) and whole KiB, so rank * ref_len is a valid generator offset
    ref_len = max(640000, int(gib * 2**30) // 640000 * 640000)
    engine = {"auto": S.ENGINE_AUTO, "brute": S.ENGINE_BRUTE, "filter": S.ENGINE_FILTER}[args.engine]
    mix = S.capi.lib().spm_hip_mix64
    ref = ctx.generate(SEED_TEXT, rank * ref_len, ref_len)
    t0 = time.perf_counter()
    alleles, pool, cov = S.synth_variants(SEED_TEXT, SEED_VAR, rank * ref_len, ref_len, n_hap)
    jst = S.Jst(ctx, ref, alleles, pool, cov.reshape(-1, 1), n_hap)
    t_create = time.perf_counter() - t0
    window = L + kmax

    # needles: p is cut from chromosome p % world, haplotype mix(p) % 64, with p % (kmax + 1) edits; every rank
    # spells out its own and the set is completed with one all-reduce (setup, not timed)
    mine = np.zeros((n_pat, L), dtype=np.int32)
    planted = {}
    for p in range(rank, n_pat, world):
        r = mix(SEED_PAT + 7919 * p)
        h = r % n_hap
        o = (r >> 8) % (jst.haplotype_length(h) - 2 * (L + kmax))
        e = p % (kmax + 1)
        mine[p] = edit_needle(jst.extract(h, o, L + kmax), L, e, SEED_PAT ^ (p << 20))
        planted[p] = (h, o, e)
    if world > 1:
        t = torch.from_numpy(mine)
        if dist.get_backend() != "gloo":
            t = t.to(dev)
        dist.all_reduce(t)
        mine = t.cpu().numpy()
    needles = [mine[p].astype(np.uint8) for p in range(n_pat)]
    ps = ctx.patterns(S.ALGO_MYERS, needles, k=kmax)

    # engines agree on a slice of the tree (the brute-force engine is too slow for all of it)
    check = None
    if rank == 0 and args.brute_sample_mib > 0:
        jst.index(window, 1024, 0, 2048)
        a = jst.search(ps, engine=engine, max_hits=1 << 22)
        b = jst.search(ps, engine=S.ENGINE_BRUTE, max_hits=1 << 22)
        check = {"blocks": 2048, "hits": int(len(a)), "hits_equal_to_default_engine": bool(np.array_equal(a, b))}
    st = jst.index(window, 1024)
    max_hits = 1 << 23
    cap = max_hits
    og = sdist.OverlappedGather(dev, cap, 3)   # N > 1: the records of search i travel while search i + 1 runs
    n_step = [0]

    def step():
        i = n_step[0]
        n_step[0] += 1
        buf = og.buffer(i)
        h = jst.search_device(ps, engine=engine, max_hits=max_hits)
        n = h.copy_to(buf.data_ptr(), cap)
        gathered = og.submit(i, n)                  # rank 0: every rank's records, rank order
        return h, gathered, n

    for _ in range(args.warmup):
        h, g, n = step()
        h.close()
    og.finish()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_main = ms_verify = ms_fan = 0.0
    launches = 0
    last = None
    for _ in range(args.steps):
        h, g, n = step()
        s1 = jst.stats()
        ms_main += s1.ms_main
        ms_verify += s1.ms_verify
        ms_fan += s1.ms_fanout
        launches += s1.main_launches
        if last is not None:
            last[0].close()
        last = (h, g, n, s1)
    og.finish()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    h, gathered, n_local, s1 = last
    # every planted needle must be reported on its source haplotype where it was cut
    rec = h.view()
    h.close()
    found = 0
    by_needle = {}
    for q in np.nonzero(np.isin(rec["pattern"], list(planted)))[0] if planted else []:
        by_needle.setdefault(int(rec["pattern"][q]), []).append((int(rec["haplotype"][q]), int(rec["pos"][q]),
                                                                 int(rec["score"][q])))
    for p, (hh, o, e) in planted.items():
        if any(a == hh and abs(b - (o + L)) <= kmax and c <= e for a, b, c in by_needle.get(p, [])):
            found += 1
    tot = torch.tensor([found, int(st.haplotype_symbols), int(st.context_symbols), int(st.unique_contexts),
                        int(st.contexts)], dtype=torch.int64, device=dev)
    if world > 1:
        if dist.get_backend() == "gloo":
            tc = tot.cpu()
            dist.all_reduce(tc)
            tot = tc
        else:
            dist.all_reduce(tot)
    found, hap_sym, ctx_sym, uniq, nctx = [int(x) for x in tot.tolist()]
    if rank != 0:
        return None
    step_s = dt / args.steps
    k_ms = ms_main / max(launches, 1)
    engine_used = {1: "brute", 2: "filter"}.get(int(s1.engine_used), "?")
    achieved = st.context_symbols / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    result = {
        "metric": "Gbases/s of haplotype sequence searched, multi-word Myers |P|=1024 k<=64 over a journaled pan-genome",
        "value": hap_sym / step_s / 1e9,
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": step_s * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"c5: {desc}", "needles": n_pat, "needle_len": L, "k": kmax, "haplotypes": n_hap,
                   "reference_bases_per_gpu": ref_len, "alleles_per_gpu": int(len(alleles)), "block_len": 1024,
                   "engine": engine_used,
                   "sharding": (f"one reference chromosome per GPU, {world} GPUs; hit records gathered to rank 0 "
                                "(count all-gather + grouped send/recv on a side stream: the records of search i "
                                "travel while search i + 1 runs)") if world > 1 else "single GPU",
                   "exchange_bytes_per_rank_and_step": int(n_local) * 24},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": c5_traffic(int(st.context_symbols), engine_used),
            # (stride 2 with presence bits as level 1 runs as an instance of the dense kernel: filter.hpp, dense_group S = 2)
            "kernel": "seed_filter_dense_kernel<4, 1, 2, true>" if engine_used == "filter" else "myers_cutoff_kernel",
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": int(st.context_symbols),
            "note": "algorithmic bytes = the deduplicated context buffer this GPU streams per search (1 byte per "
                    "symbol); the haplotype symbols it stands for are `value`",
        },
        "hits": int(gathered.shape[0]) if gathered is not None else int(n_local),
        "needles_found_on_their_haplotype": found,
        "all_planted_found": bool(found == n_pat),
        "journaled_sequence_tree": {
            "haplotype_symbols": hap_sym, "context_symbols": ctx_sym, "sharing": hap_sym / max(ctx_sym, 1),
            "contexts": nctx, "unique_contexts": uniq, "index_ms_rank0": st.ms_index, "create_s_rank0": t_create,
            "segment_hits_rank0": int(s1.segment_hits),
            "reference_Gbases_per_s": ref_len * world / step_s / 1e9,
            "context_Gbases_per_s": ctx_sym / step_s / 1e9,
        },
        "verify_ms_per_step": ms_verify / args.steps,
        "fanout_ms_per_step": ms_fan / args.steps,
        "candidates": int(s1.candidates),
        "bands_verified": int(s1.bands),
        "fell_back": int(s1.fell_back),
        "lane_steps_per_s": n_pat * hap_sym / step_s,
        "reference_equivalent_traffic_GBps": n_pat * hap_sym / step_s / 1e9,
    }
    if check is not None:
        result["brute_force_engine"] = check
    if world == 1 and not args.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline("myers", L, kmax, n_pat, ref_len)
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": 0, "kind": "port",
                                      "sample": f"unavailable: {e}"}
    return result


def c5_traffic(context_symbols, engine_used):
    """HBM bytes per launch of C5's streaming kernel from the PMC passes (profiles/pmc_traffic.json), if they were taken on
    this exact context buffer."""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get("c5")
        if pm and pm.get("algorithmic_bytes_per_launch") == context_symbols and engine_used == "filter":
            return pm["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(algo, L, kmax, n_pat_full, n_total, budget_s=15.0):
    """The oracle (CPU restatement of the reference path: one matcher per needle, one pass per matcher) timed on
    the host cores, on a bounded sample of the same workload.  A reported baseline, not the target."""
    from oracle import oracle as O
    native = O.use_native_build()  # -O3 -march=native on this host (oracle/Makefile, target `native`)
    try:  # the CPUs this process may actually run on (a GPU box hands a job a share of its cores)
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    host = host_cpu_facts()
    if host.get("cgroup_quota_cpus"):  # a cgroup CPU quota: more threads than that only take turns
        cores = min(cores, max(1, int(host["cgroup_quota_cpus"] + 0.999)))
    cores = max(1, min(cores, 64))
    o_algo = O.MYERS if algo == "myers" else O.SHIFTOR
    pats = [O.pattern(SEED_TEXT, SEED_PAT, n_total, p, L, kmax)[0] for p in range(64)]
    # calibrate on one core
    t_small = O.text(SEED_TEXT, 0, 1 << 21)
    t0 = time.perf_counter()
    O.scan_multi(o_algo, t_small, pats[:4], k=kmax, threads=1)
    rate1 = 4 * len(t_small) / max(time.perf_counter() - t0, 1e-6)  # lane-steps/s on one core
    n_sample = int(min(max(budget_s * rate1 * cores / 64, 1 << 22), 1 << 28)) & ~1023
    text = O.text(SEED_TEXT, 0, n_sample)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        hits = O.scan_multi(o_algo, text, pats, k=kmax, threads=cores)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    lane_steps = 64 * n_sample / best
    return {
        "value": lane_steps / n_pat_full / 1e9,
        "unit": "Gbases/s",
        "cores": cores,   # threads used = CPUs this process may run on (affinity, capped by the cgroup CPU quota), at most 64
        # how many cores' worth of work those threads got: all-threads rate / one-thread rate (SMT siblings and a cgroup
        # CPU quota both show up here); `value` derives from the measured all-threads rate only
        "effective_cores": lane_steps / rate1,
        "host": host,
        "kind": "port",
        "lane_steps_per_s": lane_steps,
        # one needle, one thread, one symbol: the serial VP/VN dependency chain of the bit-vector recurrence
        "ns_per_symbol_step_per_thread": 1e9 / rate1,
        "ns_per_symbol_step_per_thread_all_cores_busy": cores * 1e9 / lane_steps,
        "build": "-O3 -march=native" if native else "-O3 (portable build; the native rebuild failed)",
        "loop": "two 64-bit blocks with Ukkonen cut-off, state in registers (oracle/spm_oracle.c: "
                "spm_oracle_myers2_fast)" if algo == "myers" and 64 < L <= 128 else "generic block loop",
        "sample": f"64 needles x {n_sample / 2**20:.0f} MiB of the same synthetic text, one sequential pass per "
                  f"needle over {cores} threads, best of 2; value = measured lane-steps/s / {n_pat_full} needles "
                  f"(linear extrapolation to the full needle set); {len(hits)} hits",
    }


def host_cpu_facts():
    """Physical cores, hardware threads and the cgroup CPU quota of this host, as far as /proc and /sys tell."""
    facts = {"logical_cpus": os.cpu_count()}
    try:
        facts["affinity_cpus"] = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        phys = set()
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        if phys:
            facts["physical_cores"] = len(phys)
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                facts["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    facts["cgroup_quota_cpus"] = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                facts["cgroup_quota_cpus"] = q / per if q > 0 else None
            break
        except Exception:
            continue
    return facts


def self_launch(argv, n):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as fresh child processes (this
    parent has not touched the GPU or imported torch), relay rank 0's JSON line and the exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--text-gib", type=float, default=None, help="text per GPU in GiB (default: the workload's)")
    ap.add_argument("--needles", type=int, default=None)
    ap.add_argument("--engine", default="auto", choices=["auto", "brute", "filter"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--brute-sample-mib", type=int, default=256)
    ap.add_argument("--packed-steps", type=int, default=10, help="steps of the packed-shadow variant (0 = skip)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default single-GPU C3 run only: skip the C2 / C4 / C5 (/ c3r) lines attached as other_configs")
    ap.add_argument("--repeat-frac", type=float, default=0.01, help="c3r: fraction of the text inside repeat stretches")
    ap.add_argument("--repeat-needle-every", type=int, default=64,
                    help="c3r: besides the needles that cross a stretch by chance (uniform cut positions: ~1.7 %% of them at "
                         "1 %% repeats), every N-th needle is cut across a stretch on purpose (0: none)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    import libspm_amd as S
    from libspm_amd import dist as sdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    backend = os.environ.get("SPM_BENCH_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on one GPU
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = S.Context(local_rank, stream=stream.cuda_stream)
        env = (S, sdist, torch, dist, rank, world, dev, ctx)
        if args.workload == "c5":
            result = run_c5(args, *env)
        else:
            result = run_scan(args, *env, workload=args.workload)
        # One driver-timed line that covers every BASELINE config: the default run (C3, one GPU) also executes C2, C4's
        # per-GPU shard, C5 and the repeat-rich C3 text, each >= 3 timed repetitions (test/benchmark/CMakeLists.txt:12-14)
        default_run = (rank == 0 and world == 1 and args.workload == "c3" and not args.no_other_configs
                       and args.engine == "auto" and args.text_gib is None and args.needles is None)
        if default_run:
            result["other_configs"] = other_configs(args, env)
    if default_run:
        # what this GPU streams at best (measured now, not a tracked constant); `frac` stays against the 8 TB/s spec peak
        ctx.close()
        torch.cuda.empty_cache()
        probe = live_stream_probe()
        rf = result["roofline"]
        if probe:
            rf["stream_read_probe"] = probe
            rf["frac_of_stream_read_probe"] = rf["achieved"] / probe
            rf["stream_read_probe_source"] = "measured in this run (tools/hbm_read_probe, 16 GiB, best access shape)"
        else:
            rf["stream_read_probe_source"] = "profiles/r01/hbm_read_probe_16GiB.log (tools/hbm_read_probe not built)"
        rf["traffic_source"] = "rocprofv3 PMC passes of the same workload (profiles/pmc_traffic.json); not measured in this run"
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            algo, L, kmax, n_pat, gib, _ = WORKLOADS[args.workload]
            n_pat = args.needles if args.needles is not None else n_pat
            n_total = (int((args.text_gib if args.text_gib is not None else gib) * 2**30) & ~1023) * world
            try:
                result["cpu_baseline"] = cpu_baseline(algo, L, kmax, n_pat, n_total)
            except Exception as e:  # the oracle is test infrastructure; its absence must not hide the GPU number
                result["cpu_baseline"] = {"value": None, "unit": "Gbases/s", "cores": 0, "kind": "port",
                                          "sample": f"unavailable: {e}"}
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


def other_configs(args, env):
    """C2, C4 (its per-GPU 8 GiB shard), C5, c3r and reads100 on this GPU, condensed to what VERDICT r01 #2 asks for."""
    import copy
    out = {}
    # (c4 / reads100: their VALU-bound kernel follows the engine clock, which ramps over the first launches -- 3.8 -> 3.35 ms
    # over five: three warm-up scans, five timed)
    for name, steps, warmup in (("c2", 20, 3), ("c4", 5, 3), ("c5", 10, 2), ("c3r", 5, 2), ("reads100", 5, 3)):
        a = copy.copy(args)
        a.workload, a.steps, a.warmup = name, steps, warmup
        a.no_cpu_baseline, a.brute_sample_mib, a.packed_steps = True, 0, 0
        t0 = time.perf_counter()
        try:
            r = run_c5(a, *env) if name == "c5" else run_scan(a, *env, workload=name)
        except Exception as e:  # one config failing must not hide the others (nor the headline)
            out[name] = {"error": f"{type(e).__name__}: {e}"}
            continue
        rf = r["roofline"]
        o = {"metric": r["metric"], "value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"],
             "steps": steps, "warmup": warmup, "workload": r["config"]["workload"],
             "roofline": {k: rf[k] for k in ("achieved", "frac", "kernel", "kernel_ms", "launches_per_step",
                                             "algorithmic_bytes_per_launch") if k in rf},
             "hits": r["hits"], "all_planted_found": r["all_planted_found"], "fell_back": r["fell_back"],
             "candidates": r["candidates"], "verify_ms_per_step": r["verify_ms_per_step"],
             "setup_and_run_s": time.perf_counter() - t0}
        for k in ("patterns_create_ms", "patterns_create", "first_scan_ms"):
            if k in r:
                o[k] = r[k]
        if "traffic" in rf and rf["traffic"] is not None:
            o["roofline"]["traffic"] = rf["traffic"]
        for k in ("fallback_spans", "repeat_text", "needles_found_on_their_haplotype", "journaled_sequence_tree",
                  "fanout_ms_per_step", "parity_slice"):
            if k in r:
                o[k] = r[k]
        out[name] = o
    return out


def run_scan(args, S, sdist, torch, dist, rank, world, dev, ctx, workload):
    """Configs C2 / C3 / C4 (and c3r): one scan of this rank's text shard against the whole needle set per step."""
    algo, L, kmax, n_pat, gib, desc = WORKLOADS[workload]
    if args.text_gib is not None:
        gib = args.text_gib
    if args.needles is not None:
        n_pat = args.needles
    per_gpu = int(gib * 2**30) & ~1023
    n_total = per_gpu * world
    window = L + kmax
    sp = sdist.ShardPlan(n_total, rank, world, window)   # owned range, left context, global coordinates (tests/test_dist_gloo.py)
    lo, hi, ovl = sp.lo, sp.hi, sp.ovl
    repeats = workload == "c3r"
    rep_ppm = int(round(args.repeat_frac * 1e6)) if repeats else 0

    if repeats:
        # uniform text with a stated fraction of tandem-repeat / low-complexity stretches, 1/8 of the needles cut across
        # them (libspm_amd/csrc/synth.hpp: repeat_base / synth_repeat_pattern; the oracle regenerates any slice)
        text = ctx.generate_repeats(SEED_TEXT, lo - ovl, (hi - lo) + ovl, rep_ppm)
        needles = [S.synth_repeat_pattern(SEED_TEXT, SEED_PAT, n_total, p, L, kmax, rep_ppm, args.repeat_needle_every)[0]
                   for p in range(n_pat)]
    else:
        text = ctx.generate(SEED_TEXT, lo - ovl, (hi - lo) + ovl)
        needles = [S.synth_pattern(SEED_TEXT, SEED_PAT, n_total, p, L, kmax)[0] for p in range(n_pat)]
    s_algo = S.ALGO_MYERS if algo == "myers" else S.ALGO_SHIFTOR
    needles = np.stack(needles)     # reads of one length: one row each (the layout the C ABI takes; no per-needle Python work)
    # matcher construction (the reference: one constructor per needle, myers_matcher.hpp:40-43): host wall clock of the
    # call that builds every table of the set and uploads it -- not part of a step, reported beside it
    torch.cuda.synchronize()
    t_create = time.perf_counter()
    ps = ctx.patterns(s_algo, needles, k=kmax)
    ctx.synchronize()
    patterns_create_ms = (time.perf_counter() - t_create) * 1e3
    bs = ps.build_stats()
    engine = {"auto": S.ENGINE_AUTO, "brute": S.ENGINE_BRUTE, "filter": S.ENGINE_FILTER}[args.engine]
    # how the records travel: one fused all-gather of fixed-size buffers (C2, C3) or count + send/recv (C4, c3r)
    xp = sdist.ExchangePlan(n_pat, many_hits=repeats)
    max_hits, cap, fused = xp.max_hits, xp.cap, xp.fused
    hit_buf = xp.new_buffer(dev)

    def step():
        # SCAN_DEFER: a scan that cannot need a second attempt on the device side alone (exact sets: C2) returns once its
        # kernels are enqueued; with the device-side fused copy the step then has no host synchronisation at all, and its
        # statistics are read one step later (below) -- the GPU never waits for the host between steps
        h = S.scan(ctx, text, ps, sp.scan_begin, sp.scan_end, engine=engine, left_context=True,
                   pos_offset=sp.pos_offset, max_hits=max_hits, flags=S.SCAN_DEFER if fused else 0)
        if fused:
            h.copy_fused_device(hit_buf.data_ptr(), cap)     # [count, status | records], written on the device
            n = 0
        else:
            n = h.copy_to(hit_buf[1:].data_ptr(), cap)      # records (D2D on this stream)
        return h, xp.exchange(hit_buf, n)                    # N > 1: the collective; N = 1: a view

    # the first scan of a fresh needle set (cold buffers, nothing learnt about the text yet): what a one-shot
    # `matcher(haystack, callback)` waits for
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    h, g = step()
    torch.cuda.synchronize()
    first_scan_ms = (time.perf_counter() - t_first) * 1e3
    h.close()
    for _ in range(max(args.warmup - 1, 0)):
        h, g = step()
        h.close()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ms_main = ms_verify = 0.0
    launches = 0
    last = None
    prev = None
    for _ in range(args.steps + 1):
        cur = step() if _ < args.steps else None
        if not fused:                   # this step has synchronised with the host already (its count): read it now, so
            prev, cur = cur, None       # that its buffers (1 GiB of hit slots on a repeat-rich text) serve the next step
        if prev is not None:            # fused: the step before -- its kernels are done or running behind this step's
            h, g = prev
            st = h.stats()
            ms_main += st.ms_main
            ms_verify += st.ms_verify
            launches += st.main_launches
            last = (st, g)
            h.close()
        prev = cur
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    st, gathered = last
    if rank != 0:
        return None
    recs = xp.records(gathered).cpu().numpy().view(np.uint8).reshape(-1, 16)
    hits = np.frombuffer(recs.tobytes(), dtype=S.HIT_DTYPE)
    found = np.unique(hits["pattern"])
    ms_per_step = dt / args.steps * 1e3
    value = n_total / (dt / args.steps) / 1e9
    # average duration of one launch of the dominant kernel (HIP events on the scan's stream).  A needle set that
    # outgrows one LDS table runs several passes per scan (C4), each streaming the whole shard
    k_ms = ms_main / max(launches, 1)
    achieved = (hi - lo) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    engine_used = {1: "brute", 2: "filter"}.get(int(st.engine_used), "?")
    # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs, corrected as the MI355X guide
    # prescribes; profiles/pmc_traffic.json says how) -- only when it was measured on this exact config
    traffic = None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(workload)
        if pm and pm["text_bytes_per_gpu"] == hi - lo and engine_used == "filter":
            traffic = pm["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    result = {
        "metric": {"c3": "Gbases/s scanned, Myers k<=3 |P|=100", "c2": "Gbases/s scanned, Shift-Or |P|=32",
                   "c4": "Gbases/s scanned, Myers k<=3 |P|=150, 100k needles",
                   "reads100": "Gbases/s scanned, Myers k<=3 |P|=100, 100k needles",
                   "c3r": "Gbases/s scanned, Myers k<=3 |P|=100, repeat-rich text"}[workload],
        "value": value,
        "unit": "Gbases/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32",
        "data": "synthetic",
        "config": {"workload": f"{workload}: {desc}", "needles": n_pat, "needle_len": L, "k": kmax,
                   "text_bytes_per_gpu": hi - lo, "engine": engine_used,
                   "sharding": (f"text position, {world} shard(s), {window - 1}-symbol left context, hit records "
                                + ("exchanged with one fused all-gather per step (fixed-size [count | records] buffers of "
                                   f"{(cap + 1) * 16} bytes per rank)" if fused else
                                   "gathered to rank 0 per step (count all-gather + grouped send/recv: "
                                   "libspm_amd.dist.gatherv_hits)"))
                   if world > 1 else "single GPU"},
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "kernel": ("seed_filter_dense_kernel" if bs.dense else "seed_filter_kernel") if engine_used == "filter" else "myers_brute_kernel",
            "kernel_ms": k_ms,
            "launches_per_step": launches / max(args.steps, 1),
            "algorithmic_bytes_per_launch": hi - lo,
            "whole_step_GBps": (hi - lo) / (dt / args.steps) / 1e9,
            "whole_step_frac": (hi - lo) / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
            "stream_read_probe": HBM_STREAM_PROBE_GBS,
            "frac_of_stream_read_probe": achieved / HBM_STREAM_PROBE_GBS,
        },
        "hits": int(len(hits)),
        "needles_found": int(len(found)),
        "all_planted_found": bool(len(found) == n_pat),
        "patterns_create_ms": patterns_create_ms,
        "patterns_create": {"ms_in_the_library": bs.ms_total, "ms_tables": bs.ms_tables, "ms_index": bs.ms_index,
                            "ms_upload": bs.ms_upload, "device_bytes": int(bs.bytes_device), "host_threads": int(bs.threads), "filter_passes": int(bs.passes),
                            "dense_pass": bool(bs.dense), "keys": int(bs.keys),
                            "anchor_sixteenths": int(bs.anchor_sixteenths), "stride": int(bs.stride)},
        "first_scan_ms": first_scan_ms,
        "verify_ms_per_step": ms_verify / args.steps,
        "candidates": int(st.n_candidates),
        "bands_verified": int(st.n_bands),
        "fell_back": int(st.fell_back),
        "fallback_spans": int(st.fallback_spans),
        "lane_steps_per_s": n_pat * n_total / (dt / args.steps),
        # what the reference's traffic model (one full text pass PER needle) would have moved
        # in the same time -- for comparison only, NOT a roofline figure
        "reference_equivalent_traffic_GBps": n_pat * n_total / (dt / args.steps) / 1e9,
    }
    if repeats:
        result["repeat_text"] = {
            "fraction_requested": args.repeat_frac,
            "needles_across_a_stretch_on_purpose": (n_pat // args.repeat_needle_every) if args.repeat_needle_every else 0,
            "generator": "1024-base blocks; a block holds one stretch (16..256 bases) with probability frac*1024/136: "
                         "half tandem repeats (unit 1..6 bases, 1/64 impurities), half low-complexity (one base 7/8); "
                         "needles are cut at uniform positions of this text (crossing stretches at the natural rate) and "
                         "every --repeat-needle-every-th one across a stretch on purpose (random overlap)",
            "fallback_symbols": int(st.fallback_symbols),
        }
        # parity on a slice: the filter's hits == the brute-force engine's, on 64 MiB of this text
        nb = min(hi - lo, 64 << 20)
        hb = S.scan(ctx, text, ps, 0, nb, engine=S.ENGINE_BRUTE, max_hits=max_hits)
        hf = S.scan(ctx, text, ps, 0, nb, engine=engine, max_hits=max_hits)
        vb, vf = hb.view(), hf.view()
        result["parity_slice"] = {"bytes": nb, "hits": int(len(vf)),
                                  "equal_to_brute_force_engine": bool(np.array_equal(vb, vf)),
                                  "engine": {1: "brute", 2: "filter"}.get(int(hf.stats().engine_used), "?")}
        hb.close()
        hf.close()

    # brute-force engine (the one-lane-per-needle kernel) on a bounded slice, for reference next to the filter
    if world == 1 and args.engine != "brute" and args.brute_sample_mib > 0:
        nb = min(hi - lo, args.brute_sample_mib << 20)
        hb = S.scan(ctx, text, ps, 0, nb, engine=S.ENGINE_BRUTE, max_hits=max_hits)
        sb = hb.stats()
        hf = S.scan(ctx, text, ps, 0, nb, engine=engine, max_hits=max_hits)
        same = bool(np.array_equal(hb.view(), hf.view()))
        result["brute_force_engine"] = {
            "Gbases_per_s": nb / (sb.ms_main * 1e-3) / 1e9,
            "lane_steps_per_s": n_pat * nb / (sb.ms_main * 1e-3),
            "sample_bytes": nb,
            "hits_equal_to_default_engine": same,
            # this engine is bounded by integer-VALU issue, not HBM
            "valu": valu_roofline(workload, n_pat, n_pat * nb / (sb.ms_main * 1e-3)),
        }
        hb.close()
        hf.close()

    # the same scan over the optional 2-bit shadow of the text (spm_hip_text_pack): a quarter of the HBM traffic.
    # Reported next to `value`, never as `value`: the contract's algorithmic bytes are the 1-byte text.
    if world == 1 and args.packed_steps > 0 and engine_used == "filter":
        text.pack()
        for _ in range(2):
            S.scan(ctx, text, ps, ovl, ovl + (hi - lo), engine=engine, left_context=True, max_hits=max_hits).close()
        torch.cuda.synchronize()
        t0p = time.perf_counter()
        kms = 0.0
        for _ in range(args.packed_steps):
            hp = S.scan(ctx, text, ps, ovl, ovl + (hi - lo), engine=engine, left_context=True,
                        pos_offset=lo - ovl, max_hits=max_hits)
            stp = hp.stats()
            kms += stp.ms_main
            vp = hp.view() if _ == args.packed_steps - 1 else None
            hp.close()
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0p) / args.packed_steps
        result["packed_text_shadow"] = {
            "Gbases_per_s": n_total / dtp / 1e9,
            "ms_per_step": dtp * 1e3,
            "kernel_ms": kms / args.packed_steps,
            "hbm_bytes_streamed_per_launch": (hi - lo) // 4,
            "hits_equal_to_unpacked": bool(np.array_equal(np.sort(vp, order=["pattern", "pos"]),
                                                          np.sort(hits, order=["pattern", "pos"]))),
            "note": "optional 2-bit re-encoding of the resident text, built once per text (+25 % HBM); "
                    "excluded from `value` and from `roofline`",
        }
    ps.close()
    text.close()
    del hit_buf, gathered, last, g
    torch.cuda.empty_cache()
    return result


if __name__ == "__main__":
    main()
