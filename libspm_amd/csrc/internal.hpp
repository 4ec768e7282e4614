// internal.hpp -- what the translation units of libspm_hip.so share: the compiled needle set, the arguments of one scan, and
// the entry points of the engines.  (patterns.hip: needle sets; text.hip: context + haystacks; scan.hip: the scan driver;
// scan_brute.hip / scan_filter.hip: the engines, each with its kernels; hits.hip: results; jst.hip: journaled sequences;
// comm.hip: the multi-GPU exchange.)
#pragma once

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "common.hpp"
#include "index_build.hpp"

using namespace spm_hip;

// ----------------------------------------------------------------------------------------------------
// pattern set
// ----------------------------------------------------------------------------------------------------
struct spm_patterns : spm_hip::seed_index // (the seed index: passes, entries, seed layout -- index_build.hpp)
{
    spm_ctx *ctx = nullptr;
    int algo = 0;
    uint32_t n = 0;
    uint32_t sigma = 4;
    std::vector<uint8_t> ranks;
    std::vector<uint32_t> offsets;
    std::vector<int32_t> m, k; // padded to n_groups*64
    uint32_t n_groups = 0;
    uint32_t max_m = 0;
    uint32_t max_window = 0;
    uint32_t max_k = 0;
    uint32_t NW = 1;   // 32-bit words per needle in the brute kernels (power of two)
    bool is_myers() const { return algo == SPM_ALGO_MYERS || algo == SPM_ALGO_MYERS_PREFIX; }
    // device
    void *d_arena = nullptr;   // the set's one device allocation: every d_* table below points into it (patterns.hip)
    uint32_t *d_peq = nullptr; // [group][sigma+1][NW][64], needles top-aligned
    uint32_t *d_peq_bot = nullptr; // Myers only: same shape, needles bottom-aligned (cut-off kernel)
    uint32_t *d_peq_verify = nullptr; // exact matchers: Myers-style match masks (top-aligned) for the verify kernel;
                                      // for Myers sets verification reads d_peq itself
    uint32_t *d_hp0 = nullptr; // prefix: [group][NW][64]
    int32_t *d_m = nullptr;
    int32_t *d_k = nullptr;
    mutable uint64_t cand_hint = 0; // most candidates a filter scan of this set has produced so far
    uint8_t *d_surplus = nullptr;   // per needle: seeds - k (candidate merging); nullptr = no needle has k >= kMergeMinK
    uint8_t *d_ranks = nullptr;     // filterable sets: the needles' symbols, back to back (whole-seed check of a candidate)
    uint32_t *d_offsets = nullptr;  // ... and where each needle starts
    uint32_t *d_needle_pk = nullptr;  // dna4 sets: the needles 2 bits per symbol, 16 per word (piece count of a candidate)
    uint32_t *d_pk_offsets = nullptr; // ... and the first word of each
    uint16_t *d_seed_q = nullptr;
    pass_entry *d_pass_tab = nullptr; // the passes' key directories, for resolve_kernel
    uint4 *d_entries = nullptr;
    mutable uint64_t hit_hint = 0;    // most hits a filter scan of this set has reported so far (sizes the dedupe set)
    mutable bool scanned = false;     // the hints come from at least one completed filter scan
    mutable int exact_whole = -1;     // 1: k = 0 and every needle is its own single seed (decided at the first scan)
    mutable uint64_t band_hint = 0;   // ... and band-list slots it drew (sizes the verification grid)
    spm_build_stats build{};          // what spm_hip_patterns_create spent where
};

using clk = std::chrono::steady_clock;
inline float ms_since(clk::time_point t) { return std::chrono::duration<float, std::milli>(clk::now() - t).count(); }
// SPM_HIP_TRACE=1: one stderr line per C-ABI call that does work, with its timings
inline bool spm_trace_on()
{
    const char *v = getenv("SPM_HIP_TRACE");
    return v && *v && *v != '0';
}


// every knob of a scan, read from the environment in ONE place, once per C-ABI call that scans (defaults: the measured best)
struct scan_tuning
{
    int cand_cap = 0;
    int brute_cutoff = 1;
    int verify_wave_threads = 64;
    int verify_wave_lds_kb = 0;
    int verify_wave_nb = 1;
    int verify_wave_min_words = 8;
    int band_factor = 4;
    int band = 32;
    int exact_from_resolve = 1;
    int exact_skip_dedupe = 1;
    int dense_debug = 0;
    int s2_u = 2;
    int force_masked = 0;
    int filter_threads = 0;
    int spans_per_wave = 0;
    int span = 0;
    int span_budget = 0;
    int dyn = -1;
    int filter_u = 8;
    int nt = 1;
    int anchor_u = 4;
    int seed_check = 1;
    int flank_check = 1;
    int pieces_check = 1;
    int resolve_wgs_per_cu = 5;
    int resolve_surv_per_wg = 1024;
    int verify_runs = 1;
    int verify_runs_min_bands = 65536;
    int band_cap = 0;
    int resolve_debug = 0;
    static scan_tuning from_env()
    {
        scan_tuning T;
        T.cand_cap = env_int("SPM_HIP_FILTER_CAND_CAP", 0);
        T.brute_cutoff = env_int("SPM_HIP_BRUTE_CUTOFF", 1);
        T.verify_wave_threads = env_int("SPM_HIP_VERIFY_WAVE_THREADS", 64);
        T.verify_wave_lds_kb = env_int("SPM_HIP_VERIFY_WAVE_LDS_KB", 0);
        T.verify_wave_nb = env_int("SPM_HIP_VERIFY_WAVE_NB", 1);
        T.verify_wave_min_words = env_int("SPM_HIP_VERIFY_WAVE_MIN_WORDS", 8);
        T.band_factor = env_int("SPM_HIP_FILTER_BAND_FACTOR", 4);
        T.band = env_int("SPM_HIP_FILTER_BAND", 32);
        T.exact_from_resolve = env_int("SPM_HIP_EXACT_FROM_RESOLVE", 1);
        T.exact_skip_dedupe = env_int("SPM_HIP_EXACT_SKIP_DEDUPE", 1);
        T.dense_debug = env_int("SPM_HIP_DENSE_DEBUG", 0);
        T.s2_u = env_int("SPM_HIP_FILTER_S2_U", 2);
        T.force_masked = env_int("SPM_HIP_FILTER_FORCE_MASKED", 0);
        T.filter_threads = env_int("SPM_HIP_FILTER_THREADS", 0);
        T.spans_per_wave = env_int("SPM_HIP_FILTER_SPANS_PER_WAVE", 0);
        T.span = env_int("SPM_HIP_FILTER_SPAN", 0);
        T.span_budget = env_int("SPM_HIP_FILTER_SPAN_BUDGET", 0);
        T.dyn = env_int("SPM_HIP_FILTER_DYN", -1);
        T.filter_u = env_int("SPM_HIP_FILTER_U", 8);
        T.nt = env_int("SPM_HIP_FILTER_NT", 1);
        T.anchor_u = env_int("SPM_HIP_FILTER_ANCHOR_U", 4);
        T.seed_check = env_int("SPM_HIP_VERIFY_SEED_CHECK", 1);
        T.flank_check = env_int("SPM_HIP_FLANK_CHECK", 1);
        T.pieces_check = env_int("SPM_HIP_PIECES_CHECK", 1);
        T.resolve_wgs_per_cu = env_int("SPM_HIP_RESOLVE_WGS_PER_CU", 5);
        T.resolve_surv_per_wg = env_int("SPM_HIP_RESOLVE_SURV_PER_WG", 1024);
        T.verify_runs = env_int("SPM_HIP_VERIFY_RUNS", 1);
        T.verify_runs_min_bands = env_int("SPM_HIP_VERIFY_RUNS_MIN_BANDS", 65536);
        T.band_cap = env_int("SPM_HIP_FILTER_BAND_CAP", 0);
        T.resolve_debug = env_int("SPM_HIP_RESOLVE_DEBUG", 0);
        return T;
    }
};

struct scan_args
{
    spm_ctx *ctx;
    const spm_text *text;
    uint64_t begin, end, ctx_begin;
    const spm_patterns *ps;
    spm_scan_opts opts;
    const void *state_in;
    void *state_out;
    spm_hits *hits;
    scan_tuning tune;                        // the environment's knobs, as this call found them
    const uint64_t *seg_offsets = nullptr; // host; n_segments + 1 entries
    uint64_t n_segments = 0;
    const uint64_t *d_seg_offsets = nullptr; // the same table already resident on the device (journaled-sequence index)
    const uint32_t *d_seg_owned = nullptr;   // optional per-segment offset of the first wanted end symbol (filter engine)
    uint64_t cand_cap_override = 0;          // retry after a survivor overflow: the count the first attempt needed
    uint64_t band_scale = 0;                 // retry after a band list / table overflow: that much more room
    bool seen_full = false;                  // retry after a dedupe-set overflow: size it for the caller's hit buffer
    bool need_seen = false;                  // exact sets reporting from the resolve kernel: the dedupe set after all (spans gave up)
    bool seen_skipped = false;               // ... this run went without it
    bool exact_used = false;                 // this run reported its hits from the resolve kernel (no bands, no verification)
    std::vector<uint64_t> seg_host;          // host copy fetched on demand when only the device table was given
    // span-local fallback: the filter run leaves these for the brute-force re-scan of the spans that gave up
    unsigned long long *d_seen = nullptr;
    uint32_t seen_mask = 0;
    uint64_t *d_ovf = nullptr;               // overflow list in the scratch buffer: {begin, symbols} per span
    const std::vector<uint64_t> *tiles = nullptr; // brute pass over an explicit tile table {scan_lo, own_lo, own_hi}
};

constexpr uint64_t kOvfCap = 1ull << 17; // spans the overflow list holds (2 MiB); beyond: whole-scan fallback


// ---- state blobs: ABI state <-> the kernels' [group][rows][64] layout (patterns.hip) ----
void state_to_internal(const spm_patterns *p, const void *state, std::vector<uint32_t> &out);
void state_from_internal(const spm_patterns *p, const std::vector<uint32_t> &in, void *state);
// ---- haystack allocation (text.hip) ----
int text_alloc(spm_ctx *ctx, uint64_t n, uint32_t sigma, spm_text **out);
// ---- the engines ----
int ensure_scratch(spm_ctx *ctx, size_t bytes);
// one brute-force pass; `report` = false suppresses hits (state-only pass)   (scan_brute.hip)
int run_brute(const scan_args &A, uint64_t begin, uint64_t end, uint64_t ctx_begin, const uint32_t *d_state_in,
              uint32_t *d_state_out, bool report, bool single_tile);
// the seed filter's launches for one scan: streaming pass(es), resolve, verification   (scan_filter.hip)
int run_filter(const scan_args &A);
// after_launch: work the caller wants on the stream right behind the filter scan's kernels, before the host reads the
// counters (the pan-genome search's fan-out); spm_hits::hook_final tells whether it saw the final hit list.   (scan.hip)
int scan_impl(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end, const spm_patterns *patterns,
              const spm_scan_opts *opts_in, const void *state_in, void *state_out, const uint64_t *seg_offsets,
              uint64_t n_segments, spm_hits **out, const uint64_t *d_seg_offsets = nullptr,
              const uint32_t *d_seg_owned = nullptr, const std::function<int(spm_hits *)> *after_launch = nullptr);

// Every translation unit with kernels is a code object of its own, loaded by the HIP runtime at the first launch out of
// it (~1-3 ms each).  spm_hip_init loads them all, so that the first scan of a process does not pay for it.
void spm_warm_text_kernels();
void spm_warm_brute_kernels();
void spm_warm_filter_kernels();
void spm_warm_hits_kernels();
void spm_warm_jst_kernels();

// deferred scans (SPM_SCAN_DEFER): read the counters back, and repeat the scan if it needs attention   (scan.hip)
int spm_complete_deferred(spm_hits *h);
