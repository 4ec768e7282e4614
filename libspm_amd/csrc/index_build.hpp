// index_build.hpp -- host side of the seed filter: which windows of which needles are indexed, and the tables the
// kernels of filter.hpp read.  PURE HOST C++17 (no HIP): spm_hip.hip includes it for the product, index_host.cpp compiles
// it with plain g++ for the sanitizer build and the host self-check.
//
// What the reference does at this point is O(|P|) per needle: its matcher constructors build one SeqAn pattern each
// (/root/reference/libspm/libspm/matcher/myers_matcher.hpp:40-43, shiftor_matcher.hpp:38-40).  A set of 100 000 needles
// is 400 000+ seeds here, so the build is threaded (index_tuning::threads).
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/spm_hip.h"
#include "filter_shared.hpp"

namespace spm_hip
{

inline int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

inline uint32_t next_pow2(uint32_t x)
{
    uint32_t p = 1;
    while (p < x)
        p <<= 1;
    return p;
}

// every knob of the index build, read from the environment in ONE place (spm_hip_patterns_create / the self-check)
struct index_tuning
{
    int force_keylen = 0, force_stride = 0;
    int max_keys = 57344; // keys per LDS fingerprint table
    int max_passes = 256;
    int hash = 2, probes = 4, bitmap_words = 0;
    int anchor = 1;
    int dedupe = 1, merge_run = 12;
    int dense = 1;          // 0: never, 1: when the sparse plan needs several passes or stride 1, 2: whenever the set admits it
    int dense_min_density = 0; // force at least this many sixteenths of the dimers as anchors (diagnostics)
    int dense_max_density = 8; // a set that needs more than this many sixteenths keeps its sparse passes (dense = 2: no limit)
    int dense_cmax = 4;     // pieces of a needle may overlap up to this many deep (c k + 1 pieces then)
    int dense_sweeps = 2;   // rounds of key re-selection that make needles share presence bits (0: first-fit keys)
    int sparse_bits = 1;    // sparse passes of stride 1 / 2 over dna4: presence bits + L2 buckets as level 1 (0: fingerprint table)
    int threads = 0;        // 0: hardware concurrency, at most 16
    static index_tuning from_env()
    {
        index_tuning T;
        T.force_keylen = env_int("SPM_HIP_FILTER_KEYLEN", 0);
        T.force_stride = env_int("SPM_HIP_FILTER_STRIDE", 0);
        T.max_keys = std::max(1024, env_int("SPM_HIP_FILTER_MAX_KEYS", 57344));
        T.max_passes = std::max(1, env_int("SPM_HIP_FILTER_MAX_PASSES", 256));
        T.hash = std::max(0, std::min(2, env_int("SPM_HIP_FILTER_HASH", 2)));
        T.probes = std::max(1, std::min(4, env_int("SPM_HIP_FILTER_PROBES", 4)));
        T.bitmap_words = env_int("SPM_HIP_FILTER_BITMAP_WORDS", 0);
        T.anchor = env_int("SPM_HIP_FILTER_ANCHOR", 1);
        T.dedupe = env_int("SPM_HIP_FILTER_DEDUPE", 1);
        T.merge_run = std::max(1, env_int("SPM_HIP_FILTER_MERGE_RUN", 12));
        T.dense = env_int("SPM_HIP_FILTER_DENSE", 1);
        T.dense_min_density = env_int("SPM_HIP_FILTER_DENSE_MIN_DENSITY", 0);
        T.dense_max_density = std::max(2, std::min(16, env_int("SPM_HIP_FILTER_DENSE_MAX_DENSITY", 8)));
        T.dense_cmax = std::max(1, std::min(8, env_int("SPM_HIP_FILTER_DENSE_CMAX", 4)));
        T.dense_sweeps = std::max(0, std::min(8, env_int("SPM_HIP_FILTER_DENSE_SWEEPS", 2)));
        T.sparse_bits = env_int("SPM_HIP_FILTER_BITS", 1);
        T.threads = env_int("SPM_HIP_BUILD_THREADS", 0);
        return T;
    }
    unsigned n_threads() const
    {
        if (threads > 0)
            return (unsigned)std::min(threads, 64);
        const unsigned hc = std::thread::hardware_concurrency();
        return std::max(1u, std::min(16u, hc ? hc : 1u));
    }
};

// A team of worker threads that lives as long as one index build: run(n, fn) calls fn(begin, end, thread) over [0, n) in
// contiguous slices, one per thread (the caller takes slice 0).  The rounds of dense_share_bits start 64 of these; spawning
// threads for each would cost more than the work.
class thread_team
{
  public:
    explicit thread_team(unsigned n) : n_(std::max(1u, n))
    {
        for (unsigned t = 1; t < n_; ++t)
            workers_.emplace_back([this, t]() { loop(t); });
    }
    ~thread_team()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
            ++epoch_;
        }
        cv_.notify_all();
        for (std::thread &w : workers_)
            w.join();
    }
    unsigned size() const { return n_; }
    template <typename F>
    void run(size_t n, F fn)
    {
        if (n_ <= 1 || n < 2) {
            fn((size_t)0, n, 0u);
            return;
        }
        std::function<void(unsigned)> job = [&](unsigned t) { fn(n * t / n_, n * (t + 1) / n_, t); };
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &job;
            pending_ = n_ - 1;
            ++epoch_;
        }
        cv_.notify_all();
        job(0);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&]() { return pending_ == 0; });
        job_ = nullptr;
    }

  private:
    void loop(unsigned t)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(unsigned)> *job = nullptr;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&]() { return epoch_ != seen; });
                seen = epoch_;
                if (stop_)
                    return;
                job = job_;
            }
            (*job)(t);
            {
                std::lock_guard<std::mutex> g(m_);
                if (--pending_ == 0)
                    done_.notify_one();
            }
        }
    }
    unsigned n_;
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(unsigned)> *job_ = nullptr;
    unsigned pending_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

// fn(begin, end, thread) over [0, n) in contiguous slices
template <typename F>
inline void parallel_slices(size_t n, unsigned n_threads, F fn)
{
    if (n_threads <= 1 || n < 2) {
        fn((size_t)0, n, 0u);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(n_threads);
    for (unsigned t = 0; t < n_threads; ++t) {
        const size_t b = n * t / n_threads, e = n * (t + 1) / n_threads;
        th.emplace_back([=]() { fn(b, e, t); });
    }
    for (std::thread &x : th)
        x.join();
}

// the needle set as the index build sees it (views; owned by the caller)
struct needle_view
{
    int algo = 0;
    uint32_t n = 0, sigma = 4;
    const uint8_t *ranks = nullptr;
    const uint32_t *offsets = nullptr; // n + 1
    const int32_t *m = nullptr, *k = nullptr;
    uint32_t max_k = 0;
    bool is_myers() const { return algo == SPM_ALGO_MYERS || algo == SPM_ALGO_MYERS_PREFIX; }
};

// one pass over the text: the level-1 image every workgroup stages into LDS, the exact key directory, and (dense passes)
// the bucketed fingerprint table in between
struct filter_index
{
    // anchored passes: every key begins with a dimer d (sym0 | sym1 << 2) with (d ^ anchor_c) & anchor_cm == 0; cm = 0: unanchored
    uint32_t anchor_c = 0, anchor_cm = 0;
    bool ok = false;
    uint32_t dense = 0;      // 1: presence bits in LDS + fingerprint buckets in L2, anchors = union of n_pat patterns
    uint32_t n_pat = 0, pat_c[kDensePatterns] = {0, 0, 0}, pat_cm[kDensePatterns] = {0, 0, 0};
    uint32_t dimer_set = 0xFFFF; // bit d set: windows beginning with dimer d are looked up
    uint32_t bucket_shift = 0;   // bucket = (key * C) >> bucket_shift
    uint32_t stride = 0;
    uint32_t key_len = 16;
    uint32_t bitmap_words = 0;
    uint32_t n_probes = 0;
    uint32_t hash_variant = 0;
    uint32_t lds_words = 0;
    uint32_t chd_slot_mask = 0, chd_bucket_shift = 0, chd_disp_off = 0;
    uint32_t ht_mask = 0;
    uint64_t n_keys = 0;
    uint64_t n_entries = 0; // entries of the exact table after identical (key, needle) pairs were merged
    uint32_t max_range = 0; // largest diagonal range of a merged entry
    // host images (dropped after the upload unless the caller keeps them: the self-check does)
    std::vector<uint32_t> h_image;
    std::vector<u32x4> h_ht; // the directory: {key, first entry, entries, -}, open addressing, an empty slot has .z == 0
    std::vector<uint16_t> h_buckets; // dense: kDenseSlots 16-bit slots per bucket
    // device copies (spm_hip.hip)
    uint32_t *d_bitmap = nullptr;
    u32x4 *d_ht = nullptr;
    u32x4 *d_buckets = nullptr;
};

struct seed_index
{
    std::vector<filter_index> fidx; // one per pass; empty = the seed filter does not apply
    uint32_t filter_stride = 0;
    uint32_t filter_key_len = 16;
    bool filter_anchored = false; // stride 1, one key per seed, chosen to begin with an anchor dimer of its pass
    bool filter_dense = false;    // one dense pass (fidx.size() == 1)
    uint32_t filter_max_range = 0; // largest diagonal range over all passes
    // seed layout: needle p has seed_n[p] seeds at seed_off[seed_first[p] + j]; sparse passes: all seed_q[p] symbols long;
    // dense pass: seed_len[seed_first[p] + j] symbols, and they may overlap seed_c[p] deep (seed_n[p] >= seed_c[p] k + 1)
    std::vector<uint16_t> seed_q, seed_n, seed_off, seed_len;
    std::vector<uint8_t> seed_c;
    std::vector<uint32_t> seed_first;
    std::vector<u32x4> h_entries; // exact entries of all passes, grouped by key: {val = needle << 11 | offset, seed
                                  // signature, range code, key}
};

// identical (key, needle) entries beyond this many are merged into one with a diagonal range.  (Measured on the 1 % repeat
// text, 16 / 128 needles across a stretch: > 4: 6.4 / 12.2 ms, > 8: 5.0 / 9.2, > 12: 5.0 / 8.4, > 24: 5.1 / 8.7, never: 5.1 /
// 9.4 -- merged entries skip the per-offset checks and cost bands, single ones cost checks.)
constexpr size_t kMergeRun = 12;

struct seed_key // one indexed window: needle p, seed at offset o of the needle, window starting r symbols into the seed
{
    uint32_t p, o, r;
    uint32_t q; // length of the seed
};

// A symbol the 2-bit keys can hold: A, C, G, T.  dna5 (seqan3 ranks A0 C1 G2 N3 T4): everything but N; dna15 (A0 B1 C2 D3
// G4 H5 K6 M7 N8 R9 S10 T11 V12 W13 Y14): A, C, G, T only.
inline bool key_symbol(uint32_t sigma, uint8_t c)
{
    return sigma == 4 ? c < 4 : sigma == 5 ? (c < 5 && c != 3) : (c == 0 || c == 2 || c == 4 || c == 11);
}
inline uint32_t key_code(uint32_t sigma, uint8_t c) // 2-bit code of a key symbol
{
    return sigma == 4 ? (c & 3u) : sigma == 5 ? (c == 4 ? 3u : c) : (c == 11 ? 3u : (uint32_t)c >> 1);
}

// Seeds of one needle.  The pigeonhole argument needs n DISJOINT pieces of the needle (n = k + 1, or k + 2 for needles
// with many errors: two intact pieces on nearby diagonals) -- they need not tile it.  A needle of key symbols only is cut
// into n pieces of q = floor(m / n) at offsets j * q.  A needle with an N (or, in dna15, any other ambiguity code) takes
// its pieces from its stretches of key symbols -- the n first pieces of the largest length q that yields n of them --,
// because a piece with an N can only occur where the text has an N too, and the filter never looks there: an intact
// piece WITHOUT one is found like any other seed.  false: the needle has no such layout with q >= q_floor.
inline bool layout_seeds(const needle_view &nv, uint32_t p, uint32_t q_floor, uint32_t &n_out, uint32_t &q_out,
                         std::vector<uint16_t> &off)
{
    const uint32_t m = (uint32_t)nv.m[p], k = nv.is_myers() ? (uint32_t)nv.k[p] : 0;
    const uint8_t *pat = nv.ranks + nv.offsets[p];
    bool clean = true;
    for (uint32_t i = 0; i < m; ++i)
        clean = clean && key_symbol(nv.sigma, pat[i]);
    const seed_plan sp = plan_seeds(m, k);
    off.clear();
    if (clean) {
        n_out = sp.n;
        q_out = sp.q;
        for (uint32_t j = 0; j < sp.n; ++j)
            off.push_back((uint16_t)(j * sp.q));
        return sp.q >= q_floor;
    }
    std::vector<std::pair<uint32_t, uint32_t>> runs; // (begin, length) of the stretches of key symbols
    for (uint32_t i = 0; i < m;) {
        if (!key_symbol(nv.sigma, pat[i])) {
            ++i;
            continue;
        }
        uint32_t j = i;
        while (j < m && key_symbol(nv.sigma, pat[j]))
            ++j;
        runs.emplace_back(i, j - i);
        i = j;
    }
    for (uint32_t n : {sp.n, k + 1}) { // (a needle that cannot afford the surplus seed keeps k + 1)
        for (uint32_t q = m / n; q >= q_floor && q > 0; --q) {
            uint64_t have = 0;
            for (const auto &r : runs)
                have += r.second / q;
            if (have < n)
                continue;
            for (const auto &r : runs)
                for (uint32_t j = 0; j + q <= r.second && off.size() < n; j += q)
                    off.push_back((uint16_t)(r.first + j));
            n_out = n;
            q_out = q;
            return true;
        }
        if (sp.n == k + 1)
            break;
    }
    return false;
}

struct index_kv // one indexed window on its way into the tables
{
    uint32_t key, val, sig, meta;
};

// key of the window seed[r, r + H) and the entry fields that go with it (filter.hpp: seed_sig_ok, kRngSingle)
inline index_kv make_kv(const needle_view &nv, const seed_key &it, uint32_t H)
{
    const uint32_t p = it.p, o = it.o, r = it.r, q = it.q;
    const uint8_t *pat = nv.ranks + nv.offsets[p];
    uint32_t key = 0;
    for (uint32_t i = 0; i < H; ++i)
        key |= key_code(nv.sigma, pat[o + r + i]) << (2 * i);
    // signature: the REST of the seed -- its r symbols before the key window, then those after it --, the first 16 of
    // them, 2 bits each.  With the key that is the whole seed when q <= key_len + 16, so the resolve kernel checks "the
    // seed occurs here unchanged" in registers
    uint32_t sig = 0, ns = 0;
    for (uint32_t i = r > 16 ? r - 16 : 0; i < r && ns < 16; ++i, ++ns) // (the 16 symbols next to the window)
        sig |= key_code(nv.sigma, pat[o + i]) << (2 * ns);
    for (uint32_t i = r + H; i < q && ns < 16; ++i, ++ns)
        sig |= key_code(nv.sigma, pat[o + i]) << (2 * ns);
    // range code of a single entry: where the window sits in its seed (r), how many rest symbols the signature holds
    // (ns), and whether that is the whole rest
    const uint32_t meta = kRngSingle | (r & 0x1F) | (ns << 5) | (ns == q - H ? kRngWhole : 0u);
    return {key, (p << 11) | (o + r), sig, meta};
}

// Exact level: a directory key -> (first entry, count) with open addressing, and the entries of a key side by side in
// one array (all passes share it).  A survivor costs one short directory probe; its entries -- a key that twenty needles
// share has twenty -- are then dealt to the lanes of the wave one pair each (resolve_kernel), instead of one lane walking a
// probe sequence while 63 wait.  `keys` comes back sorted by key.
inline void build_directory(std::vector<index_kv> &keys, std::vector<uint16_t> &ranges, filter_index &F,
                            std::vector<u32x4> &entries)
{
    {
        // stable LSD radix sort by key (3 passes of 11 bits) of (key, index) pairs -- the passes read and write 8-byte pairs
        // in sequence; sorting an index array THROUGH the 16-byte records was six passes of cache misses --, then one gather
        const size_t n = keys.size();
        std::vector<uint64_t> pa(n), pb(n);
        for (size_t i = 0; i < n; ++i)
            pa[i] = ((uint64_t)keys[i].key << 32) | (uint64_t)i;
        for (uint32_t shift = 0; shift < 32; shift += 11) {
            uint32_t count[2049] = {0};
            for (size_t i = 0; i < n; ++i)
                ++count[((pa[i] >> (32 + shift)) & 2047u) + 1];
            for (uint32_t b = 0; b < 2048; ++b)
                count[b + 1] += count[b];
            for (size_t i = 0; i < n; ++i)
                pb[count[(pa[i] >> (32 + shift)) & 2047u]++] = pa[i];
            pa.swap(pb);
        }
        std::vector<uint32_t> order(n);
        for (size_t i = 0; i < n; ++i)
            order[i] = (uint32_t)pa[i];
        std::vector<index_kv> k2(n);
        std::vector<uint16_t> r2(n);
        for (size_t i = 0; i < n; ++i) {
            k2[i] = keys[order[i]];
            r2[i] = ranges[order[i]];
        }
        keys.swap(k2);
        ranges.swap(r2);
    }
    size_t n_distinct = 0;
    for (size_t i = 0; i < keys.size(); ++i)
        n_distinct += (i == 0 || keys[i].key != keys[i - 1].key) ? 1 : 0;
    const uint32_t ht_size = next_pow2((uint32_t)std::max<uint64_t>(1024, n_distinct * 2));
    F.ht_mask = ht_size - 1;
    F.h_ht.assign(ht_size, u32x4{0, 0, 0, 0});
    entries.reserve(entries.size() + keys.size());
    for (size_t i = 0; i < keys.size();) {
        size_t j = i;
        while (j < keys.size() && keys[j].key == keys[i].key)
            ++j;
        // (the directory of 400 000 keys is 16 MB of 16-byte slots hit at random: ask for the slots of the keys a few steps
        // ahead while this one is placed -- the inserts were 20 ms of a 100 ms build as a chain of cache misses)
        if (j + 12 < keys.size())
            __builtin_prefetch(&F.h_ht[ht_hash(keys[j + 12].key) & F.ht_mask], 1, 0);
        uint32_t slot = ht_hash(keys[i].key) & F.ht_mask;
        while (F.h_ht[slot].z != 0)
            slot = (slot + 1) & F.ht_mask;
        F.h_ht[slot] = u32x4{keys[i].key, (uint32_t)entries.size(), (uint32_t)(j - i), 0};
        for (size_t q = i; q < j; ++q)
            entries.push_back(u32x4{keys[q].val, keys[q].sig, ranges[q], keys[q].key});
        i = j;
    }
}

// level 1 of a dense pass -- and of sparse passes that look at 8 or 16 windows per 16 symbols, where two LDS reads and two
// multiplies per window (the fingerprint table) cost more than the pass can hide: one presence bit per key in LDS, then a
// bucketed fingerprint table in L2 for the windows whose bit is set (filter_shared.hpp)
inline void build_bits_level1(const std::vector<index_kv> &keys, filter_index &F)
{
    F.h_image.assign((1u << kDenseBloomBits) / 32, 0);
    for (const index_kv &e : keys) {
        const uint32_t b = dense_bloom_index(e.key);
        F.h_image[b >> 5] |= 1u << (b & 31);
    }
    F.bitmap_words = F.lds_words = (uint32_t)F.h_image.size();
    uint32_t lg = 12; // about two keys per bucket of kDenseSlots
    while ((1ull << lg) * 2 < keys.size() && lg < 24)
        ++lg;
    F.bucket_shift = 32 - lg;
    F.h_buckets.assign((size_t)kDenseSlots << lg, 0);
    for (const index_kv &e : keys) {
        uint16_t *bk = F.h_buckets.data() + (size_t)dense_bucket(e.key, F.bucket_shift) * kDenseSlots;
        const uint16_t fp = (uint16_t)dense_fp(e.key);
        uint32_t s = 0;
        while (s < kDenseSlots && bk[s] != 0 && bk[s] != fp)
            ++s;
        if (s < kDenseSlots)
            bk[s] = fp;
        else
            bk[kDenseSlots - 1] = (uint16_t)kDenseAcceptAll; // overflow: this bucket lets every window through
    }
}

inline int build_one_index(const needle_view &nv, const index_tuning &T, const std::vector<seed_key> &items, uint32_t S,
                           filter_index &F, std::vector<u32x4> &entries)
{
    F.ok = false;
    std::vector<index_kv> keys;
    keys.reserve(items.size());
    for (const seed_key &it : items)
        keys.push_back(make_kv(nv, it, F.key_len)); // window seed[r, r+H) -- inside the seed because r <= q - H
    F.n_keys = keys.size();
    if (F.n_keys == 0)
        return SPM_OK;
    // A periodic seed puts the same key at several offsets of one needle (a homopolymer run: at every shift of every
    // seed).  More than kMergeRun such entries -- the needle IS a repeat there -- are merged into one with a diagonal
    // range: a text window then yields ONE pair per needle, counted into the bands of all the offsets, without per-offset
    // checks (they would pass wherever the text carries the same repeat).  Shorter runs -- a needle that merely ends in a
    // repeat -- stay apart, each with its own seed signature.  (Not for sets whose bands count seed hits: the count
    // works on single diagonals.)
    std::vector<uint16_t> ranges(keys.size(), 0);
    for (size_t i = 0; i < keys.size(); ++i)
        ranges[i] = (uint16_t)keys[i].meta;
    const bool band_merging = nv.max_k >= kMergeMinK && nv.max_k <= 1000;
    if (!band_merging && T.dedupe != 0) {
        std::sort(keys.begin(), keys.end(),
                  [](const index_kv &a, const index_kv &b) { return a.key != b.key ? a.key < b.key : a.val < b.val; });
        size_t w = 0;
        for (size_t i = 0; i < keys.size();) {
            size_t j = i + 1;
            while (j < keys.size() && keys[j].key == keys[i].key && (keys[j].val >> 11) == (keys[i].val >> 11))
                ++j;
            if (j - i > (size_t)T.merge_run) {
                const uint32_t span = (keys[j - 1].val & 0x7FF) - (keys[i].val & 0x7FF);
                keys[w] = keys[i];
                ranges[w] = (uint16_t)(kRngRun | span);
                F.max_range = std::max<uint32_t>(F.max_range, span);
                ++w;
            } else {
                for (size_t q = i; q < j; ++q) {
                    keys[w] = keys[q];
                    ranges[w] = (uint16_t)keys[q].meta;
                    ++w;
                }
            }
            i = j;
        }
        keys.resize(w);
        ranges.resize(w);
    }
    F.n_entries = keys.size();
    F.stride = S;
    F.n_probes = (uint32_t)T.probes;
    F.hash_variant = (uint32_t)T.hash;
    std::vector<uint32_t> &image = F.h_image; // what every workgroup stages into LDS
    image.clear();
    if (T.sparse_bits != 0 && T.hash == 2 && nv.sigma == 4 && S <= 2 && F.anchor_cm == 0) {
        // 8 or 16 windows of every 16 symbols are looked up: presence bits (one LDS read, no multiply) + L2 buckets
        F.hash_variant = 4;
        build_bits_level1(keys, F);
        build_directory(keys, ranges, F, entries);
        F.ok = true;
        return SPM_OK;
    }
    if (F.hash_variant == 2) {
        // ---- perfect-hash fingerprint table (hash-and-displace, see filter_shared.hpp) ----
        std::vector<uint32_t> uniq;
        uniq.reserve(keys.size());
        for (const index_kv &e : keys)
            uniq.push_back(e.key);
        std::sort(uniq.begin(), uniq.end());
        uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
        uint32_t n_slots = 1024;
        while (n_slots < 2 * uniq.size() && n_slots < 65536)
            n_slots <<= 1;
        bool ok = uniq.size() <= (size_t)(0.96 * n_slots);
        const uint32_t n_buckets = std::max(64u, n_slots / 8);
        uint32_t lg = 0;
        while ((1u << lg) < n_buckets)
            ++lg;
        const uint32_t shift = 32 - lg;
        std::vector<uint16_t> fp(n_slots, 0xFFFF), disp(n_buckets, 0);
        if (ok) {
            // bucket the keys (counting sort: no per-bucket vectors)
            std::vector<uint32_t> b_begin(n_buckets + 1, 0), b_keys(uniq.size());
            for (uint32_t k : uniq)
                ++b_begin[(chd_hash(k).x >> shift) + 1];
            for (uint32_t b = 0; b < n_buckets; ++b)
                b_begin[b + 1] += b_begin[b];
            {
                std::vector<uint32_t> fill(b_begin.begin(), b_begin.end() - 1);
                for (uint32_t k : uniq)
                    b_keys[fill[chd_hash(k).x >> shift]++] = k;
            }
            std::vector<uint32_t> order(n_buckets);
            for (uint32_t b = 0; b < n_buckets; ++b)
                order[b] = b;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
                return b_begin[a + 1] - b_begin[a] > b_begin[b + 1] - b_begin[b];
            });
            std::vector<uint8_t> used(n_slots, 0);
            uint32_t slots[64];
            for (uint32_t b : order) {
                const uint32_t nb = b_begin[b + 1] - b_begin[b];
                if (nb == 0)
                    break;
                if (nb > 64) {
                    ok = false;
                    break;
                }
                const uint32_t *B = b_keys.data() + b_begin[b];
                chd_hashes hh[64];
                for (uint32_t i = 0; i < nb; ++i)
                    hh[i] = chd_hash(B[i]);
                bool placed = false;
                for (uint32_t d = 0; d < 65536 && !placed; ++d) {
                    bool good = true;
                    for (uint32_t i = 0; i < nb && good; ++i) {
                        const uint32_t sl = chd_slot(hh[i], d, n_slots - 1);
                        if (used[sl])
                            good = false;
                        for (uint32_t j = 0; j < i && good; ++j)
                            if (slots[j] == sl)
                                good = false;
                        slots[i] = sl;
                    }
                    if (good) {
                        for (uint32_t i = 0; i < nb; ++i) {
                            used[slots[i]] = 1;
                            fp[slots[i]] = (uint16_t)hh[i].f;
                        }
                        disp[b] = (uint16_t)d;
                        placed = true;
                    }
                }
                if (!placed) {
                    ok = false;
                    break;
                }
            }
        }
        if (ok) {
            F.chd_slot_mask = n_slots - 1;
            F.chd_bucket_shift = shift;
            F.chd_disp_off = n_slots * 2;
            image.resize((n_slots * 2 + n_buckets * 2) / 4);
            memcpy(image.data(), fp.data(), n_slots * 2);
            memcpy((uint8_t *)image.data() + n_slots * 2, disp.data(), n_buckets * 2);
            F.bitmap_words = (uint32_t)image.size();
        } else {
            F.hash_variant = 1; // key set too dense for the fingerprint table: Bloom cascade
        }
    }
    if (nv.sigma != 4 && F.hash_variant != 2)
        return SPM_OK; // the dna5 kernel is built for the fingerprint table only
    if (F.hash_variant != 2) {
        uint64_t want_bits = F.n_keys * 32;
        uint32_t words = 1024;
        while ((uint64_t)words * 32 < want_bits && words < 32768)
            words <<= 1;
        const int force_w = T.bitmap_words;
        if (force_w >= 256 && force_w <= 32768 && (force_w & (force_w - 1)) == 0)
            words = (uint32_t)force_w;
        F.bitmap_words = words;
        image.assign(words, 0);
        const uint32_t idx_mask = words * 32 - 1;
        for (const index_kv &e : keys)
            for (uint32_t pr = 0; pr < F.n_probes; ++pr) {
                const uint32_t hh = (F.hash_variant ? bloom_hash<1>(e.key, pr) : bloom_hash<0>(e.key, pr)) & idx_mask;
                image[hh >> 5] |= 1u << (hh & 31);
            }
    }
    F.lds_words = (uint32_t)image.size();
    build_directory(keys, ranges, F, entries);
    F.ok = true;
    return SPM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense pass: ONE pass for a needle set of any size (filter_shared.hpp says how the kernel looks windows up).
//
// Pieces.  The pigeonhole argument in its general form: if no needle POSITION lies in more than c pieces, k edits destroy
// at most c k pieces (an edit touches one position, or the gap between two, and a piece holding a gap holds its left
// position), so c k + 1 pieces leave one intact.  c = 1 is the usual "k + 1 disjoint pieces".  Every piece is a 16-symbol
// key window that begins with an anchor dimer, plus (c = 1 only) up to 16 neighbouring symbols that no other piece claims
// -- they go into the entry's signature, so a chance match of the key dies in registers.  Keys are taken greedily from
// the left: position p is taken if it is anchored and p >= (the c-th last one taken) + 16.  A needle that cannot fill
// k + 1 pieces tries c = 2, 3, .. dense_cmax; if one needle still fails, the anchor set grows.
//
// Anchors.  A union of <= kDensePatterns dimer patterns.  Densities are tried in ascending order (1/8, 3/16, 1/4, 5/16,
// 3/8, 1/2, 3/4, 1); at every density the candidate sets are ranked on a sample of the needles by how many of them they
// leave without a layout, and the best one is tried on all.  (Random 150-symbol needles, k = 3: 1/8 leaves 0.2 % of them
// to c >= 2 and none uncovered; 100-symbol ones need 3/16.)
// ---------------------------------------------------------------------------------------------------------------------
struct dense_anchor_set
{
    uint32_t n_pat = 0, c[kDensePatterns] = {0, 0, 0}, cm[kDensePatterns] = {0, 0, 0};
    uint32_t dimers = 0; // bit d set: dimer d is an anchor
    uint32_t sixteenths() const
    {
        uint32_t n = 0;
        for (uint32_t d = 0; d < 16; ++d)
            n += (dimers >> d) & 1u;
        return n;
    }
};

inline uint32_t dimer_mask_of(uint32_t c, uint32_t cm)
{
    uint32_t set = 0;
    for (uint32_t d = 0; d < 16; ++d)
        set |= (((d ^ c) & cm) == 0 ? 1u : 0u) << d;
    return set;
}

inline dense_anchor_set anchor_union(const dense_anchor_set &a, uint32_t c, uint32_t cm)
{
    dense_anchor_set r = a;
    r.c[r.n_pat] = c & cm;
    r.cm[r.n_pat] = cm;
    ++r.n_pat;
    r.dimers |= dimer_mask_of(c, cm);
    return r;
}

// greedy key positions of one needle for overlap depth c; returns how many were found (at most `want`)
inline uint32_t dense_pick(const uint8_t *pat, uint32_t m, uint32_t dimers, uint32_t c, uint32_t want, uint16_t *pos)
{
    if (m < kKeyMax)
        return 0;
    uint32_t n = 0;
    for (uint32_t i = 0; i + kKeyMax <= m && n < want; ++i) {
        const uint32_t d = (pat[i] & 3u) | ((uint32_t)(pat[i + 1] & 3u) << 2);
        if (!((dimers >> d) & 1u))
            continue;
        if (n >= c && i < (uint32_t)pos[n - c] + kKeyMax)
            continue;
        pos[n++] = (uint16_t)i;
    }
    return n;
}

// smallest overlap depth with which the needle gets its c k + 1 pieces (0: none up to cmax)
inline uint32_t dense_layout(const uint8_t *pat, uint32_t m, uint32_t k, uint32_t dimers, uint32_t cmax, uint16_t *pos,
                             uint32_t &n_out)
{
    for (uint32_t c = 1; c <= cmax; ++c) {
        const uint32_t want = c * k + 1;
        if (want > 255)
            break;
        if (dense_pick(pat, m, dimers, c, want, pos) == want) {
            n_out = want;
            return c;
        }
    }
    return 0;
}

constexpr uint32_t kDenseMaxPieces = 8 * 7 + 1; // cmax <= 8, k <= 7

inline bool dense_eligible(const needle_view &nv)
{
    if (nv.sigma != 4 || nv.max_k >= kMergeMinK || nv.n == 0)
        return false;
    for (uint32_t p = 0; p < nv.n; ++p) {
        const uint32_t m = (uint32_t)nv.m[p], k = nv.is_myers() ? (uint32_t)nv.k[p] : 0;
        if (m < kKeyMax * (k + 1) || m > 2047)
            return false; // (k + 1 disjoint 16-symbol keys must fit whatever the anchors)
    }
    return true;
}

// the needles of [begin, end) (every step-th one) that have no layout with this anchor set: counted, and listed up to `cap`
inline uint64_t dense_uncovered(const needle_view &nv, uint32_t dimers, uint32_t cmax, size_t begin, size_t end, size_t step,
                                std::vector<uint32_t> *list = nullptr, size_t cap = 0)
{
    uint64_t bad = 0;
    uint16_t pos[kDenseMaxPieces];
    for (size_t p = begin; p < end; p += step) {
        uint32_t n = 0;
        const uint32_t k = nv.is_myers() ? (uint32_t)nv.k[p] : 0;
        if (dense_layout(nv.ranks + nv.offsets[p], (uint32_t)nv.m[p], k, dimers, cmax, pos, n) == 0) {
            ++bad;
            if (list && list->size() < cap)
                list->push_back((uint32_t)p);
        }
    }
    return bad;
}

// Densities are tried in ascending order.  At every rung the candidate sets are ranked by how many needles they leave
// without a layout -- on a sample of the set and on the HARD needles, those that a set tried earlier could not place (a
// poly-A needle only ever begins with AA: one such needle decides which sets are worth trying at all) --, and the best one
// is tried on all needles; what it fails on joins the hard list and the rung is ranked once more before the next one.
inline bool choose_dense_anchors(const needle_view &nv, const index_tuning &T, dense_anchor_set &out)
{
    const unsigned nt = T.n_threads();
    const uint32_t cmax = (uint32_t)T.dense_cmax;
    std::vector<uint32_t> hard;
    auto uncovered_all = [&](uint32_t dimers) {
        std::vector<uint64_t> part(nt, 0);
        std::vector<std::vector<uint32_t>> lists(nt);
        parallel_slices(nv.n, nt, [&](size_t b, size_t e, unsigned t) { part[t] = dense_uncovered(nv, dimers, cmax, b, e, 1, &lists[t], 8); });
        uint64_t s = 0;
        for (unsigned t = 0; t < nt; ++t) {
            s += part[t];
            for (uint32_t p : lists[t])
                if (hard.size() < 256)
                    hard.push_back(p);
        }
        return s;
    };
    const size_t sample_step = std::max<size_t>(1, nv.n / 1024);
    // rank the candidates of one rung (sample: once; hard needles: at every attempt) and try the best untried one on all
    // needles, a few times; true: `out` covers every needle
    auto try_rung = [&](const std::vector<dense_anchor_set> &cands, dense_anchor_set &best_seen) {
        std::vector<uint64_t> base(cands.size(), 0);
        parallel_slices(cands.size(), nt, [&](size_t b, size_t e, unsigned) {
            for (size_t i = b; i < e; ++i)
                base[i] = dense_uncovered(nv, cands[i].dimers, cmax, 0, nv.n, sample_step);
        });
        std::vector<uint8_t> tried(cands.size(), 0);
        uint64_t best_fail = ~0ull;
        for (int attempt = 0; attempt < 4; ++attempt) {
            size_t best = cands.size();
            uint64_t best_bad = ~0ull;
            for (size_t i = 0; i < cands.size(); ++i) {
                if (tried[i])
                    continue;
                uint64_t bad = base[i];
                for (uint32_t p : hard)
                    bad += 4096 * dense_uncovered(nv, cands[i].dimers, cmax, p, (size_t)p + 1, 1);
                if (bad < best_bad) {
                    best_bad = bad;
                    best = i;
                }
            }
            if (best == cands.size())
                break;
            tried[best] = 1;
            const size_t hard_before = hard.size();
            const uint64_t fail = uncovered_all(cands[best].dimers);
            if (fail < best_fail) {
                best_fail = fail;
                best_seen = cands[best];
            }
            if (fail == 0) {
                out = cands[best];
                return true;
            }
            if (hard.size() == hard_before || fail > nv.n / 2000 + 4)
                break; // (nothing new to learn from / this density is far from enough: the next rung)
            // The needles it failed on: if most other sets of this rung place them, the failure is a matter of numbers (a
            // hundred thousand needles, each set unlucky with one or two) and the next set will meet its own: the next
            // rung.  If few sets place them (a poly-A needle wants AA), they now steer the ranking: once more.
            uint64_t placed = 0, asked = 0;
            for (size_t h = hard_before; h < hard.size(); ++h)
                for (size_t i = 0; i < cands.size(); i += 3) {
                    ++asked;
                    placed += dense_uncovered(nv, cands[i].dimers, cmax, hard[h], (size_t)hard[h] + 1, 1) == 0 ? 1 : 0;
                }
            if (2 * placed > asked)
                break;
        }
        return false;
    };
    // all patterns with `care` compared bits
    auto patterns = [&](uint32_t care) {
        std::vector<std::pair<uint32_t, uint32_t>> r;
        for (uint32_t cm = 0; cm < 16; ++cm) {
            if ((uint32_t)__builtin_popcount(cm) != care)
                continue;
            for (uint32_t c = 0; c < 16; ++c)
                if ((c & ~cm) == 0)
                    r.emplace_back(c, cm);
        }
        return r;
    };
    const dense_anchor_set none;
    auto singles = [&](uint32_t care) {
        std::vector<dense_anchor_set> r;
        for (const auto &pc : patterns(care))
            r.push_back(anchor_union(none, pc.first, pc.second));
        return r;
    };
    auto extended = [&](const dense_anchor_set &base, uint32_t care) {
        std::vector<dense_anchor_set> r;
        for (const auto &pc : patterns(care)) {
            const uint32_t add = dimer_mask_of(pc.first, pc.second);
            if ((add & base.dimers) == 0) // (disjoint: the density really grows by the pattern's share)
                r.push_back(anchor_union(base, pc.first, pc.second));
        }
        return r;
    };
    // the ladder: 2, 3, 4, 5, 6, 8, 12, 16 sixteenths of the dimers are anchors (a rung is built only if the ones below it
    // leave a needle without a layout).  Beyond dense_max_density the set is not worth a dense pass (unless forced).
    const int max_density = T.dense >= 2 ? 16 : T.dense_max_density;
    dense_anchor_set e8, e4, e2;
    for (int rung = 0; rung < 8; ++rung) {
        std::vector<dense_anchor_set> c;
        switch (rung) {
        case 0:
            // one pattern with a don't-care bit, or any two single dimers (two patterns: ~5 VALU more per 16 windows)
            c = singles(3);
            for (uint32_t d0 = 0; d0 < 16; ++d0)
                for (uint32_t d1 = d0 + 1; d1 < 16; ++d1)
                    if (__builtin_popcount(d0 ^ d1) != 1) // (those are the single patterns)
                        c.push_back(anchor_union(anchor_union(none, d0, 15), d1, 15));
            break;
        case 1: c = extended(e8, 4); break;
        case 2: c = singles(2); break;
        case 3: c = extended(e4, 4); break;
        case 4: c = extended(e4, 3); break;
        case 5: c = singles(1); break;
        case 6: c = extended(e2, 2); break;
        default: c.push_back(anchor_union(none, 0, 0)); break; // every window
        }
        if (c.empty() || (int)c[0].sixteenths() < T.dense_min_density)
            continue;
        if ((int)c[0].sixteenths() > max_density)
            return false;
        dense_anchor_set seen;
        if (try_rung(c, seen))
            return true;
        if (rung == 0)
            e8 = seen;
        else if (rung == 2)
            e4 = seen;
        else if (rung == 5)
            e2 = seen;
    }
    return false;
}

// Which windows become keys decides how many presence bits are set, i.e. how many text windows pass level 1 and cost a
// gather from L2 -- the resource the dense pass runs out of first.  A needle usually has several layouts (150 symbols, 3/16
// of the dimers: ~25 anchored windows for 4 keys), so every needle with disjoint pieces (c = 1) takes the layout whose keys
// hit the most bits that OTHER needles have set already (dynamic programme over its anchored windows: best number of shared
// bits with j keys from window i on).  Needles are taken in rounds of n / 32: a round reads the bit counts the rounds before it
// left (and its own needles' previous choice, which does not count), then its changes are applied -- the result does not
// depend on the number of threads.  Two sweeps (the first needles chose when the table was empty): random 150-symbol
// needles, k = 3, 100 000 of them: 31.7 % of the 2^20 bits set with first-fit keys, 18.4 % after one sweep, 16.0 % after two.
inline void dense_share_bits(const needle_view &nv, const index_tuning &T, uint32_t dimers, const std::vector<uint8_t> &cc,
                             const std::vector<uint32_t> &first, std::vector<uint16_t> &pos_flat)
{
    if (T.dense_sweeps <= 0)
        return;
    thread_team team(T.n_threads());
    std::vector<uint16_t> ref(1u << kDenseBloomBits, 0); // keys per presence bit
    auto key_at = [&](const uint8_t *pat, uint32_t i) {
        uint32_t key = 0;
        for (uint32_t x = 0; x < kKeyMax; ++x)
            key |= (uint32_t)(pat[i + x] & 3u) << (2 * x);
        return key;
    };
    std::vector<uint8_t> placed(nv.n, 0); // the needle's keys are counted in ref
    std::vector<uint32_t> bit_of(pos_flat.size(), 0), fresh_bit(pos_flat.size(), 0); // presence bit of every chosen key
    for (uint32_t p = 0; p < nv.n; ++p)
        if (cc[p] != 1) { // (overlapping pieces keep their first-fit keys)
            for (uint32_t s = first[p]; s < first[p + 1]; ++s)
                ++ref[bit_of[s] = dense_bloom_index(key_at(nv.ranks + nv.offsets[p], pos_flat[s]))];
            placed[p] = 1;
        }
    constexpr uint32_t kRounds = 24;
    std::vector<uint16_t> fresh(pos_flat.size());
    // The round's changes to the counters, bucketed by counter range: thread t files its needles' changes under the range
    // they fall in, then thread r applies everything filed under range r -- every change read once, every counter touched
    // by one thread (no atomics, the same result whatever the thread count), and that thread's 1/nt of the 2 MB of counters
    // stays in its cache (the updates are random: done by one thread they were a third of this function's time).
    const unsigned nt = team.size();
    std::vector<std::vector<std::vector<uint32_t>>> upd(nt, std::vector<std::vector<uint32_t>>(nt));
    auto range_of = [&](uint32_t bit) { return (unsigned)(((uint64_t)bit * nt) >> kDenseBloomBits); };
    for (int sweep = 0; sweep < T.dense_sweeps; ++sweep)
        for (uint32_t round = 0; round < kRounds; ++round) {
            const size_t rb = (size_t)nv.n * round / kRounds, re = (size_t)nv.n * (round + 1) / kRounds;
            for (auto &per_thread : upd) // (every thread's files, also of threads that get no slice this round)
                for (auto &v : per_thread)
                    v.clear();
            team.run(re - rb, [&](size_t b, size_t e, unsigned tid) {
                std::vector<uint16_t> cpos;
                std::vector<uint32_t> cbit, nxt;
                std::vector<uint8_t> shared;
                std::vector<int16_t> dp;
                for (size_t p = rb + b; p < rb + e; ++p) {
                    const uint32_t need = first[p + 1] - first[p];
                    if (cc[p] != 1 || need == 0)
                        continue;
                    const uint8_t *pat = nv.ranks + nv.offsets[p];
                    const uint32_t m = (uint32_t)nv.m[p];
                    uint32_t own[kDenseMaxPieces];
                    for (uint32_t j = 0; j < need; ++j)
                        own[j] = placed[p] ? bit_of[first[p] + j] : 0xFFFFFFFFu;
                    // the anchored windows, their bits, and whether somebody else has set them
                    cpos.clear();
                    cbit.clear();
                    shared.clear();
                    uint32_t key = key_at(pat, 0);
                    for (uint32_t i = 0; i + kKeyMax <= m; ++i) {
                        if (i)
                            key = (key >> 2) | ((uint32_t)(pat[i + kKeyMax - 1] & 3u) << 30);
                        if (!((dimers >> (key & 15u)) & 1u))
                            continue;
                        const uint32_t bit = dense_bloom_index(key);
                        uint32_t mine = 0;
                        for (uint32_t j = 0; j < need; ++j)
                            mine += own[j] == bit ? 1u : 0u;
                        cpos.push_back((uint16_t)i);
                        cbit.push_back(bit);
                        shared.push_back(ref[bit] > mine ? 1 : 0);
                    }
                    const uint32_t nc = (uint32_t)cpos.size();
                    nxt.assign(nc + 1, nc); // first candidate that does not overlap candidate i
                    for (uint32_t i = 0, t = 0; i < nc; ++i) {
                        while (t < nc && cpos[t] < cpos[i] + kKeyMax)
                            ++t;
                        nxt[i] = t;
                    }
                    // dp[j][i]: most shared bits with j keys among candidates i.., -1: no such layout
                    dp.assign((size_t)(need + 1) * (nc + 1), -1);
                    for (uint32_t i = 0; i <= nc; ++i)
                        dp[i] = 0;
                    for (uint32_t j = 1; j <= need; ++j)
                        for (uint32_t i = nc; i-- > 0;) {
                            const int16_t skip = dp[(size_t)j * (nc + 1) + i + 1];
                            const int16_t rest = dp[(size_t)(j - 1) * (nc + 1) + nxt[i]];
                            const int16_t take = rest < 0 ? (int16_t)-1 : (int16_t)(rest + shared[i]);
                            dp[(size_t)j * (nc + 1) + i] = take >= skip ? take : skip;
                        }
                    if (dp[(size_t)need * (nc + 1)] < 0) { // (cannot happen: first fit found a layout)
                        for (uint32_t j = 0; j < need; ++j) {
                            fresh[first[p] + j] = pos_flat[first[p] + j];
                            fresh_bit[first[p] + j] = dense_bloom_index(key_at(pat, pos_flat[first[p] + j]));
                        }
                    } else
                    for (uint32_t i = 0, j = need; j >= 1;) {
                        const int16_t rest = dp[(size_t)(j - 1) * (nc + 1) + nxt[i]];
                        const int16_t take = rest < 0 ? (int16_t)-1 : (int16_t)(rest + shared[i]);
                        if (take >= 0 && take == dp[(size_t)j * (nc + 1) + i] && take >= dp[(size_t)j * (nc + 1) + i + 1]) {
                            fresh[first[p] + (need - j)] = cpos[i];
                            fresh_bit[first[p] + (need - j)] = cbit[i];
                            i = nxt[i];
                            --j;
                        } else {
                            ++i;
                        }
                    }
                    // this needle's changes: filed for the counters (bit | 1 << 31: one key less), written back for the needle
                    // (nobody else reads a needle's own positions and bits)
                    for (uint32_t s2 = first[p]; s2 < first[p + 1]; ++s2) {
                        if (placed[p])
                            upd[tid][range_of(bit_of[s2])].push_back(bit_of[s2] | 0x80000000u);
                        upd[tid][range_of(fresh_bit[s2])].push_back(fresh_bit[s2]);
                        pos_flat[s2] = fresh[s2];
                        bit_of[s2] = fresh_bit[s2];
                    }
                    placed[p] = 1;
                }
            });
            team.run(nt, [&](size_t b, size_t e, unsigned) {
                for (size_t r = b; r < e; ++r)
                    for (unsigned t2 = 0; t2 < nt; ++t2)
                        for (uint32_t u : upd[t2][r]) {
                            if (u & 0x80000000u)
                                --ref[u & 0x7FFFFFFFu];
                            else
                                ++ref[u];
                        }
            });
        }
}

inline bool index_trace_on()
{
    const char *v = getenv("SPM_HIP_TRACE");
    return v && *v && *v != '0';
}

inline int build_dense_index(const needle_view &nv, const index_tuning &T, seed_index &X)
{
    using iclk = std::chrono::steady_clock;
    const auto t0 = iclk::now();
    auto ms = [&](iclk::time_point a) { return std::chrono::duration<double, std::milli>(iclk::now() - a).count(); };
    dense_anchor_set A;
    if (!choose_dense_anchors(nv, T, A))
        return SPM_OK; // (some needle has no layout even with every window looked up: not a dense set)
    const double ms_anchors = ms(t0);
    const auto t1 = iclk::now();
    const unsigned nt = T.n_threads();
    const uint32_t cmax = (uint32_t)T.dense_cmax;
    // ---- layouts, per needle (threads): key positions first (first fit), then the keys laid out back to back ----
    std::vector<uint8_t> cc(nv.n, 0), nn(nv.n, 0);
    std::vector<uint32_t> first(nv.n + 1, 0);
    std::vector<std::vector<uint16_t>> pos_of(nt); // thread t: the positions of its slice of needles, back to back
    parallel_slices(nv.n, nt, [&](size_t b, size_t e, unsigned t) {
        uint16_t pos[kDenseMaxPieces];
        std::vector<uint16_t> &out = pos_of[t];
        out.reserve((e - b) * 5);
        for (size_t p = b; p < e; ++p) {
            uint32_t n = 0;
            const uint32_t k = nv.is_myers() ? (uint32_t)nv.k[p] : 0;
            cc[p] = (uint8_t)dense_layout(nv.ranks + nv.offsets[p], (uint32_t)nv.m[p], k, A.dimers, cmax, pos, n);
            nn[p] = (uint8_t)n;
            out.insert(out.end(), pos, pos + n);
        }
    });
    for (uint32_t p = 0; p < nv.n; ++p)
        first[p + 1] = first[p] + nn[p];
    const size_t n_keys = first[nv.n];
    std::vector<uint16_t> pos_flat(n_keys);
    {
        size_t at = 0;
        for (unsigned t = 0; t < nt; ++t) { // (the slices of parallel_slices are contiguous and ascending)
            std::copy(pos_of[t].begin(), pos_of[t].end(), pos_flat.begin() + at);
            at += pos_of[t].size();
            pos_of[t] = std::vector<uint16_t>();
        }
    }
    const double ms_first_fit = ms(t1);
    dense_share_bits(nv, T, A.dimers, cc, first, pos_flat);
    const double ms_share = ms(t1) - ms_first_fit;
    X.seed_q.assign(nv.n, 0);
    X.seed_n.assign(nv.n, 0);
    X.seed_c.assign(nv.n, 1);
    X.seed_first = first;
    X.seed_off.assign(n_keys, 0);
    X.seed_len.assign(n_keys, 0);
    std::vector<index_kv> keys(n_keys);
    parallel_slices(nv.n, nt, [&](size_t b, size_t e, unsigned) {
        for (size_t p = b; p < e; ++p) {
            const uint16_t *pos = pos_flat.data() + first[p];
            const uint32_t m = (uint32_t)nv.m[p], c = cc[p], n = nn[p];
            X.seed_n[p] = (uint16_t)n;
            X.seed_c[p] = (uint8_t)c;
            for (uint32_t j = 0; j < n; ++j) {
                // the piece around key j: c = 1: the key plus what lies between it and its neighbours' halves of the gaps,
                // at most 8 symbols before and 16 in all; c > 1: the key alone
                uint32_t lo = pos[j], hi = pos[j] + kKeyMax;
                if (c == 1) {
                    const uint32_t gap_l = j == 0 ? pos[j] : (pos[j] - (pos[j - 1] + kKeyMax)) / 2;
                    const uint32_t gap_r = j + 1 == n ? m - hi : (pos[j + 1] - hi + 1) / 2;
                    const uint32_t before = std::min<uint32_t>(8, gap_l);
                    const uint32_t after = std::min<uint32_t>(16 - before, gap_r);
                    lo -= before;
                    hi += after;
                }
                const size_t s = (size_t)first[p] + j;
                X.seed_off[s] = (uint16_t)lo;
                X.seed_len[s] = (uint16_t)(hi - lo);
                keys[s] = make_kv(nv, seed_key{(uint32_t)p, lo, pos[j] - lo, hi - lo}, kKeyMax);
            }
        }
    });
    const double ms_layout = ms(t1);
    const auto t2 = iclk::now();
    filter_index F;
    F.dense = 1;
    F.key_len = kKeyMax;
    F.stride = 1;
    F.n_pat = A.n_pat;
    for (uint32_t i = 0; i < A.n_pat; ++i) {
        F.pat_c[i] = A.c[i];
        F.pat_cm[i] = A.cm[i];
    }
    F.dimer_set = A.dimers;
    F.n_keys = n_keys;
    F.n_entries = n_keys;
    F.hash_variant = 3;
    // ---- level 1: presence bits; level 1b: fingerprint buckets (kDenseSlots per bucket, about two keys per bucket) ----
    build_bits_level1(keys, F);
    const double ms_level1 = ms(t2);
    const auto t3 = iclk::now();
    // ---- exact level ----
    std::vector<uint16_t> ranges(keys.size());
    for (size_t i = 0; i < keys.size(); ++i)
        ranges[i] = (uint16_t)keys[i].meta;
    build_directory(keys, ranges, F, X.h_entries);
    if (index_trace_on()) {
        uint64_t set = 0;
        for (uint32_t w : F.h_image)
            set += (uint64_t)__builtin_popcount(w);
        fprintf(stderr, "[spm_hip] dense index: %zu keys, anchors %u/16 (%u pattern(s)), %.1f %% of the presence bits set; anchors %.2f ms, "
                        "layouts %.2f (first fit %.2f, shared bits %.2f), bits + buckets %.2f, directory %.2f (%u threads)\n", n_keys,
                A.sixteenths(), A.n_pat, 100.0 * (double)set / (double)(1u << kDenseBloomBits), ms_anchors, ms_layout, ms_first_fit,
                ms_share, ms_level1, ms(t3), nt);
    }
    F.ok = true;
    X.fidx.push_back(std::move(F));
    X.filter_stride = 1;
    X.filter_key_len = kKeyMax;
    X.filter_anchored = false;
    X.filter_dense = true;
    X.filter_max_range = 0;
    return SPM_OK;
}

// Seed filter applicability + partition of the needle set into passes whose keys fit one LDS table (or the one dense pass).
inline int build_filter_index(const needle_view &nv, const index_tuning &T, seed_index &X)
{
    X = seed_index();
    if ((nv.sigma != 4 && nv.sigma != 5 && nv.sigma != 15) || nv.algo == SPM_ALGO_MYERS_PREFIX || nv.n == 0 ||
        nv.n >= (1u << 21))
        return SPM_OK;
    uint32_t qmin = 0xFFFFFFFFu;
    uint64_t n_seeds = 0;
    X.seed_q.assign(nv.n, 0);
    X.seed_n.assign(nv.n, 0);
    X.seed_first.assign(nv.n + 1, 0);
    std::vector<uint16_t> off;
    bool sparse_ok = true;
    for (uint32_t p = 0; p < nv.n && sparse_ok; ++p) {
        const uint32_t m = (uint32_t)nv.m[p];
        if (m == 0 || m > 2047)
            return SPM_OK;
        uint32_t n = 0, q = 0;
        if (!layout_seeds(nv, p, kKeyMin, n, q, off)) {
            sparse_ok = false; // (one needle without a layout keeps the whole set off the sparse passes)
            break;
        }
        X.seed_q[p] = (uint16_t)q;
        X.seed_n[p] = (uint16_t)n;
        X.seed_first[p] = (uint32_t)X.seed_off.size();
        X.seed_off.insert(X.seed_off.end(), off.begin(), off.end());
        qmin = std::min(qmin, q);
        n_seeds += n;
    }
    X.seed_first[nv.n] = (uint32_t)X.seed_off.size();
    if (sparse_ok && qmin < kKeyMin)
        sparse_ok = false;
    // key length H and stride S: a window of H symbols at every S-th text position needs S <= q - H + 1.
    // Seeds of >= 17 symbols use full 32-bit keys; shorter seeds give up one or two symbols of key for stride 2
    // (half the windows), which costs far less than the extra spurious key matches it lets through.
    // Seeds of <= 12 symbols: the whole seed is the key (stride 1) -- every symbol of key divides the chance matches by 4.
    uint32_t H = kKeyMax;
    uint32_t Smax = 1;
    if (sparse_ok) {
        if (qmin < kKeyMax + 1)
            H = qmin <= 12 ? qmin : qmin - 1;
        if ((double)n_seeds / std::pow(4.0, (double)H) > kMaxSurvivorShare)
            sparse_ok = false; // too many keys for their length: most text windows would match one by chance
    }
    if (sparse_ok) {
        const int force_h = T.force_keylen;
        if (force_h >= (int)kKeyMin && force_h <= (int)std::min(qmin, kKeyMax))
            H = (uint32_t)force_h;
        while (Smax * 2 <= 16 && Smax * 2 <= qmin - (H - 1))
            Smax *= 2;
        if (T.force_stride > 0 && (uint32_t)T.force_stride <= Smax)
            Smax = (uint32_t)T.force_stride;
    }
    // keys per pass: the fingerprint table has 65536 slots; the hash-and-displace build succeeds up to ~88 % load
    // (57 344 keys).  If a dense batch cannot be placed, the whole set is re-partitioned with smaller batches rather
    // than dropping to the Bloom cascade.
    const uint64_t cap0 = (uint64_t)T.max_keys;
    const bool want_chd = T.hash == 2;
    // ---- the dense pass instead? ----
    // A sparse pass streams the text at the HBM rate as long as it looks at <= 4 windows per 16 symbols of ONE table.  A
    // set that needs several such passes, or stride 1 (16 windows per 16 symbols: LDS- and VALU-bound at ~3x the HBM time),
    // is better off with the one dense pass (~2x the HBM time whatever the number of keys).
    if (T.dense != 0 && dense_eligible(nv)) {
        bool want = T.dense >= 2 || !sparse_ok;
        if (!want) {
            uint32_t S = Smax;
            double best = 1e300;
            for (uint32_t s = Smax; s >= 1; s >>= 1) {
                const double passes = (double)((n_seeds * s + cap0 - 1) / cap0);
                const double cost = passes * (1.0 + 0.3 * ((double)Smax / s - 1.0));
                if (cost < best) {
                    best = cost;
                    S = s;
                }
            }
            const uint64_t passes = (n_seeds * S + cap0 - 1) / cap0;
            want = passes >= 2 || S == 1;
        }
        if (want) {
            seed_index D;
            const int rc = build_dense_index(nv, T, D);
            if (rc != SPM_OK)
                return rc;
            if (!D.fidx.empty()) {
                X = std::move(D);
                return SPM_OK;
            }
        }
    }
    if (!sparse_ok) {
        X = seed_index();
        return SPM_OK;
    }
    const uint64_t caps[3] = {cap0, cap0 * 7 / 8, cap0 * 3 / 4};
    for (int attempt = 0; attempt < 3; ++attempt) {
        const uint64_t cap = caps[attempt];
        bool dense_failure = false;
        // stride: the largest one when a single pass suffices; otherwise the one minimising passes x cost per pass
        // (a pass with stride S looks at 16/S windows per 16 symbols; measured cost grows ~0.3x per doubling)
        uint32_t S = Smax;
        if (n_seeds * Smax > cap && T.force_stride <= 0) {
            double best = 1e300;
            for (uint32_t s = Smax; s >= 1; s >>= 1) {
                const double passes = (double)((n_seeds * s + cap - 1) / cap);
                const double cost = passes * (1.0 + 0.3 * ((double)Smax / s - 1.0));
                if (cost < best) {
                    best = cost;
                    S = s;
                }
            }
        }
        const uint32_t max_passes = (uint32_t)T.max_passes;
        X.filter_stride = S;
        X.filter_key_len = H;
        // ---- which windows are indexed, and in which pass ----
        std::vector<std::vector<seed_key>> pass_items;
        std::vector<uint32_t> pass_anchor;
        X.filter_anchored = false;
        const uint64_t n_passes0 = (n_seeds * S + cap - 1) / cap;
        if (S == 1 && n_passes0 > 1 && nv.sigma == 4 && H == 16 && qmin > H && T.anchor != 0) {
            // Anchored keys.  A set this large gets ONE key per seed (stride 1: every text window is looked up, in every
            // pass) -- but which of the seed's q - H + 1 windows that is, is ours to choose.  Pass i takes only keys whose
            // first two symbols (a "dimer", 4 bits: sym0 | sym1 << 2) match ITS anchor pattern: (dimer ^ c) & cm == 0, at
            // first one dimer per pass (cm = 15).  The streaming kernel then looks up only the text windows that begin
            // with the anchor -- 1 in 16 -- instead of all: a window beginning with anything else cannot equal a key of the
            // pass.  Lossless: an intact seed still has its key window in the text.  Each seed goes to a pass in which it
            // has such a window among its first 32 (fewest choices first, least-loaded pass).  A seed that finds no place
            // (seeds of 37 symbols, 22 windows, 7 passes: a few in a million) widens a pass's pattern by one don't-care bit
            // -- that pass looks up 2 in 16 windows.
            constexpr uint32_t kAnchorWindows = 32; // (an entry records where its window sits in the seed in 5 bits)
            struct cand
            {
                uint32_t p, o, dimers, n_ok; // dimers: bit d set = one of the windows r <= min(31, q - H) begins with d
            };
            std::vector<cand> seeds;
            seeds.reserve(n_seeds);
            for (uint32_t p = 0; p < nv.n; ++p) {
                const uint8_t *pat = nv.ranks + nv.offsets[p];
                const uint32_t q = X.seed_q[p];
                for (uint32_t j = 0; j < X.seed_n[p]; ++j) {
                    const uint32_t o = X.seed_off[X.seed_first[p] + j];
                    uint32_t dm = 0;
                    for (uint32_t r = 0; r <= std::min<uint32_t>(kAnchorWindows - 1, q - H); ++r)
                        dm |= 1u << ((pat[o + r] & 3u) | ((pat[o + r + 1] & 3u) << 2));
                    seeds.push_back({p, o, dm, 0});
                }
            }
            const uint32_t np_min = (uint32_t)((n_seeds + cap - 1) / cap);
            for (uint32_t np = np_min; np <= np_min + 1 && pass_items.empty(); ++np) {
                std::vector<uint32_t> pc(np), pcm(np, 15u), sets(np);
                for (uint32_t i = 0; i < np; ++i) {
                    pc[i] = (5u * i + 3u) & 15u; // (a fixed shuffle of the dimers: neighbouring passes differ in both symbols)
                    sets[i] = dimer_mask_of(pc[i], pcm[i]);
                }
                for (cand &c : seeds) {
                    c.n_ok = 0;
                    for (uint32_t i = 0; i < np; ++i)
                        c.n_ok += (c.dimers & sets[i]) ? 1u : 0u;
                }
                std::vector<uint32_t> order(seeds.size());
                for (uint32_t i = 0; i < order.size(); ++i)
                    order[i] = i;
                std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return seeds[x].n_ok < seeds[y].n_ok; });
                std::vector<std::vector<uint32_t>> members(np);
                bool all = true;
                for (uint32_t idx : order) {
                    const cand &c = seeds[idx];
                    uint32_t best = np;
                    for (uint32_t i = 0; i < np; ++i)
                        if ((c.dimers & sets[i]) && members[i].size() < cap && (best == np || members[i].size() < members[best].size()))
                            best = i;
                    if (best == np) { // no place: widen the narrowest pattern that then takes this seed
                        uint32_t bbit = 0;
                        for (uint32_t i = 0; i < np; ++i) {
                            if (members[i].size() >= cap)
                                continue;
                            for (uint32_t bit = 1; bit < 16; bit <<= 1)
                                if ((pcm[i] & bit) && (c.dimers & dimer_mask_of(pc[i], pcm[i] & ~bit)) &&
                                    (best == np || __builtin_popcount(pcm[i]) > __builtin_popcount(pcm[best]))) {
                                    best = i;
                                    bbit = bit;
                                }
                        }
                        if (best == np) {
                            all = false;
                            break;
                        }
                        pcm[best] &= ~bbit;
                        sets[best] = dimer_mask_of(pc[best], pcm[best]);
                    }
                    members[best].push_back(idx);
                }
                if (!all)
                    continue;
                pass_items.resize(np);
                for (uint32_t i = 0; i < np; ++i) {
                    for (uint32_t idx : members[i]) {
                        const cand &c = seeds[idx];
                        const uint8_t *pat = nv.ranks + nv.offsets[c.p];
                        uint32_t r = 0;
                        while (!((sets[i] >> ((pat[c.o + r] & 3u) | ((pat[c.o + r + 1] & 3u) << 2))) & 1u))
                            ++r;
                        pass_items[i].push_back({c.p, c.o, r, X.seed_q[c.p]});
                    }
                    pass_anchor.push_back(pc[i] | (pcm[i] << 4));
                }
                X.filter_anchored = true;
            }
        }
        if (pass_items.empty()) {
            uint32_t p0 = 0;
            while (p0 < nv.n) {
                uint64_t keys = 0;
                uint32_t p1 = p0;
                while (p1 < nv.n) {
                    const uint64_t add = (uint64_t)X.seed_n[p1] * S;
                    if (keys + add > cap && p1 > p0)
                        break;
                    keys += add;
                    ++p1;
                }
                pass_items.emplace_back();
                pass_anchor.push_back(0u); // (no bit of the dimer is compared: every window is looked up)
                for (uint32_t p = p0; p < p1; ++p)
                    for (uint32_t j = 0; j < X.seed_n[p]; ++j)
                        for (uint32_t r = 0; r < S; ++r)
                            pass_items.back().push_back({p, X.seed_off[X.seed_first[p] + j], r, X.seed_q[p]});
                p0 = p1;
            }
        }
        if (pass_items.size() > max_passes) {
            X = seed_index();
            return SPM_OK; // too many passes to be worth it: brute force
        }
        // the passes are independent: build them side by side, then append their entries in pass order
        const size_t np = pass_items.size();
        std::vector<filter_index> Fs(np);
        std::vector<std::vector<u32x4>> ents(np);
        std::vector<int> rcs(np, SPM_OK);
        {
            std::atomic<size_t> next{0};
            const unsigned nt = (unsigned)std::min<size_t>(T.n_threads(), np);
            auto work = [&]() {
                for (size_t pi = next++; pi < np; pi = next++) {
                    filter_index &F = Fs[pi];
                    F.key_len = H;
                    F.anchor_c = pass_anchor[pi] & 15u;
                    F.anchor_cm = pass_anchor[pi] >> 4;
                    F.dimer_set = dimer_mask_of(F.anchor_c, F.anchor_cm);
                    rcs[pi] = build_one_index(nv, T, pass_items[pi], S, F, ents[pi]);
                }
            };
            if (nt <= 1) {
                work();
            } else {
                std::vector<std::thread> th;
                for (unsigned t = 0; t < nt; ++t)
                    th.emplace_back(work);
                for (std::thread &x : th)
                    x.join();
            }
        }
        bool all_ok = true;
        for (size_t pi = 0; pi < np; ++pi) {
            if (rcs[pi] != SPM_OK)
                return rcs[pi];
            all_ok = all_ok && Fs[pi].ok;
        }
        X.fidx.clear();
        X.h_entries.clear();
        if (!all_ok) {
            X = seed_index();
            return SPM_OK;
        }
        for (size_t pi = 0; pi < np; ++pi) {
            // (a pass numbered its entries from 0: move its directory to where its entries land)
            const uint32_t base = (uint32_t)X.h_entries.size();
            if (base)
                for (u32x4 &d : Fs[pi].h_ht)
                    if (d.z != 0)
                        d.y += base;
            X.h_entries.insert(X.h_entries.end(), ents[pi].begin(), ents[pi].end());
            X.fidx.push_back(std::move(Fs[pi]));
        }
        X.filter_max_range = 0;
        for (const filter_index &F : X.fidx) {
            dense_failure = dense_failure || (want_chd && F.hash_variant != 2 && F.hash_variant != 4);
            X.filter_max_range = std::max(X.filter_max_range, F.max_range);
        }
        if (!dense_failure || attempt == 2)
            return SPM_OK;
        X.fidx.clear();
        X.h_entries.clear();
    }
    return SPM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Host-only self-check of the seed index (no device): builds the tables exactly as spm_hip_patterns_create does and
// verifies the properties the filter's losslessness rests on.  stats: see include/spm_hip.h.
// ---------------------------------------------------------------------------------------------------------------------
inline int host_selftest(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                         const uint16_t *k, uint32_t sigma, uint64_t *stats)
{
    if (!ranks_concat || !offsets || !stats || n_patterns == 0)
        return SPM_E_INVALID;
    std::vector<int32_t> mm(n_patterns, 0), kk(n_patterns, 0);
    needle_view nv;
    nv.algo = algo;
    nv.n = n_patterns;
    nv.sigma = sigma;
    nv.ranks = ranks_concat;
    nv.offsets = offsets;
    for (uint32_t p = 0; p < n_patterns; ++p) {
        mm[p] = (int32_t)(offsets[p + 1] - offsets[p]);
        kk[p] = (nv.is_myers() && k) ? k[p] : 0;
        nv.max_k = std::max<uint32_t>(nv.max_k, (uint32_t)kk[p]);
    }
    nv.m = mm.data();
    nv.k = kk.data();
    seed_index ps;
    const int rc = build_filter_index(nv, index_tuning::from_env(), ps);
    memset(stats, 0, 8 * sizeof(uint64_t));
    if (rc != SPM_OK)
        return rc;
    stats[0] = ps.fidx.size();   // passes (0 = the seed filter does not apply)
    stats[1] = ps.filter_stride | ((uint64_t)ps.filter_key_len << 32); // S | H << 32
    if (ps.fidx.empty())
        return SPM_OK;
    const uint32_t S = ps.filter_stride;
    // level-1 membership test, exactly as the kernels evaluate it
    auto level1 = [&](const filter_index &F, uint32_t key) -> bool {
        if (F.dense || F.hash_variant == 4) {
            if (F.dense && !((F.dimer_set >> (key & 15u)) & 1u))
                return false; // the streaming kernel does not look this window up
            const uint32_t b = dense_bloom_index(key);
            if (!((F.h_image[b >> 5] >> (b & 31)) & 1u))
                return false;
            const uint16_t *bk = F.h_buckets.data() + (size_t)dense_bucket(key, F.bucket_shift) * kDenseSlots;
            if (bk[kDenseSlots - 1] == kDenseAcceptAll)
                return true;
            for (uint32_t s = 0; s < kDenseSlots; ++s)
                if (bk[s] == dense_fp(key))
                    return true;
            return false;
        }
        if (F.hash_variant == 2) {
            const uint16_t *fp = reinterpret_cast<const uint16_t *>(F.h_image.data());
            const uint16_t *disp = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(F.h_image.data()) + F.chd_disp_off);
            const chd_hashes hh = chd_hash(key);
            const uint32_t d = disp[hh.x >> F.chd_bucket_shift];
            return fp[chd_slot(hh, d, F.chd_slot_mask)] == hh.f;
        }
        const uint32_t idx_mask = F.bitmap_words * 32 - 1;
        for (uint32_t pr = 0; pr < F.n_probes; ++pr) {
            const uint32_t x = (F.hash_variant ? bloom_hash<1>(key, pr) : bloom_hash<0>(key, pr)) & idx_mask;
            if (!((F.h_image[x >> 5] >> (x & 31)) & 1))
                return false;
        }
        return true;
    };
    auto level2 = [&](const filter_index &F, uint32_t key, uint32_t val) -> bool {
        uint32_t slot = ht_hash(key) & F.ht_mask;
        for (;;) {
            const u32x4 d = F.h_ht[slot];
            if (d.z == 0)
                return false;
            if (d.x == key) {
                for (uint32_t q = 0; q < d.z; ++q) {
                    const u32x4 e = ps.h_entries[d.y + q];
                    if (e.w != key)
                        return false; // the entries of a key lie side by side
                    if ((e.x >> 11) == (val >> 11)) {
                        // the entry itself, or a run whose diagonal range covers this offset
                        const uint32_t x0 = e.x & 0x7FF, x = val & 0x7FF;
                        if (!(e.z & kRngRun) ? x == x0 : (x >= x0 && x <= x0 + (e.z & 0x7FF)))
                            return true;
                    }
                }
                return false;
            }
            slot = (slot + 1) & F.ht_mask;
        }
    };
    uint64_t checked = 0, missing = 0;
    uint64_t keys_total = 0;
    for (const filter_index &F : ps.fidx)
        keys_total += F.n_keys;
    stats[2] = keys_total;
    if (ps.filter_dense) {
        // every needle: c k + 1 pieces, no position in more than c of them, each piece inside the needle and holding its
        // key window, which begins with an anchor and is found at all three levels with the piece's entry
        const filter_index &F = ps.fidx[0];
        uint64_t deep = 0; // needles with c > 1
        for (uint32_t p = 0; p < n_patterns; ++p) {
            const uint32_t m = (uint32_t)mm[p], kp = (uint32_t)kk[p], c = ps.seed_c[p], sn = ps.seed_n[p];
            if (c == 0 || sn < c * kp + 1)
                return SPM_E_INVALID;
            deep += c > 1 ? 1 : 0;
            std::vector<uint8_t> cover(m, 0);
            const uint8_t *pat = ranks_concat + offsets[p];
            for (uint32_t j = 0; j < sn; ++j) {
                const uint32_t o = ps.seed_off[ps.seed_first[p] + j], q = ps.seed_len[ps.seed_first[p] + j];
                if (q < kKeyMax || o + q > m)
                    return SPM_E_INVALID;
                for (uint32_t i = 0; i < q; ++i)
                    if (++cover[o + i] > c)
                        return SPM_E_INVALID;
                // the piece's entry: some window of the piece, anchored, with val = (p, window offset) and a signature
                // that spells the rest of the piece
                uint32_t found = 0;
                for (uint32_t r = 0; r + kKeyMax <= q; ++r) {
                    uint32_t key = 0;
                    for (uint32_t i = 0; i < kKeyMax; ++i)
                        key |= key_code(sigma, pat[o + r + i]) << (2 * i);
                    if (level1(F, key) && level2(F, key, (p << 11) | (o + r)))
                        ++found;
                }
                ++checked;
                if (found < 1)
                    ++missing;
            }
        }
        stats[3] = checked;
        stats[4] = missing;
        uint64_t fp = 0;
        const uint64_t trials = 1 << 20;
        for (uint64_t t = 0; t < trials; ++t)
            fp += level1(F, (uint32_t)mix64(0xC0FFEE + t)) ? 1 : 0;
        stats[5] = fp;
        stats[6] = trials;
        uint32_t sixteenths = 0;
        for (uint32_t d = 0; d < 16; ++d)
            sixteenths += (F.dimer_set >> d) & 1u;
        stats[7] = 3u | ((uint64_t)sixteenths << 8) | (deep << 16);
        return missing ? SPM_E_INVALID : SPM_OK;
    }
    {
        uint64_t expect = 0;
        for (uint32_t p = 0; p < n_patterns; ++p)
            expect += (uint64_t)ps.seed_n[p] * S;
        if (expect != keys_total)
            return SPM_E_INVALID;
    }
    // every needle belongs to exactly one pass; every indexed window of every seed is found at both levels
    size_t fi = 0;
    uint64_t in_pass = 0;
    for (uint32_t p = 0; p < n_patterns; ++p) {
        const uint32_t q = ps.seed_q[p], sn = ps.seed_n[p];
        if (S > q - (ps.filter_key_len - 1) || sn < (uint32_t)kk[p] + 1)
            return SPM_E_INVALID; // sampling would miss occurrences / too few seeds for the pigeonhole argument
        for (uint32_t j = 0; j < sn; ++j) { // seeds: inside the needle, disjoint, key symbols only
            const uint32_t o = ps.seed_off[ps.seed_first[p] + j];
            if (o + q > (uint32_t)mm[p] || (j && o < (uint32_t)ps.seed_off[ps.seed_first[p] + j - 1] + q))
                return SPM_E_INVALID;
            for (uint32_t i = 0; i < q; ++i)
                if (!key_symbol(sigma, ranks_concat[offsets[p] + o + i]))
                    return SPM_E_INVALID;
        }
        if (ps.filter_anchored) {
            // every seed has ONE key, in one pass, and that key begins with an anchor dimer of the pass (so the streaming
            // kernel, which looks up only such windows, meets it)
            const uint8_t *pat = ranks_concat + offsets[p];
            for (uint32_t j = 0; j < sn; ++j) {
                const uint32_t o = ps.seed_off[ps.seed_first[p] + j];
                uint32_t found = 0;
                for (const filter_index &F : ps.fidx)
                    for (uint32_t r = 0; r + ps.filter_key_len <= q && r < 32; ++r) {
                        uint32_t key = 0;
                        for (uint32_t i = 0; i < ps.filter_key_len; ++i)
                            key |= key_code(sigma, pat[o + r + i]) << (2 * i);
                        if ((((key & 0xF) ^ F.anchor_c) & F.anchor_cm) == 0 && level1(F, key) && level2(F, key, (p << 11) | (o + r)))
                            ++found;
                    }
                ++checked;
                if (found < 1)
                    ++missing;
            }
            continue;
        }
        const uint64_t mine = (uint64_t)sn * S;
        while (fi < ps.fidx.size() && in_pass + mine > ps.fidx[fi].n_keys) {
            if (in_pass != ps.fidx[fi].n_keys)
                return SPM_E_INVALID;
            ++fi;
            in_pass = 0;
        }
        if (fi >= ps.fidx.size())
            return SPM_E_INVALID;
        in_pass += mine;
        const filter_index &F = ps.fidx[fi];
        const uint8_t *pat = ranks_concat + offsets[p];
        for (uint32_t j = 0; j < sn; ++j)
            for (uint32_t r = 0; r < S; ++r) {
                const uint32_t o = ps.seed_off[ps.seed_first[p] + j];
                uint32_t key = 0;
                for (uint32_t i = 0; i < ps.filter_key_len; ++i)
                    key |= key_code(sigma, pat[o + r + i]) << (2 * i);
                ++checked;
                if (!level1(F, key) || !level2(F, key, (p << 11) | (o + r)))
                    ++missing;
            }
    }
    uint64_t anchor_sum = 0; // dimers looked up, over all passes
    for (const filter_index &F : ps.fidx)
        anchor_sum += 1ull << (4 - __builtin_popcount(F.anchor_cm));
    stats[3] = checked;
    stats[4] = missing;
    // false-positive rate of level 1 on pseudo-random keys (first pass)
    uint64_t fp = 0;
    const uint64_t trials = 1 << 20;
    for (uint64_t t = 0; t < trials; ++t)
        fp += level1(ps.fidx[0], (uint32_t)mix64(0xC0FFEE + t) &
                                     (ps.filter_key_len >= 16 ? 0xFFFFFFFFu : ((1u << (2 * ps.filter_key_len)) - 1)))
                  ? 1
                  : 0;
    stats[5] = fp;
    stats[6] = trials;
    stats[7] = ps.fidx[0].hash_variant | (ps.filter_anchored ? anchor_sum << 8 : 0);
    return missing ? SPM_E_INVALID : SPM_OK;
}

} // namespace spm_hip
