// filter.hpp -- lossless pigeonhole seed filter + bit-vector verification (dna4 texts).
//
// Why: with one lane per pattern every text byte feeds n_patterns * ~15*NW integer ops, so the brute-force scan is
// bound by the integer VALU four orders of magnitude below the HBM roof.  The reference itself
// ships the way out -- a q-gram pigeonhole filter in front of verification
// (/root/reference/libspm/libspm/matcher/pigeonhole_matcher.hpp:67-123) -- and this file is the
// MI355X formulation of that idea, arranged so that the text is streamed ONCE at HBM speed:
//
//   Pigeonhole: cut needle P (|P| = m, k errors) into k+1 disjoint seeds of q = floor(m/(k+1)) symbols.  Any
//   occurrence with <= k edits contains at least one seed unedited.  (Needles with k >= 8 get k+2 seeds: two of
//   them survive, on diagonals <= k apart -- see "candidate merging" below.)
//   Sampling:   an exact seed occurrence text[ts, ts+q) contains, for every stride S <= q-H+1, exactly one
//   H-symbol window that starts at a text position t = 0 (mod S); it equals seed[r, r+H) for r = t-ts in [0,S).
//   So it suffices to look at text windows at multiples of S and to index S shifted H-mers per seed.
//   H = 16 symbols = one 32-bit key after 2-bit packing (12..15 for seeds shorter than 17 symbols).
//
//   Level 1 (filter kernel, the streaming kernel): each lane loads 16 text bytes (one 16-byte coalesced load),
//   packs them to 2 bits/base with 4 v_dot4_u32_u8, takes the previous lane's word through DPP wave_shr:1, forms
//   the 16/S windows with v_alignbit and looks them up in a perfect-hash fingerprint table held in LDS (two LDS
//   reads per window; a Bloom cascade is kept as fallback).  Survivors are looked up in an exact key table in L2
//   and emitted as candidates (text position, needle, needle offset).
//   Level 2 (verification): the Myers recurrence over the m+3k symbols around the candidate diagonal, cold-started
//   m+k symbols before the first end position it is responsible for -- exact by the window property used for
//   tiling.  Short needles: one lane per candidate (verify_kernel); long needles: one lane per 32-row block
//   (verify_wave_kernel).  Duplicates (several seeds of one occurrence) are removed with an atomicCAS hash set keyed
//   by (needle, end).
//
// Exactness does not depend on the text being random: every true hit has a surviving seed, every candidate is
// verified by the full recurrence.  Pathological inputs only cost time; if the candidate buffer overflows the
// host re-runs the scan with the brute-force engine.
#pragma once

#include "brute.hpp"
#include "common.hpp"
#include "synth.hpp"

namespace spm_hip
{

constexpr uint32_t kKeyMax = 16; // symbols per key: 16 whenever the seeds allow it, down to kKeyMin for short seeds
constexpr uint32_t kKeyMin = 12;

// Seeds of one needle: n pieces of q symbols at offsets j*q.  k+1 pieces guarantee one intact piece per occurrence;
// needles with many errors get k+2 (two intact pieces on nearby diagonals), which lets the verification stage count seed
// hits per diagonal band and skip bands with a single one (candidate merging below).
constexpr uint32_t kMergeMinK = 8;
struct seed_plan
{
    uint32_t n, q;
};
__host__ __device__ inline seed_plan plan_seeds(uint32_t m, uint32_t k)
{
    const uint32_t surplus = (k >= kMergeMinK && k <= 1000 && m / (k + 2) >= kKeyMin) ? 2u : 1u;
    return {k + surplus, m / (k + surplus)};
}
constexpr uint32_t kHtEmpty = 0xFFFFFFFFu;

struct candidate
{
    uint64_t t;   // text index of the window start            | merged: first diagonal of the band (int64)
    uint32_t val; // pattern << 11 | offset of the window inside the pattern   | merged: pattern << 11 | band width - 1
    uint32_t pad; // diagonal range r: the key also sits at offsets up to offset + r of this needle (periodic seeds:
                  // one entry, one candidate, verified over the diagonals t - offset - r .. t - offset)
                  //                                            | merged: kCandMerged | (segment index + 1)
};
constexpr uint32_t kCandMerged = 0x80000000u;

// Candidate slots are handed to the waves in chunks of 16: one atomic on the shared counter per chunk instead of one
// per survivor (a single address takes ~100 atomics/us; 14-symbol keys on a 1.5 GiB text produce 10^5 survivors, which
// cost 0.8 of the kernel's 1.2 ms before).  Slots a wave reserved but did not fill are marked invalid.
constexpr uint32_t kCandChunk = 32;
constexpr uint32_t kCandInvalid = 0xFFFFFFFFu; // candidate.val of an unused slot (pattern index 2^21 - 1 never exists)
// The chunk a wave is filling lives in LDS behind the level-1 table (kCandRec words per wave: {base lo, base hi, used,
// size, candidates of the current span, span given up, span begin lo, hi}; touched only on the rare survivor path and
// once per span, so nothing stays live across the streaming loop): slots [base + used, base + size) are free.
// size = kCandChunk, or the number of survivors of one ballot if that is larger.
//
// Span budget: a span (the unit of work a wave dequeues) that produces more than `span_budget` candidates -- or meets a
// full candidate buffer -- gives up: it emits nothing more, its text range goes to the overflow list, and the host
// re-scans exactly those ranges with the brute-force kernel (hits deduplicated through the same `seen` set).  Repeat-rich
// megabases then cost their own brute-force time, not a re-run of the whole scan.
constexpr uint32_t kCandLdsSlot = 4; // words after lds_words where the per-wave chunk records start
constexpr uint32_t kCandRec = 8;     // words per wave

struct filter_params
{
    const uint8_t *text;
    uint64_t text_alloc;
    uint64_t lo, hi;          // examine windows fully inside [lo, hi)
    uint32_t stride;          // S in {1,2,4,8,16}
    uint32_t bitmap_words;    // power of two, <= 32768 (128 KiB)
    uint32_t n_probes;        // Bloom probes per key
    uint32_t span_chunks;     // chunks per span
    uint32_t span_unit;       // symbols per chunk: 1024 (1-byte text) or 4096 (2-bit shadow)
    uint32_t dynamic;         // 1: waves draw spans from counters[4] instead of a static round-robin
    uint32_t queue_cap;       // 0: resolve survivors on the spot; else they are queued in LDS (strides 1 and 2)
    uint32_t key_len;         // H: symbols per key (12..16); windows are H symbols, keys 2H bits
    uint32_t key_mask;        // (1 << 2H) - 1
    uint32_t hash_variant;    // 0/1: Bloom cascade with mul / xor-shift hashes, 2: perfect-hash fingerprints
    uint32_t lds_words;       // size of the LDS image (bitmap, or fingerprint table + displacement table)
    uint32_t chd_slot_mask;   // fingerprint slots - 1
    uint32_t chd_bucket_shift; // bucket = x >> shift
    uint32_t chd_disp_off;    // byte offset of the displacement table inside the LDS image
    const uint32_t *bitmap;   // [bitmap_words]
    const uint2 *ht;          // exact table: (key, val), val == kHtEmpty marks an empty slot
    const uint16_t *ht_rng;   // per slot: diagonal range of the entry (candidate::pad)
    uint32_t ht_mask;
    uint32_t span_budget;     // candidates one span may produce before it gives up
    candidate *cand;
    unsigned long long *counters; // [1] = candidate slots drawn, [6] = spans that gave up, [2] = hard overflow
    uint64_t cand_cap;
    uint64_t *ovf_spans;      // [ovf_cap][2]: {first text index, symbols} of every span that gave up
    uint64_t ovf_cap;
};

template <int HV>
__host__ __device__ inline uint32_t bloom_hash(uint32_t key, uint32_t i)
{
    // i-th probe index (before masking to the bitmap size).
    if (HV != 1) {
        // multiplicative hashing, distinct odd constants (v_mul_lo_u32 is quarter rate on CDNA)
        const uint32_t c[4] = {0x9E3779B1u, 0x85EBCA6Bu, 0xC2B2AE35u, 0x27D4EB2Fu};
        uint32_t x = key ^ (key >> (15 + i));
        return (x * c[i & 3]) >> 7;
    } else {
        // xor-shift + rotate: 3 full-rate VALU.  Index bits are GF(2)-linear in the key; on the text side the
        // keys are (near) uniform 16-mers, so linearity costs nothing there.
        const uint32_t sh[4] = {14, 11, 17, 9};
        const uint32_t ro[4] = {0, 7, 13, 19};
        const uint32_t x = key ^ (key >> sh[i & 3]);
        return ro[i & 3] ? ((x >> ro[i & 3]) | (x << (32 - ro[i & 3]))) : x;
    }
}

// ---- perfect-hash fingerprint table (hash-and-displace) ---------------------------------------------------------
// The key set is static, so level 1 can be (almost) exact instead of probabilistic: every key k gets the slot
//     slot(k) = (s1(k) + D[bucket(k)] * s2(k)) & slot_mask
// where the displacement D[b] is chosen on the host, bucket by bucket, so that no two keys share a slot; the slot
// holds a 16-bit fingerprint of its key.  A text window is a candidate iff the fingerprint at its slot matches:
// two LDS reads, no cascade, false-positive rate = load * 2^-16 (< 1e-5), so the exact key table in L2 is
// consulted practically only for real seed matches.
struct chd_hashes
{
    uint32_t x;  // key * C: bucket = x >> shift, s1 = x >> 3
    uint32_t s2; // per-key stride of the displacement (odd)
    uint32_t f;  // 16-bit fingerprint
};

__host__ __device__ inline chd_hashes chd_hash(uint32_t key)
{
    chd_hashes h;
    h.x = key * 0x9E3779B1u; // (two 24-bit multiplies instead of this quarter-rate one measured 8 % SLOWER on C4: the
                             // weaker mixing costs more in LDS bank conflicts than the multiply saves)
    h.s2 = (key | 1u) & 0xFFFFFFu;
    h.f = (key ^ (key >> 16)) & 0xFFFFu;
    return h;
}

__host__ __device__ inline uint32_t chd_slot(const chd_hashes &h, uint32_t d, uint32_t slot_mask)
{
    return ((h.x >> 3) + d * h.s2) & slot_mask;
}

__host__ __device__ inline uint32_t ht_hash(uint32_t key)
{
    uint32_t x = key ^ (key >> 16);
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}

// text bytes 16*lane .. 16*lane+15 of a chunk -> 32-bit word, base i at bits 2i..2i+1
__device__ __forceinline__ uint32_t pack16(const uint4 v)
{
    const uint32_t W = 0x40100401u; // byte weights 1,4,16,64
    const uint32_t p0 = __builtin_amdgcn_udot4(v.x, W, 0u, false);
    const uint32_t p1 = __builtin_amdgcn_udot4(v.y, W, 0u, false);
    const uint32_t p2 = __builtin_amdgcn_udot4(v.z, W, 0u, false);
    const uint32_t p3 = __builtin_amdgcn_udot4(v.w, W, 0u, false);
    return p0 | (p1 << 8) | (p2 << 16) | (p3 << 24);
}

// dna5 haystacks (seqan3 ranks A0 C1 G2 N3 T4): same 2-bit word with T folded onto 3, plus a 16-bit mask of the N
// positions; a window that contains an N cannot equal any (N-free) key and is dropped.  ~7 VALU per dword.
__device__ __forceinline__ uint32_t pack16_dna5(const uint4 v, uint32_t &nmask)
{
    const uint32_t W = 0x40100401u, WN = 0x08040201u;
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
    uint32_t code = 0;
    nmask = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t isT = (x[i] >> 2) & 0x01010101u;            // byte == 4
        const uint32_t isN = x[i] & (x[i] >> 1) & 0x01010101u;     // byte == 3
        code |= __builtin_amdgcn_udot4(x[i] - isT, W, 0u, false) << (8 * i);
        nmask |= __builtin_amdgcn_udot4(isN, WN, 0u, false) << (4 * i);
    }
    return code;
}

__device__ __forceinline__ uint4 load_text16(const uint8_t *text, uint64_t idx, uint64_t limit)
{
    // idx % 16 == 0.  Bytes at or beyond `limit` read as 0.
    if (idx + 16 <= limit)
        return *reinterpret_cast<const uint4 *>(text + idx);
    uint32_t w[4] = {0, 0, 0, 0};
    for (int b = 0; b < 16; ++b)
        if (idx + b < limit)
            w[b >> 2] |= (uint32_t)text[idx + b] << (8 * (b & 3));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

template <bool NT>
__device__ __forceinline__ uint4 load16_stream(const uint8_t *p)
{
    if (NT) {
        // streamed once: nontemporal keeps it from displacing the key table in L2 and measures +13 % on a pure
        // 16 GiB read (tools/hbm_read_probe: 7.0 vs 6.2 TB/s)
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
        return make_uint4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1),
                          __builtin_nontemporal_load(q + 2), __builtin_nontemporal_load(q + 3));
    }
    return *reinterpret_cast<const uint4 *>(p);
}

// one atomic per wave: add the lanes' counts to a statistics counter (call with the wave converged)
__device__ __forceinline__ void wave_count_add(unsigned long long *counter, uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v)
        atomicAdd(counter, (unsigned long long)v);
}

// this wave's chunk record in LDS
__device__ __forceinline__ uint32_t *cand_chunk_of(const filter_params &P, const uint32_t *lds)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    return const_cast<uint32_t *>(lds) + P.lds_words + kCandLdsSlot + kCandRec * wave;
}

// A wave starts a span: fresh budget (wave-uniform call).  `len` = symbols the span covers.
__device__ __forceinline__ void span_open(const filter_params &P, const uint32_t *lds, uint32_t lane, uint64_t begin)
{
    if (lane == 0) {
        uint32_t *ck = cand_chunk_of(P, lds);
        ck[4] = 0;
        ck[5] = 0;
        ck[6] = (uint32_t)begin;
        ck[7] = (uint32_t)(begin >> 32);
    }
}

// the current span gives up (wave-uniform call)
__device__ __forceinline__ void span_give_up(const filter_params &P, uint32_t *ck, uint32_t lane, uint64_t span_symbols)
{
    if (lane == 0) {
        ck[5] = 1;
        const unsigned long long i = atomicAdd(&P.counters[6], 1ull);
        if (i < P.ovf_cap) {
            P.ovf_spans[2 * i] = ((uint64_t)ck[7] << 32) | ck[6];
            P.ovf_spans[2 * i + 1] = span_symbols;
        } else {
            atomicAdd(&P.counters[2], 1ull); // no room to remember it: the host re-runs the whole scan
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// mark the unused tail of a wave's chunk invalid (wave-uniform call)
__device__ __forceinline__ void cand_close(const filter_params &P, const uint32_t *ck, uint32_t lane)
{
    const uint32_t used = ck[2], size = ck[3];
    const uint64_t idx = (((uint64_t)ck[1] << 32) | ck[0]) + used + lane;
    if (used + lane < size && idx < P.cand_cap) {
        candidate c;
        c.t = 0;
        c.val = kCandInvalid;
        c.pad = 0;
        P.cand[idx] = c;
    }
}

// Level 2 for one survivor per lane (wave-uniform call; `probing` marks the lanes that hold one): probe the exact key
// table, append every (needle, offset) the key belongs to as a candidate.  Candidate slots come in chunks for every
// stride: one atomic on the shared counter per kCandChunk candidates (a single address takes ~100 atomics/us, and a
// repeat-rich text produces 10^6..10^7 candidates).
template <int S>
__device__ __forceinline__ void resolve_survivors(const filter_params &P, bool probing, uint32_t key, uint64_t t,
                                                  uint32_t lane, const uint32_t *lds)
{
    uint32_t *ck = cand_chunk_of(P, lds);
    if (__builtin_amdgcn_readfirstlane(ck[5]) != 0)
        return; // this span has given up: the brute-force kernel will scan it
    bool emit = false;
    uint32_t val = 0, rng = 0;
    uint32_t slot = ht_hash(key) & P.ht_mask;
    while (__ballot(probing) != 0) {
        emit = false;
        if (probing) {
            const uint2 e = P.ht[slot];
            if (e.y == kHtEmpty) {
                probing = false;
            } else {
                if (e.x == key) {
                    emit = true;
                    val = e.y;
                    rng = P.ht_rng[slot];
                }
                slot = (slot + 1) & P.ht_mask;
            }
        }
        const uint64_t m = __ballot(emit);
        if (m != 0) {
            const uint32_t n = __popcll(m);
            uint32_t used = (uint32_t)__builtin_amdgcn_readfirstlane(ck[2]);
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(ck[4]) + n;
            const uint64_t span_symbols = (uint64_t)P.span_chunks * (uint64_t)(P.span_unit);
            if (cnt > P.span_budget) {
                span_give_up(P, ck, lane, span_symbols);
                return;
            }
            if (used + n > (uint32_t)__builtin_amdgcn_readfirstlane(ck[3])) { // wave-uniform: close, draw the next
                cand_close(P, ck, lane);
                const uint32_t size = n > kCandChunk ? n : kCandChunk;
                if (lane == 0) {
                    const unsigned long long b = atomicAdd(&P.counters[1], (unsigned long long)size);
                    ck[0] = (uint32_t)b;
                    ck[1] = (uint32_t)(b >> 32);
                    ck[2] = 0;
                    ck[3] = size;
                }
                used = 0;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const uint64_t cbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(ck[1]) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane(ck[0]);
            if (cbase + used + n > P.cand_cap) { // the candidate buffer is full
                span_give_up(P, ck, lane, span_symbols);
                return;
            }
            if (emit) {
                const uint64_t idx = cbase + used + __popcll(m & ((1ull << lane) - 1));
                candidate c;
                c.t = t;
                c.val = val;
                c.pad = rng;
                P.cand[idx] = c;
            }
            if (lane == 0) {
                ck[2] = used + n;
                ck[4] = cnt;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Strides 1 and 2 meet 10^5..10^6 survivors per scan.  Resolving each one where it turns up costs the whole wave an L2
// round trip per survivor (~3 us); instead they are queued in LDS (key + text position) and resolved 64 at a time.
// Per wave: {count, key[cap], t_lo[cap], t_hi[cap]} behind the chunk records.
constexpr uint32_t kQueueCap = 96;
constexpr uint32_t kQueueWords = 4 + 3 * kQueueCap; // per wave
constexpr uint32_t kQueueLdsSlot = kCandLdsSlot + kCandRec * 16; // words after lds_words where the queues start

__device__ __forceinline__ uint32_t *survivor_queue_of(const filter_params &P, const uint32_t *lds)
{
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    return const_cast<uint32_t *>(lds) + P.lds_words + kQueueLdsSlot + kQueueWords * wave;
}

template <int S>
__device__ __forceinline__ void drain_survivors(const filter_params &P, uint32_t lane, const uint32_t *lds)
{
    uint32_t *q = survivor_queue_of(P, lds);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(q[0]);
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t e = base + lane;
        const bool have = e < n;
        const uint32_t key = have ? q[4 + e] : 0u;
        const uint64_t t = have ? (((uint64_t)q[4 + 2 * kQueueCap + e] << 32) | q[4 + kQueueCap + e]) : 0ull;
        resolve_survivors<S>(P, have, key, t, lane, lds);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0)
        q[0] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Level 1 + level 2 on NWD 2-bit-packed words per lane (w[j] = 16 bases, prev[j] = the 16 bases before them).
// PK selects how word j maps to a text position (1-byte text vs. packed shadow).
// KM: keys shorter than 16 symbols (masked); only strides 1 and 2 ever carry such keys.
template <int S, int NWD, int HV, int SIG, bool PK, bool KM>
__device__ __forceinline__ void filter_words(const filter_params &P, const uint32_t (&w)[NWD],
                                             const uint32_t (&prev)[NWD], const uint32_t (&nv)[SIG == 5 ? NWD : 1],
                                             uint64_t gbase, uint32_t lane, const uint32_t *lds, uint32_t idx_mask)
{
    constexpr int NWIN = 16 / S; // windows per word
    const uint32_t kmask = KM ? P.key_mask : 0xFFFFFFFFu;
    static_assert(NWD * NWIN <= 32, "one mask bit per window of a group");
    // windows d = S, 2S, .., 16 of chunk u: text start t = L_u - 16 + d, key = bits [2d, 2d+32) of (w:prev)
    uint32_t pos_mask = 0;
    if constexpr (HV == 2) {
        // perfect-hash fingerprints: round 1 reads the displacement of every window's bucket, round 2 the
        // fingerprint at the displaced slot -- two LDS round trips per group, all windows in flight together
        const uint16_t *fp_tab = reinterpret_cast<const uint16_t *>(lds);
        const uint16_t *disp_tab = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(lds) + P.chd_disp_off);
        uint32_t xs[NWD * NWIN], ds[NWD * NWIN];
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
#pragma unroll
            for (int i = 0; i < NWIN; ++i) {
                const int d = S * (i + 1);
                const uint32_t key = (d == 16 ? w[u] : alignbit(w[u], prev[u], (2 * d) & 31)) & kmask;
                const uint32_t x = chd_hash(key).x;
                xs[u * NWIN + i] = x;
                ds[u * NWIN + i] = disp_tab[x >> P.chd_bucket_shift];
            }
        }
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
#pragma unroll
            for (int i = 0; i < NWIN; ++i) {
                const int d = S * (i + 1);
                const uint32_t key = (d == 16 ? w[u] : alignbit(w[u], prev[u], (2 * d) & 31)) & kmask;
                const uint32_t slot = ((xs[u * NWIN + i] >> 3) + ds[u * NWIN + i] * ((key | 1u) & 0xFFFFFFu)) &
                                      P.chd_slot_mask;
                ds[u * NWIN + i] = fp_tab[slot];
            }
        }
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
#pragma unroll
            for (int i = 0; i < NWIN; ++i) {
                const int d = S * (i + 1);
                const uint32_t key = (d == 16 ? w[u] : alignbit(w[u], prev[u], (2 * d) & 31)) & kmask;
                const uint32_t miss = (ds[u * NWIN + i] ^ key ^ (key >> 16)) & 0xFFFFu;
                pos_mask |= (miss == 0 ? 1u : 0u) << (u * NWIN + i);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < NWD; ++u) {
#pragma unroll
            for (int i = 0; i < NWIN; ++i) {
                const int d = S * (i + 1);
                const uint32_t key = (d == 16 ? w[u] : alignbit(w[u], prev[u], (2 * d) & 31)) & kmask;
                const uint32_t h = bloom_hash<HV>(key, 0) & idx_mask;
                const uint32_t word = lds[h >> 5];
                pos_mask |= __builtin_amdgcn_ubfe(word, h, 1) << (u * NWIN + i);
            }
        }
        // cascade: further probes only for survivors
        for (uint32_t pr = 1; pr < P.n_probes; ++pr) {
            if (__ballot(pos_mask != 0) == 0)
                break;
            uint32_t keep = 0;
#pragma unroll
            for (int u = 0; u < NWD; ++u) {
#pragma unroll
                for (int i = 0; i < NWIN; ++i) {
                    if (pos_mask & (1u << (u * NWIN + i))) {
                        const int d = S * (i + 1);
                        const uint32_t key = (d == 16 ? w[u] : alignbit(w[u], prev[u], (2 * d) & 31)) & kmask;
                        const uint32_t h = bloom_hash<HV>(key, pr) & idx_mask;
                        const uint32_t word = lds[h >> 5];
                        keep |= __builtin_amdgcn_ubfe(word, h, 1) << (u * NWIN + i);
                    }
                }
            }
            pos_mask = keep;
        }
    }
    if constexpr (SIG == 5) {
        if (__ballot(pos_mask != 0) != 0) {
#pragma unroll
            for (int u = 0; u < NWD; ++u) {
#pragma unroll
                for (int i = 0; i < NWIN; ++i) {
                    const int d = S * (i + 1);
                    if (__builtin_amdgcn_ubfe(nv[u], d, P.key_len) != 0) // an N inside the window
                        pos_mask &= ~(1u << (u * NWIN + i));
                }
            }
        }
    }
    // survivors: exact key table
    while (__ballot(pos_mask != 0) != 0) {
        if (__builtin_amdgcn_readfirstlane(cand_chunk_of(P, lds)[5]) != 0)
            break; // the span has given up
        uint64_t t = 0;
        uint32_t key = 0;
        bool probing = false;
        if (pos_mask != 0) {
            const int bit = __ffs(pos_mask) - 1;
            pos_mask &= pos_mask - 1;
            const int u = bit / NWIN, i = bit % NWIN;
            const int d = S * (i + 1);
            uint32_t wu = w[0], pu = prev[0];
#pragma unroll
            for (int q = 1; q < NWD; ++q)
                if (u == q) {
                    wu = w[q];
                    pu = prev[q];
                }
            key = (d == 16 ? wu : alignbit(wu, pu, (uint32_t)(2 * d) & 31u)) & kmask;
            // first base of word u of this lane: 1-byte text = chunk u, 16 bytes per lane; packed shadow = load u>>2
            // (4096 bases), 64 bases per lane, word u&3
            const uint64_t wpos = PK ? gbase + (uint64_t)(u >> 2) * 4096 + (uint64_t)lane * 64 + (uint64_t)(u & 3) * 16
                                     : gbase + (uint64_t)u * 1024 + (uint64_t)lane * 16;
            const int64_t ts = (int64_t)wpos - 16 + d;
            if (ts >= (int64_t)P.lo && (uint64_t)ts + P.key_len <= P.hi) {
                t = (uint64_t)ts;
                probing = true;
            }
        }
        if (S <= 2 && P.queue_cap != 0) {
            // queue them; the wave resolves 64 at a time
            const uint64_t m = __ballot(probing);
            if (m != 0) {
                uint32_t *q = survivor_queue_of(P, lds);
                uint32_t qn = (uint32_t)__builtin_amdgcn_readfirstlane(q[0]);
                const uint32_t n = __popcll(m);
                if (qn + n > kQueueCap) {
                    drain_survivors<S>(P, lane, lds);
                    qn = 0;
                }
                if (probing) {
                    const uint32_t e = qn + __popcll(m & ((1ull << lane) - 1));
                    q[4 + e] = key;
                    q[4 + kQueueCap + e] = (uint32_t)t;
                    q[4 + 2 * kQueueCap + e] = (uint32_t)(t >> 32);
                }
                if (lane == 0)
                    q[0] = qn + n;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            resolve_survivors<S>(P, probing, key, t, lane, lds);
        }
    }
}


// One group of UU consecutive 1-KiB chunks, already in registers.  chunk u of the group starts at text index
// gbase + 1024*u; this lane holds its bytes [16*lane, 16*lane+16).
template <int S, int UU, int HV, int SIG, bool KM>
__device__ __forceinline__ void filter_group(const filter_params &P, const uint4 (&cur)[UU], uint64_t gbase,
                                             uint32_t &carry_in, uint32_t &carry_n, uint32_t lane,
                                             const uint32_t *lds, uint32_t idx_mask)
{
    uint32_t w[UU], prev[UU];
    uint32_t nv[SIG == 5 ? UU : 1]; // dna5: (this lane's N mask << 16) | previous lane's N mask
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        uint32_t nm = 0;
        w[u] = SIG == 5 ? pack16_dna5(cur[u], nm) : pack16(cur[u]);
        prev[u] = __builtin_amdgcn_update_dpp(0u, w[u], 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
        if (lane == 0)
            prev[u] = carry_in;
        carry_in = __builtin_amdgcn_readlane(w[u], 63);
        if (SIG == 5) {
            uint32_t np = __builtin_amdgcn_update_dpp(0u, nm, 0x138, 0xF, 0xF, false);
            if (lane == 0)
                np = carry_n;
            carry_n = __builtin_amdgcn_readlane(nm, 63);
            nv[u] = (nm << 16) | np;
        }
    }
    filter_words<S, UU, HV, SIG, false, KM>(P, w, prev, nv, gbase, lane, lds, idx_mask);
}

// Streaming structure: a wave owns spans of consecutive 1-KiB chunks; per iteration it works on a GROUP of U chunks
// (U KiB contiguous per wave) while the unconditional 16-byte loads of the next group are already in flight, so
// each wave keeps 2*U KiB outstanding.  Only groups that lie fully inside the text take this path; the ragged end
// of the text goes through a guarded one-chunk loop (bytes past the end read as 0).
// 512 threads: 8 waves per CU is the measured best for the HBM-bound 1-byte text, and a bound of 1024 leaves 128 VGPRs,
// which made the masked-key stride-2 variant spill to scratch inside the streaming loop (+31 % kernel time).
template <int S, int U, bool NT, int HV, int SIG, bool KM>
__global__ __launch_bounds__(512) void seed_filter_kernel(const filter_params P)
{
    extern __shared__ uint32_t lds[];
    // ---- stage the level-1 table in LDS (once per workgroup; the grid is persistent) ----
    for (uint32_t i = threadIdx.x; i < P.lds_words; i += blockDim.x)
        lds[i] = P.bitmap[i];
    if ((threadIdx.x & 63) == 0) { // this wave's candidate chunk: none drawn yet
        uint32_t *ck = lds + P.lds_words + kCandLdsSlot + kCandRec * (threadIdx.x >> 6);
        for (uint32_t i = 0; i < kCandRec; ++i)
            ck[i] = 0; // size 0: the first survivor draws a chunk; no span open yet
        if (P.queue_cap != 0)
            lds[P.lds_words + kQueueLdsSlot + kQueueWords * (threadIdx.x >> 6)] = 0; // empty survivor queue
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint64_t wave_id =
        (uint64_t)blockIdx.x * waves_per_wg + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint32_t idx_mask = P.bitmap_words * 32 - 1;

    const uint64_t base0 = P.lo & ~1023ull; // chunks are 1 KiB aligned relative to text[0]
    const uint64_t n_chunks = (P.hi - base0 + 1023) / 1024;
    const uint64_t n_whole = (P.hi - base0) / 1024; // chunks that lie fully inside the text
    const uint64_t span = P.span_chunks;            // multiple of U (host guarantees)
    const uint64_t n_spans = (n_chunks + span - 1) / span;
    const uint8_t *lane_text = P.text + base0 + (uint64_t)lane * 16;

    // Dynamic scheduling (evens out the tail; static round-robin measured 15-20 % slower on 16 GiB):
    //   dynamic == 1: every wave draws its next span with one returning atomic on counters[4];
    //   dynamic == 2: one atomic per WORKGROUP hands one span to each of its waves (waves_per_wg times fewer
    //                 atomics at the same granularity; a single head saturates near 88 dequeues/us, which a 1 GiB
    //                 text with 16 KiB spans would exceed).  Sharding the head per XCD group was slower (no
    //                 balancing across shards).
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint64_t sp = wave_id;
    for (;;) {
        if (P.dynamic == 1) {
            unsigned long long t = 0;
            if (lane == 0)
                t = atomicAdd(&P.counters[4], 1ull);
            sp = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)t);
        } else if (P.dynamic == 2) {
            __syncthreads(); // every wave has read the previous base
            if (threadIdx.x == 0) {
                const unsigned long long t = atomicAdd(&P.counters[4], (unsigned long long)waves_per_wg);
                lds[P.lds_words] = (uint32_t)t;
                lds[P.lds_words + 1] = (uint32_t)(t >> 32);
            }
            __syncthreads();
            const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lds[P.lds_words + 1]) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane(lds[P.lds_words]);
            if (base >= n_spans)
                break; // uniform over the workgroup
            sp = base + wave_in_wg;
            if (sp >= n_spans)
                continue; // this wave idles for the last round but keeps meeting the barriers
        }
        if (sp >= n_spans)
            break;
        const uint64_t c_begin = sp * span;
        const uint64_t c_end = c_begin + span < n_chunks ? c_begin + span : n_chunks;
        span_open(P, lds, lane, base0 + c_begin * 1024);
        // word of the lane "before lane 0": last 16 bytes of the previous chunk
        uint32_t carry_in = 0, carry_n = 0;
        {
            const uint64_t cb = base0 + c_begin * 1024;
            if (cb >= 16 && lane == 0) {
                const uint4 before = load_text16(P.text, cb - 16, P.hi);
                carry_in = SIG == 5 ? pack16_dna5(before, carry_n) : pack16(before);
            }
            carry_in = __builtin_amdgcn_readfirstlane(carry_in);
            carry_n = __builtin_amdgcn_readfirstlane(carry_n);
        }
        // ---- fast path: whole groups ----
        uint64_t ch = c_begin;
        const uint64_t whole_end = c_end < n_whole ? c_end : n_whole;
        const uint64_t fast_end = c_begin + (whole_end > c_begin ? (whole_end - c_begin) / U * U : 0);
        if (ch < fast_end) {
            uint4 nxt[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                nxt[u] = load16_stream<NT>(lane_text + (ch + u) * 1024);
            for (; ch < fast_end; ch += U) {
                uint4 cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    cur[u] = nxt[u];
                // prefetch unconditionally (no branch around the loads); past the last group of the span re-read
                // one chunk of the current group -- in bounds, L2-resident, 1/U of a group
                const bool more = ch + U < fast_end;
                const uint64_t pf = more ? ch + U : ch;
                const uint64_t ustride = more ? 1024 : 0;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    nxt[u] = load16_stream<NT>(lane_text + pf * 1024 + (uint64_t)u * ustride);
                // strides 1 and 2: keep the loads here -- left alone, the scheduler sinks them to the very end of the
                // group's work and the next iteration waits out the whole HBM latency.  (The larger strides are
                // scheduled well as they are; a barrier there only forces the packing to wait for all eight loads.)
                if constexpr (S <= 2)
                    __builtin_amdgcn_sched_barrier(0);
                filter_group<S, U, HV, SIG, KM>(P, cur, base0 + ch * 1024, carry_in, carry_n, lane, lds, idx_mask);
            }
        }
        // ---- ragged end ----
        for (; ch < c_end; ++ch) {
            uint4 one[1];
            one[0] = load_text16(P.text, base0 + ch * 1024 + (uint64_t)lane * 16, P.hi);
            filter_group<S, 1, HV, SIG, KM>(P, one, base0 + ch * 1024, carry_in, carry_n, lane, lds, idx_mask);
        }
        if constexpr (S <= 2) { // queued survivors are charged to the span they came from
            if (P.queue_cap != 0)
                drain_survivors<S>(P, lane, lds);
        }
        sp += n_waves;
    }
    cand_close(P, cand_chunk_of(P, lds), lane);
}

// ---------------------------------------------------------------------------------------------------
// Optional 2-bit shadow of a dna4 haystack (spm_hip_text_pack): 16 symbols per uint32, same bit order as pack16.
// A text that is scanned many times (one reference, many needle batches) is then streamed at a quarter of the
// HBM traffic; hits are identical.  text_pack_kernel builds it in one pass and flags symbols outside {0..3}.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_pack_kernel(const uint8_t *__restrict__ text, uint64_t n,
                                                        uint32_t *__restrict__ packed, uint64_t n_words,
                                                        unsigned int *bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned int any_bad = 0;
    for (uint64_t wi = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < n_words; wi += stride) {
        const uint4 v = load_text16(text, wi * 16, n);
        any_bad |= (v.x | v.y | v.z | v.w) & 0xFCFCFCFCu;
        packed[wi] = pack16(v);
    }
    if (any_bad)
        atomicOr(bad, 1u);
}

// Same filter, fed from the shadow: one 16-byte load per lane = 4 words = 64 symbols; a wave-load ("p-chunk") covers
// 4096 symbols.  U2 p-chunks per group, the next group's loads in flight.  The shadow is zero-padded to whole
// p-chunks, so every load is unconditional; windows reaching past the text are dropped by the range check.
template <int S, int U2, int HV, bool KM>
__global__ __launch_bounds__(1024) void seed_filter_packed_kernel(const filter_params P, const uint4 *__restrict__ shadow)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < P.lds_words; i += blockDim.x)
        lds[i] = P.bitmap[i];
    if ((threadIdx.x & 63) == 0) { // this wave's candidate chunk: none drawn yet
        uint32_t *ck = lds + P.lds_words + kCandLdsSlot + kCandRec * (threadIdx.x >> 6);
        for (uint32_t i = 0; i < kCandRec; ++i)
            ck[i] = 0; // size 0: the first survivor draws a chunk; no span open yet
        if (P.queue_cap != 0)
            lds[P.lds_words + kQueueLdsSlot + kQueueWords * (threadIdx.x >> 6)] = 0; // empty survivor queue
    }
    __syncthreads();

    constexpr int NWD = 4 * U2;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;
    const uint32_t idx_mask = P.bitmap_words * 32 - 1;

    const uint64_t base0 = P.lo & ~4095ull;                   // p-chunks are aligned to 4096 symbols
    const uint64_t n_chunks = (P.hi - base0 + 4095) / 4096;
    const uint64_t span = P.span_chunks;                      // multiple of U2
    const uint64_t n_spans = (n_chunks + span - 1) / span;
    const uint4 *lane_src = shadow + (base0 / 64) + lane;     // one uint4 = 64 symbols

    uint64_t sp = wave_id;
    for (;;) {
        if (P.dynamic == 1) {
            unsigned long long t = 0;
            if (lane == 0)
                t = atomicAdd(&P.counters[4], 1ull);
            sp = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(t >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)t);
        } else if (P.dynamic == 2) {
            __syncthreads();
            if (threadIdx.x == 0) {
                const unsigned long long t = atomicAdd(&P.counters[4], (unsigned long long)waves_per_wg);
                lds[P.lds_words] = (uint32_t)t;
                lds[P.lds_words + 1] = (uint32_t)(t >> 32);
            }
            __syncthreads();
            const uint64_t base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(lds[P.lds_words + 1]) << 32) |
                                  (uint32_t)__builtin_amdgcn_readfirstlane(lds[P.lds_words]);
            if (base >= n_spans)
                break;
            sp = base + wave_in_wg;
            if (sp >= n_spans)
                continue;
        }
        if (sp >= n_spans)
            break;
        const uint64_t c_begin = sp * span;
        const uint64_t c_end = c_begin + span < n_chunks ? c_begin + span : n_chunks;
        const uint64_t fast_end = c_begin + (c_end - c_begin + U2 - 1) / U2 * U2; // whole groups (shadow is padded)
        span_open(P, lds, lane, base0 + c_begin * 4096);
        // the word in front of this span's first word
        uint32_t carry_in = 0;
        {
            const uint64_t first_word = (base0 + c_begin * 4096) / 16;
            if (first_word > 0 && lane == 0)
                carry_in = reinterpret_cast<const uint32_t *>(shadow)[first_word - 1];
            carry_in = __builtin_amdgcn_readfirstlane(carry_in);
        }
        uint4 nxt[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u)
            nxt[u] = load16_stream<true>(reinterpret_cast<const uint8_t *>(lane_src + (c_begin + u) * 64));
        for (uint64_t ch = c_begin; ch < fast_end; ch += U2) {
            uint4 cur[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u)
                cur[u] = nxt[u];
            const bool more = ch + U2 < fast_end;
            const uint64_t pf = more ? ch + U2 : ch;
            const uint64_t ustride = more ? 64 : 0;
#pragma unroll
            for (int u = 0; u < U2; ++u)
                nxt[u] = load16_stream<true>(reinterpret_cast<const uint8_t *>(lane_src + pf * 64 + (uint64_t)u * ustride));
            __builtin_amdgcn_sched_barrier(0); // as in seed_filter_kernel: the prefetch stays ahead of the group's work
            uint32_t w[NWD], prev[NWD];
            const uint32_t nv[1] = {0};
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                w[4 * u + 0] = cur[u].x;
                w[4 * u + 1] = cur[u].y;
                w[4 * u + 2] = cur[u].z;
                w[4 * u + 3] = cur[u].w;
                uint32_t p0 = __builtin_amdgcn_update_dpp(0u, cur[u].w, 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
                if (lane == 0)
                    p0 = carry_in;
                carry_in = __builtin_amdgcn_readlane(cur[u].w, 63);
                prev[4 * u + 0] = p0;
                prev[4 * u + 1] = cur[u].x;
                prev[4 * u + 2] = cur[u].y;
                prev[4 * u + 3] = cur[u].z;
            }
            filter_words<S, NWD, HV, 4, true, KM>(P, w, prev, nv, base0 + ch * 4096, lane, lds, idx_mask);
        }
        if constexpr (S <= 2) {
            if (P.queue_cap != 0)
                drain_survivors<S>(P, lane, lds);
        }
        sp += n_waves;
    }
    cand_close(P, cand_chunk_of(P, lds), lane);
}

// ---------------------------------------------------------------------------------------------------
// verification: one lane per candidate
// ---------------------------------------------------------------------------------------------------
struct verify_params
{
    const uint8_t *text;
    uint64_t text_alloc; // readable bytes from text
    uint64_t ctx_begin;  // first symbol of the haystack that may be consumed
    uint64_t scan_begin; // owned: last symbol index in [scan_begin, scan_end)
    uint64_t scan_end;
    uint64_t pos_offset;
    const candidate *cand;
    const unsigned long long *counters; // [1] candidates
    uint64_t cand_cap;
    const uint32_t *peq32; // the brute table: [group][sigma+1][nw_table][64], needles top-aligned
    uint32_t sigma;        // alphabet size; LDS holds sigma+1 rows per thread (row sigma = no match)
    uint32_t nw_table;     // words per needle in that table
    uint32_t max_k;        // largest k of the set: 2*max_k + 1 end-position slots per candidate
    uint32_t key_len;      // symbols per key (a key window must lie inside one segment)
    const int32_t *m;      // per pattern
    const int32_t *k;
    uint32_t report_begin; // 1: exact matchers report begin = end - m
    uint32_t max_span;     // merged candidates: extra end positions a diagonal band answers for (0: none merged)
    uint32_t cand_counter; // index of the candidate count in `counters` (1: raw candidates, 3: merged bands)
    uint32_t wave_text;    // wave-per-candidate kernel: bytes of LDS per group for the candidate's text window
    const uint8_t *needle_ranks;    // whole-seed check: the needles' symbols back to back (nullptr: check disabled)
    const uint32_t *needle_offsets; // start of every needle in needle_ranks
    uint32_t text_sigma;
    unsigned long long *seen; // hash set of (pattern << 40 | end)
    uint32_t seen_mask;
    spm_hit *hits;
    unsigned long long *hit_counter; // counters[0]
    uint64_t hit_cap;
    unsigned long long *overflow; // counters[2]
    const uint64_t *seg_offsets;  // segmented haystacks: n_segments+1 ascending offsets, or nullptr
    uint64_t n_segments;
    const uint32_t *seg_owned;    // optional, per segment: only hits whose last symbol lies at or behind this offset
                                  // are wanted (journaled-sequence contexts: the symbols before it are left context)
};

// A candidate says: the key window at needle offset x matches the text at t.  The pigeonhole argument needs a seed that
// occurs in the text UNCHANGED, and such a seed yields a candidate of its own -- so a candidate whose whole seed does
// not match exactly can be dropped without losing an occurrence.  On long texts most candidates are chance matches of
// the 16-symbol key (C4: 800 000 of 1 140 000; C3: 16 000 of 20 000); they fail this check after ~1.3 symbols.
__device__ __forceinline__ bool seed_intact(const verify_params &P, const candidate &c, int64_t hay_b, int64_t hay_e)
{
    if (!P.needle_ranks || c.pad != 0)
        return true; // (a candidate with a diagonal range stands for several offsets of a periodic seed: kept)
    const uint32_t pat = c.val >> 11, x = c.val & 0x7FF;
    const seed_plan sp = plan_seeds((uint32_t)P.m[pat], (uint32_t)P.k[pat]);
    const uint32_t o = (x / sp.q) * sp.q; // start of the seed inside the needle
    const int64_t ts = (int64_t)c.t - (int64_t)(x - o); // where the seed would start in the text
    if (ts < hay_b || ts + (int64_t)sp.q > hay_e)
        return false; // it would stick out of the haystack
    const uint8_t *nd = P.needle_ranks + P.needle_offsets[pat] + o;
    const uint8_t *tx = P.text + ts;
    const uint32_t w0 = x - o, w1 = w0 + P.key_len; // the key window itself already matched
    for (uint32_t i = 0; i < sp.q; ++i) {
        if (i >= w0 && i < w1)
            continue;
        if (tx[i] != nd[i])
            return false;
    }
    return true;
}

// One lane per candidate.  Same recurrence and layout as the brute kernel (32-bit words, v_bitop3, needles
// top-aligned so that the score delta is bit 31 of the top word): the candidate's needle rows are staged from the
// brute table into LDS ([symbol][word][thread], conflict-free whatever the per-lane symbol), only the NWN top words
// that can hold needle rows are processed, and the text window is read in prefetched 16-byte blocks -- the
// per-symbol dependency chain is one LDS round trip.
template <int NWN>
__global__ __launch_bounds__(256) void verify_kernel(const verify_params P)
{
    extern __shared__ uint32_t vlds[]; // [sigma + 1][NWN][blockDim.x] words, then [2*max_k + 1 + max_span][blockDim.x] uint16
    const uint32_t tid = threadIdx.x;
    const uint32_t nthr = blockDim.x;
    const uint32_t rows = P.sigma + 1;
    uint16_t *hitbuf = reinterpret_cast<uint16_t *>(vlds + (size_t)rows * NWN * nthr);
    const uint32_t n_slots = 2 * P.max_k + 1 + P.max_span;
    for (uint32_t r = 0; r < n_slots; ++r)
        hitbuf[(size_t)r * nthr + tid] = 0;
    unsigned long long n_cand = P.counters[P.cand_counter];
    if (n_cand > P.cand_cap)
        n_cand = P.cand_cap; // overflow is handled by the host (brute-force re-run)
    const uint64_t stride = (uint64_t)gridDim.x * nthr;
    uint32_t n_valid = 0;
    for (uint64_t ci = (uint64_t)blockIdx.x * nthr + tid; ci < n_cand; ci += stride) {
        const candidate c = P.cand[ci];
        if (c.val == kCandInvalid)
            continue; // a slot its wave reserved but did not fill
        ++n_valid;
        const uint32_t pat = c.val >> 11;
        const bool merged = (c.pad & kCandMerged) != 0;
        // raw candidate: diagonals t - x - r .. t - x (r = c.pad);  merged band: diagonals t .. t + span
        const int64_t x = merged ? 0 : (int64_t)(c.val & 0x7FF) + (int64_t)c.pad;
        const int64_t span = merged ? (int64_t)(c.val & 0x7FF) : (int64_t)c.pad;
        const int64_t m = P.m[pat];
        const int64_t k = P.k[pat];
        const int64_t d = (int64_t)c.t - x; // first diagonal: needle position 0 <-> text index d
        // exclusive end positions this candidate answers for: e in [d+m-k, d+span+m+k]
        int64_t e_lo = d + m - k;
        int64_t e_hi = d + m + k + span;
        // ownership: last symbol e-1 in [scan_begin, scan_end)
        int64_t own_b = (int64_t)P.scan_begin, own_e = (int64_t)P.scan_end, hay_b = (int64_t)P.ctx_begin;
        if (P.seg_offsets) {
            // every segment is a haystack of its own: find the one holding the key window, clamp to it
            uint64_t lo = 0, hi = P.n_segments; // invariant: seg_offsets[lo] <= t < seg_offsets[hi]
            if (merged) {
                lo = (c.pad & ~kCandMerged) - 1;
            } else {
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (P.seg_offsets[mid] <= c.t)
                        lo = mid;
                    else
                        hi = mid;
                }
            }
            const int64_t sb = (int64_t)P.seg_offsets[lo], se = (int64_t)P.seg_offsets[lo + 1];
            if (!merged && (int64_t)c.t + (int64_t)P.key_len > se)
                continue; // the key window straddles two haystacks
            own_b = P.seg_owned ? sb + (int64_t)P.seg_owned[lo] : sb;
            own_e = se;
            hay_b = sb;
        }
        if (!merged && !seed_intact(P, c, hay_b, own_e))
            continue;
        if (e_lo < own_b + 1)
            e_lo = own_b + 1;
        if (e_hi > own_e)
            e_hi = own_e;
        if (e_lo > e_hi)
            continue;
        // cold start m+k symbols before the first end position (or at the haystack start)
        int64_t ws = e_lo - (m + k);
        if (ws < hay_b)
            ws = hay_b;
        // stage this needle's rows: brute table [group][row][word][lane], top NWN words
        const uint32_t *src = P.peq32 + (((size_t)(pat >> 6) * rows) * P.nw_table + (P.nw_table - NWN)) * 64 + (pat & 63);
        for (uint32_t r = 0; r < rows; ++r)
#pragma unroll
            for (int w = 0; w < NWN; ++w)
                vlds[((size_t)r * NWN + w) * nthr + tid] = src[((size_t)r * P.nw_table + w) * 64];
        bool any_hit = false;
        myers_lane<NWN, false> L;
        {
            const int32_t off = NWN * 32 - (int32_t)m;
#pragma unroll
            for (int w = 0; w < NWN; ++w) {
                const int32_t lo = off - w * 32;
                L.VP[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
                L.VN[w] = 0;
            }
            L.score = (int32_t)m;
        }
        const int64_t blk0 = ws & ~15ll;
        uint4 nxt = load_text16(P.text, (uint64_t)blk0, P.text_alloc);
        for (int64_t blk = blk0; blk < e_hi; blk += 16) {
            const uint4 cur = nxt;
            if (blk + 16 < e_hi)
                nxt = load_text16(P.text, (uint64_t)(blk + 16), P.text_alloc);
            const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t p = blk + i;
                if (p < ws || p >= e_hi)
                    continue;
                uint32_t sym = (words[i >> 2] >> (8 * (i & 3))) & 0xFF;
                sym = sym < P.sigma ? sym : P.sigma;
                L.step_strided(vlds + ((size_t)sym * NWN) * nthr + tid, nthr);
                const int64_t e = p + 1;
                if (L.score <= (int32_t)k && e >= e_lo) {
                    // remember the hit; emission is deferred until the whole wave has finished scanning so that the
                    // CAS / append atomics (1-2 us each) are issued once per end-position slot for all lanes together
                    // instead of stalling the wave at ~every lane's own hit
                    hitbuf[(size_t)(e - e_lo) * nthr + tid] = (uint16_t)(L.score + 1);
                    any_hit = true;
                }
            }
        }
        // ---- deferred emission: slot r = end position e_lo + r ----
        if (__ballot(any_hit) != 0) {
            for (int64_t r = 0; r < (int64_t)n_slots; ++r) {
                uint32_t sc1 = 0;
                if (any_hit && r <= e_hi - e_lo) {
                    sc1 = hitbuf[(size_t)r * nthr + tid];
                    hitbuf[(size_t)r * nthr + tid] = 0;
                }
                bool fresh = false;
                const int64_t e = e_lo + r;
                if (sc1) {
                    // dedupe across the seeds of one occurrence
                    const unsigned long long key = ((unsigned long long)pat << 40) | (unsigned long long)e;
                    uint32_t slot = (uint32_t)(mix64(key)) & P.seen_mask;
                    // the set holds at most one key per reportable hit at load <= 1/2; a probe sequence this long means
                    // more hits than the caller's buffer takes: flag it, the host re-runs with the brute engine
                    bool placed = false;
                    for (uint32_t tries = 0; tries < 512 && !placed; ++tries) {
                        const unsigned long long old = atomicCAS(&P.seen[slot], ~0ull, key);
                        if (old == ~0ull) {
                            fresh = true;
                            placed = true;
                        } else if (old == key) {
                            placed = true;
                        } else {
                            slot = (slot + 1) & P.seen_mask;
                        }
                    }
                    if (!placed)
                        atomicAdd(P.overflow, 1ull);
                }
                if (__ballot(fresh) != 0)
                    wave_append_hits(fresh, (P.report_begin ? (uint64_t)(e - m) : (uint64_t)e) + P.pos_offset, pat,
                                     (int32_t)sc1 - 1, P.hits, P.hit_counter, P.hit_cap);
            }
        }
    }
    if (P.cand_counter == 1) // raw candidates: how many reserved slots were real ones
        wave_count_add(P.hit_counter + 5, n_valid);
}


// ---- wave-per-candidate verification for long needles ----------------------------------------------------------
// One lane per 32-row block of the needle instead of one lane per candidate: G lanes form a systolic array that runs
// the block-based Myers recurrence (Myers 1999 / Hyyro 2003: every block takes the horizontal delta hin of the block
// above and hands its own hout down).  At step t lane b processes text column t - b; the text symbol and hout travel
// one lane per step through DPP wave_shr:1.  A column of a 1024-row needle costs one step of ~40 instructions
// instead of 32 blocks x 13 on a single lane, so the latency of one verification drops from ~1.5 ms to ~0.1 ms --
// what matters when a scan leaves a few hundred long bands to verify.  Reads the bottom-aligned table of the cut-off
// kernel ([group][row][nw_table][64]); each lane keeps the <= 5 match masks of its block in registers.
template <int G>
__global__ __launch_bounds__(256) void verify_wave_kernel(const verify_params P, const uint32_t *__restrict__ peq_bot)
{
    // per group of G lanes: [2*max_k + 1 + max_span] uint16 hit slots, then the candidate's text window (wave_text bytes)
    extern __shared__ uint16_t whit[];
    constexpr uint32_t GPW = 64 / G;
    const uint32_t lane = threadIdx.x & 63, gl = lane & (G - 1), gw = lane / G;
    const uint32_t wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint32_t n_slots = 2 * P.max_k + 1 + P.max_span;
    const uint32_t group_bytes = ((n_slots * 2 + 15) & ~15u) + P.wave_text;
    uint8_t *gbase = reinterpret_cast<uint8_t *>(whit) + (size_t)(wave * GPW + gw) * group_bytes;
    uint16_t *hb = reinterpret_cast<uint16_t *>(gbase);
    uint8_t *tw = gbase + ((n_slots * 2 + 15) & ~15u);
    for (uint32_t r = gl; r < n_slots; r += G)
        hb[r] = 0;
    unsigned long long n_cand = P.counters[P.cand_counter];
    if (n_cand > P.cand_cap)
        n_cand = P.cand_cap;
    const uint64_t stride = (uint64_t)gridDim.x * waves * GPW;
    const uint32_t rows = P.sigma + 1;
    uint32_t n_valid = 0;
    for (uint64_t base = ((uint64_t)blockIdx.x * waves + wave) * GPW; base < n_cand; base += stride) {
        const uint64_t ci = base + gw;
        bool active = ci < n_cand;
        uint32_t pat = 0;
        int64_t m = 1, k = 0, e_lo = 0, e_hi = -1, ws = 0;
        candidate c;
        c.val = kCandInvalid;
        if (active)
            c = P.cand[ci];
        active = active && c.val != kCandInvalid;
        if (active && gl == 0)
            ++n_valid;
        if (active) {
            pat = c.val >> 11;
            const bool merged = (c.pad & kCandMerged) != 0;
            const int64_t x = merged ? 0 : (int64_t)(c.val & 0x7FF) + (int64_t)c.pad;
            const int64_t span = merged ? (int64_t)(c.val & 0x7FF) : (int64_t)c.pad;
            m = P.m[pat];
            k = P.k[pat];
            const int64_t d = (int64_t)c.t - x;
            e_lo = d + m - k;
            e_hi = d + m + k + span;
            int64_t own_b = (int64_t)P.scan_begin, own_e = (int64_t)P.scan_end, hay_b = (int64_t)P.ctx_begin;
            if (P.seg_offsets) {
                uint64_t lo = 0, hi = P.n_segments;
                if (merged) {
                    lo = (c.pad & ~kCandMerged) - 1;
                } else {
                    while (hi - lo > 1) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if (P.seg_offsets[mid] <= c.t)
                            lo = mid;
                        else
                            hi = mid;
                    }
                }
                const int64_t sb = (int64_t)P.seg_offsets[lo], se = (int64_t)P.seg_offsets[lo + 1];
                if (!merged && (int64_t)c.t + (int64_t)P.key_len > se)
                    active = false;
                own_b = P.seg_owned ? sb + (int64_t)P.seg_owned[lo] : sb;
                own_e = se;
                hay_b = sb;
            }
            if (!merged && !seed_intact(P, c, hay_b, own_e))
                active = false;
            if (e_lo < own_b + 1)
                e_lo = own_b + 1;
            if (e_hi > own_e)
                e_hi = own_e;
            if (e_lo > e_hi)
                active = false;
            ws = e_lo - (m + k);
            if (ws < hay_b)
                ws = hay_b;
        }
        const uint32_t nb = (uint32_t)((m + 31) >> 5);        // blocks of this needle
        const bool is_last = gl + 1 == nb;
        const uint32_t out_bit = is_last ? (uint32_t)((m - 1) & 31) : 31u; // where this block's hout is read
        const uint32_t n_cols = active ? (uint32_t)(e_hi - ws) : 0u;
        const bool mine = active && gl < nb;
        // ---- stage the text window [ws, e_hi) into LDS: 16-byte blocks, the group's lanes side by side ----
        const uint32_t skew = (uint32_t)(ws & 15);
        if (active) {
            const uint32_t n_bytes = (skew + n_cols + 15) & ~15u;
            for (uint32_t off = gl * 16; off < n_bytes; off += G * 16) {
                const uint4 v = load_text16(P.text, ((uint64_t)ws & ~15ull) + off, P.text_alloc);
                *reinterpret_cast<uint4 *>(tw + off) = v;
            }
        }
        uint32_t e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0;
        if (mine) {
            const uint32_t *src = peq_bot + (((size_t)(pat >> 6) * rows) * P.nw_table + gl) * 64 + (pat & 63);
            const size_t rs = (size_t)P.nw_table * 64;
            e0 = src[0];
            e1 = src[rs];
            e2 = P.sigma > 2 ? src[2 * rs] : 0u;
            e3 = P.sigma > 3 ? src[3 * rs] : 0u;
            e4 = P.sigma > 4 ? src[4 * rs] : 0u;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t Pv = 0xFFFFFFFFu, Mv = 0, ho = 0;
        int32_t score = (int32_t)m;
        bool any_hit = false;
        const uint32_t first_slot_col = (uint32_t)(e_lo - ws) - 1; // column whose end position is e_lo
        const uint32_t t_end = n_cols + nb - 1;                    // steps until the last block has seen the last column
        // wave-uniform trip count: the longest group decides
        uint32_t t_wave = active ? t_end : 0u;
        for (int o = 32; o >= (int)G; o >>= 1)
            t_wave = max(t_wave, (uint32_t)__shfl_xor((int)t_wave, o));
        t_wave = (uint32_t)__builtin_amdgcn_readfirstlane(t_wave);
        const uint8_t *my_text = tw + skew - gl; // column t - gl of this lane = my_text[t]
        uint32_t sym_next = mine ? my_text[gl] : 0u; // column 0 (clamped reads below keep every index inside the window)
        for (uint32_t t = 0; t < t_wave; ++t) {
            const uint32_t ho_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ho, 0x138, 0xF, 0xF, false);
            const uint32_t col = t - gl; // wraps for t < gl: then col >= n_cols
            const uint32_t sym = sym_next;
            {   // next column's symbol, one step ahead of its use
                uint32_t nc = col + 1;
                nc = nc < n_cols ? nc : 0u;
                sym_next = tw[skew + nc];
            }
            if (mine && col < n_cols) {
                const uint32_t lo2 = (sym & 1u) ? e1 : e0, hi2 = (sym & 1u) ? e3 : e2;
                uint32_t Eq = (sym & 2u) ? hi2 : lo2;
                if (P.sigma != 4)
                    Eq = sym < 4 ? Eq : (sym == 4 ? e4 : 0u);
                Eq = sym < P.sigma ? Eq : 0u;
                const uint32_t hin = gl == 0 ? 0u : ho_up;
                const uint32_t hp = hin & 1u, hn = hin >> 1;
                const uint32_t Xv = Eq | Mv;
                Eq |= hn;
                const uint32_t Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                uint32_t Ph = Mv | ~(Xh | Pv);
                uint32_t Mh = Pv & Xh;
                const uint32_t op = (Ph >> out_bit) & 1u, on = (Mh >> out_bit) & 1u;
                ho = op | (on << 1);
                Ph = (Ph << 1) | hp;
                Mh = (Mh << 1) | hn;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                score += (int32_t)op - (int32_t)on;
                if (is_last && score <= (int32_t)k && col >= first_slot_col) {
                    hb[col - first_slot_col] = (uint16_t)(score + 1);
                    any_hit = true;
                }
            }
        }
        (void)my_text;
        // ---- emission: the lanes of a group share its slots; wave-converged appends ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (__ballot(any_hit) != 0) {
            for (uint32_t r0 = 0; r0 < n_slots; r0 += G) {
                const uint32_t r = r0 + gl;
                uint32_t sc1 = 0;
                if (active && r < n_slots && (int64_t)r <= e_hi - e_lo) {
                    sc1 = hb[r];
                    hb[r] = 0;
                }
                bool is_new = false;
                const int64_t e = e_lo + r;
                if (sc1) {
                    const unsigned long long key = ((unsigned long long)pat << 40) | (unsigned long long)e;
                    uint32_t slot = (uint32_t)(mix64(key)) & P.seen_mask;
                    bool placed = false;
                    for (uint32_t tries = 0; tries < 512 && !placed; ++tries) {
                        const unsigned long long old = atomicCAS(&P.seen[slot], ~0ull, key);
                        if (old == ~0ull) {
                            is_new = true;
                            placed = true;
                        } else if (old == key) {
                            placed = true;
                        } else {
                            slot = (slot + 1) & P.seen_mask;
                        }
                    }
                    if (!placed)
                        atomicAdd(P.overflow, 1ull);
                }
                if (__ballot(is_new) != 0)
                    wave_append_hits(is_new, (P.report_begin ? (uint64_t)(e - m) : (uint64_t)e) + P.pos_offset, pat,
                                     (int32_t)sc1 - 1, P.hits, P.hit_counter, P.hit_cap);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (P.cand_counter == 1)
        wave_count_add(P.hit_counter + 5, n_valid);
}

// ---- candidate compaction (needles with few errors) ----------------------------------------------------------------
// One lane per candidate means a wave is as slow as its slowest lane, so dropping chance candidates inside the
// verification kernel buys nothing while every wave still holds a true one.  This pass applies the whole-seed check
// first and hands the verification a dense list of the survivors.
__global__ void compact_candidates_kernel(const verify_params P, candidate *out, unsigned long long *out_count,
                                          uint64_t out_cap)
{
    unsigned long long n = P.counters[1];
    if (n > P.cand_cap)
        n = P.cand_cap;
    const uint32_t lane = threadIdx.x & 63;
    uint32_t n_valid = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride; // wave-uniform trip count: the appends are wave-collective
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool keep = false;
        candidate c;
        c.t = 0;
        c.val = kCandInvalid;
        c.pad = 0;
        if (i < n) {
            c = P.cand[i];
            keep = c.val != kCandInvalid;
            n_valid += keep ? 1u : 0u;
        }
        if (keep) {
            int64_t hay_b = (int64_t)P.ctx_begin, hay_e = (int64_t)P.scan_end;
            if (P.seg_offsets) {
                uint64_t lo = 0, hi = P.n_segments;
                while (hi - lo > 1) {
                    const uint64_t mid = (lo + hi) >> 1;
                    if (P.seg_offsets[mid] <= c.t)
                        lo = mid;
                    else
                        hi = mid;
                }
                hay_b = (int64_t)P.seg_offsets[lo];
                hay_e = (int64_t)P.seg_offsets[lo + 1];
                keep = (int64_t)c.t + (int64_t)P.key_len <= hay_e;
            }
            keep = keep && seed_intact(P, c, hay_b, hay_e);
        }
        const uint64_t m = __ballot(keep);
        if (m != 0) {
            unsigned long long base = 0;
            const int leader = __ffsll((unsigned long long)m) - 1;
            if ((int)lane == leader)
                base = atomicAdd(out_count, (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            const uint64_t idx = base + __popcll(m & ((1ull << lane) - 1));
            if (keep && idx < out_cap)
                out[idx] = c;
        }
    }
    wave_count_add(P.hit_counter + 5, n_valid);
}

// ---- candidate merging for large k ------------------------------------------------------------------------------
// With k+1 seeds every occurrence is verified once per surviving seed (65 times for k = 64) and short keys let chance
// matches through.  Needles with k >= kMergeMinK therefore carry k+2 seeds: an occurrence with <= k errors keeps >= 2 of
// them intact, on diagonals at most k apart.  Candidates are counted per (needle, haystack, diagonal band); a band is
// verified once, over every end position its diagonals can produce, if it collected as many seed hits as the needle
// has surplus seeds (1 for needles that kept k+1 seeds).  Bands are Bw diagonals wide and overlap by k, so the intact
// seeds of one occurrence always share a band.
struct merge_params
{
    const candidate *cand;
    const unsigned long long *counters; // [1] raw candidates
    unsigned long long *out_count;      // counters[3]
    uint64_t cand_cap;
    const int32_t *m, *k;
    const uint8_t *surplus; // per needle: seeds - k
    uint32_t key_len, max_m, Bw, table_mask;
    uint64_t hay_begin; // unsegmented scans: first symbol of the haystack
    const uint64_t *seg_offsets;
    uint64_t n_segments;
    uint2 *aux;      // per candidate: {segment, primary band}; segment 0xFFFFFFFF = dropped
    uint32_t *owner; // band table: (candidate << 1 | secondary) of the first arrival, 0xFFFFFFFF = empty
    uint32_t *count; // seed hits per band
    uint2 *own_slot; // per candidate: the table slots it claimed (primary, secondary band), 0xFFFFFFFF = none
    const uint32_t *seg_owned; // optional (verify_params::seg_owned): bands that end before it are not verified
    verify_params V;           // text, needles, scan range: the whole-seed check of every candidate
    candidate *out;
    uint64_t out_cap;
};

__global__ void merge_aux_kernel(const merge_params P)
{
    unsigned long long n = P.counters[1];
    if (n > P.cand_cap)
        n = P.cand_cap;
    uint32_t n_valid = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const candidate c = P.cand[i];
        n_valid += c.val != kCandInvalid ? 1u : 0u;
        uint64_t seg = 0;
        int64_t sb = (int64_t)P.hay_begin;
        bool drop = c.val == kCandInvalid;
        if (P.seg_offsets) {
            uint64_t lo = 0, hi = P.n_segments;
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (P.seg_offsets[mid] <= c.t)
                    lo = mid;
                else
                    hi = mid;
            }
            seg = lo;
            sb = (int64_t)P.seg_offsets[lo];
            drop = drop || (int64_t)c.t + (int64_t)P.key_len > (int64_t)P.seg_offsets[lo + 1];
            if (!drop)
                drop = !seed_intact(P.V, c, sb, (int64_t)P.seg_offsets[lo + 1]);
        } else if (!drop) {
            drop = !seed_intact(P.V, c, (int64_t)P.V.ctx_begin, (int64_t)P.V.scan_end);
        }
        // diagonal relative to the haystack, shifted so that it is never negative (t >= sb, offset <= max_m)
        const int64_t dr = (int64_t)c.t - (int64_t)(c.val & 0x7FF) - sb + (int64_t)P.max_m;
        P.aux[i] = drop ? make_uint2(0xFFFFFFFFu, 0u) : make_uint2((uint32_t)seg, (uint32_t)(dr / (int64_t)P.Bw));
    }
    wave_count_add(P.out_count + 2, n_valid); // counters[5]
}

__global__ void merge_count_kernel(const merge_params P)
{
    unsigned long long n = P.counters[1];
    if (n > P.cand_cap)
        n = P.cand_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 a = P.aux[i];
        uint32_t claimed[2] = {0xFFFFFFFFu, 0xFFFFFFFFu};
        if (a.x != 0xFFFFFFFFu) {
            const candidate c = P.cand[i];
            const uint32_t pat = c.val >> 11;
            const int64_t sb = P.seg_offsets ? (int64_t)P.seg_offsets[a.x] : (int64_t)P.hay_begin;
            const int64_t dr = (int64_t)c.t - (int64_t)(c.val & 0x7FF) - sb + (int64_t)P.max_m;
            // band b covers diagonals [b*Bw, (b+1)*Bw + k]: the primary band, and the one before it if this diagonal
            // still lies in its k-wide extension
            const bool also_prev = a.y > 0 && dr - (int64_t)a.y * P.Bw <= (int64_t)P.k[pat];
            for (uint32_t sec = 0; sec <= (also_prev ? 1u : 0u); ++sec) {
                const uint32_t band = a.y - sec;
                const uint32_t me = (uint32_t)(i << 1) | sec;
                uint32_t s = (uint32_t)mix64(((uint64_t)pat << 40) ^ ((uint64_t)a.x << 17) ^ band) & P.table_mask;
                while (true) {
                    uint32_t o = atomicCAS(&P.owner[s], 0xFFFFFFFFu, me);
                    if (o == 0xFFFFFFFFu)
                        o = me;
                    const uint2 b = P.aux[o >> 1];
                    if ((P.cand[o >> 1].val >> 11) == pat && b.x == a.x && b.y - (o & 1u) == band) {
                        atomicAdd(&P.count[s], 1u);
                        if (o == me)
                            claimed[sec] = s;
                        break;
                    }
                    s = (s + 1) & P.table_mask;
                }
            }
        }
        P.own_slot[i] = make_uint2(claimed[0], claimed[1]);
    }
}

// one thread per candidate: the claimer of a band emits it if the band collected enough seed hits
__global__ void merge_select_kernel(const merge_params P)
{
    unsigned long long n = P.counters[1];
    if (n > P.cand_cap)
        n = P.cand_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint2 mine = P.own_slot[i];
        if (mine.x == 0xFFFFFFFFu && mine.y == 0xFFFFFFFFu)
            continue;
        const uint32_t pat = P.cand[i].val >> 11;
        const uint2 a = P.aux[i];
        const int64_t sb = P.seg_offsets ? (int64_t)P.seg_offsets[a.x] : (int64_t)P.hay_begin;
        for (uint32_t sec = 0; sec < 2; ++sec) {
            const uint32_t s = sec ? mine.y : mine.x;
            if (s == 0xFFFFFFFFu || P.count[s] < P.surplus[pat])
                continue;
            const uint32_t band = a.y - sec;
            const int64_t d_lo = sb - (int64_t)P.max_m + (int64_t)band * P.Bw;
            // last end position the band can produce: diagonal d_lo + Bw + k, needle length m, k more symbols
            if (P.seg_owned && d_lo + (int64_t)P.Bw + 2 * (int64_t)P.k[pat] + (int64_t)P.m[pat] <= sb + (int64_t)P.seg_owned[a.x])
                continue; // everything this band can report ends inside the haystack's unwanted prefix
            const unsigned long long idx = atomicAdd(P.out_count, 1ull);
            if (idx < P.out_cap) {
                candidate c;
                c.t = (uint64_t)(sb - (int64_t)P.max_m + (int64_t)band * P.Bw); // first diagonal of the band
                c.val = (pat << 11) | (P.Bw + (uint32_t)P.k[pat]);                 // its last diagonal: + Bw + k
                c.pad = kCandMerged | (a.x + 1);
                P.out[idx] = c;
            }
        }
    }
}

} // namespace spm_hip
