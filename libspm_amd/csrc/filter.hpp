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
//   reads per window; a Bloom cascade is kept as fallback) and records the windows that pass (survivors).  Needle
//   sets of several stride-1 passes use anchored keys: a pass looks up only the windows that begin with its dimer.
//   Resolve (resolve_kernel): survivors -> the key's entries through an exact key directory in L2 -> whole-seed and
//   piece-count checks -> diagonal bands.
//   Verification: the Myers recurrence over the m+3k symbols around the candidate diagonal, cold-started
//   m+k symbols before the first end position it is responsible for -- exact by the window property used for
//   tiling.  Short needles: one lane per candidate (verify_kernel); long needles: one lane per 32-row block
//   (verify_wave_kernel).  Duplicates (several seeds of one occurrence) are removed with an atomicCAS hash set keyed
//   by (needle, end).
//
// Exactness does not depend on the text being random: every true hit has a surviving seed, every candidate is
// verified by the full recurrence.  Pathological inputs only cost time: a span of the text that yields more survivors
// than its budget is scanned again by the brute-force kernel, that span alone.
#pragma once

#include "brute.hpp"
#include "common.hpp"
#include "filter_shared.hpp"
#include "pack.hpp"
#include "hd.hpp" // mix64

namespace spm_hip
{

// What the streaming kernel leaves behind: one record per text window whose key passed level 1 (practically: a real
// seed key).  The exact key table in L2 is NOT consulted while streaming -- an L2 round trip per survivor stalls a wave
// for microseconds, and a repeat-rich text has 10^7 of them; resolve_kernel looks them up afterwards, one lane per
// survivor, at full occupancy.
struct survivor
{
    uint32_t key;
    uint32_t t_lo, t_hi; // text index of the window start; t_hi == kSurvInvalid marks a reserved but unused slot
    uint32_t pad;        // pass (needle sub-batch) whose level-1 table it passed
};
constexpr uint32_t kSurvInvalid = 0xFFFFFFFFu;

// What verification works on: a band of diagonals of one needle in one haystack (segment) that collected enough seed hits.
// Bands are Bw diagonals wide, counted from `max_m` diagonals before the haystack's first symbol (so they are never
// negative); sets whose needles carry surplus seeds (k >= kMergeMinK) extend every band by k diagonals into the next one,
// so that the intact seeds of one occurrence always share a band.
struct band_rec
{
    uint32_t slot; // band table slot (seed-hit count; reset by the verification)
    uint32_t val;  // pattern << 11 | unused;  kBandInvalid marks a reserved but unused list slot
    uint32_t seg;  // segment index (0 for unsegmented scans)
    uint32_t band; // band index inside the haystack
};
constexpr uint32_t kBandInvalid = 0xFFFFFFFFu;
constexpr unsigned long long kBandEmpty = ~0ull;

// Slots of the survivor and band lists are handed to the waves in chunks: one atomic on the shared counter per chunk
// instead of one per ballot (a single address takes ~100 atomics/us).  Chunks grow as a wave keeps drawing -- 32, 64, ..
// 1024 -- so a quiet text costs a few hundred wasted slots and a repeat-rich one (10^7 survivors) a few thousand atomics.
// Slots a wave reserved but did not fill are marked invalid.
constexpr uint32_t kChunkMin = 32;
constexpr uint32_t kChunkMax = 1024;
// The chunk a wave of the streaming kernel is filling lives in LDS behind the level-1 table (kCandRec words per wave:
// {base lo, base hi, used, size, survivors of the current span, span given up, span begin lo, hi}; touched only on the
// rare survivor path and once per span, so nothing stays live across the streaming loop): slots [base + used, base + size)
// are free.
//
// Span budget: a span (the unit of work a wave dequeues) that produces more than `span_budget` survivors -- or meets a
// full survivor buffer -- gives up: it emits nothing more, its text range goes to the overflow list, and the host
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
    uint32_t key_len;         // H: symbols per key (12..16); windows are H symbols, keys 2H bits
    uint32_t key_mask;        // (1 << 2H) - 1
    uint32_t hash_variant;    // 0/1: Bloom cascade with mul / xor-shift hashes, 2: perfect-hash fingerprints
    uint32_t lds_words;       // size of the LDS image (bitmap, or fingerprint table + displacement table)
    uint32_t chd_slot_mask;   // fingerprint slots - 1
    uint32_t chd_bucket_shift; // bucket = x >> shift
    uint32_t chd_disp_off;    // byte offset of the displacement table inside the LDS image
    const uint32_t *bitmap;   // [bitmap_words]
    uint32_t span_budget;     // survivors one span may produce before it gives up
    uint32_t pass;            // which sub-batch of the needle set this launch filters for (survivor::pad)
    uint32_t anchor_c, anchor_cm; // anchored passes: a window is looked up iff its leading dimer d = sym0 | sym1 << 2 has
                                  // (d ^ anchor_c) & anchor_cm == 0
    // dense passes (filter_shared.hpp): lds = presence bits; a window is looked up iff its dimer matches one of the n_pat patterns
    uint32_t n_pat, pat_c[kDensePatterns], pat_cm[kDensePatterns];
    uint32_t bucket_shift;
    uint32_t dense_debug;     // diagnostics (SPM_HIP_DENSE_DEBUG): 1 drop the windows that pass level 1, 2 skip level 1 too, 4 gather from 256 buckets only (L1 hits; wrong hits)
    const uint4 *buckets;     // fingerprint buckets, L2-resident
    survivor *surv;
    unsigned long long *counters; // [1] = survivor slots drawn, [6] = spans that gave up, [2] = hard overflow
    uint64_t surv_cap;
    uint64_t *ovf_spans;      // [ovf_cap][2]: {first text index, symbols} of every span that gave up
    uint64_t ovf_cap;
};

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

// A wave starts a span: fresh budget (wave-uniform call).
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
__device__ __forceinline__ void span_give_up(const filter_params &P, uint32_t *ck, uint32_t lane)
{
    if (lane == 0) {
        ck[5] = 1;
        const unsigned long long i = atomicAdd(&P.counters[6], 1ull);
        if (i < P.ovf_cap) {
            P.ovf_spans[2 * i] = ((uint64_t)ck[7] << 32) | ck[6];
            P.ovf_spans[2 * i + 1] = (uint64_t)P.span_chunks * (uint64_t)P.span_unit;
        } else {
            atomicAdd(&P.counters[2], 1ull); // no room to remember it: the host re-runs the whole scan
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// mark the unused tail of a wave's chunk invalid (wave-uniform call)
__device__ __forceinline__ void surv_close(const filter_params &P, const uint32_t *ck, uint32_t lane)
{
    const uint32_t used = ck[2], size = ck[3];
    const uint64_t base = ((uint64_t)ck[1] << 32) | ck[0];
    for (uint32_t i = used + lane; i < size; i += 64)
        if (base + i < P.surv_cap) {
            survivor sv;
            sv.key = 0;
            sv.t_lo = 0;
            sv.t_hi = kSurvInvalid;
            sv.pad = 0;
            P.surv[base + i] = sv;
        }
}

// Record the survivors of one wave step (wave-uniform call; `has` marks the lanes that hold one).
__device__ __forceinline__ void emit_survivors(const filter_params &P, bool has, uint32_t key, uint64_t t, uint32_t lane,
                                               const uint32_t *lds)
{
    const uint64_t m = __ballot(has);
    if (m == 0)
        return;
    uint32_t *ck = cand_chunk_of(P, lds);
    const uint32_t n = __popcll(m);
    uint32_t used = (uint32_t)__builtin_amdgcn_readfirstlane(ck[2]);
    const uint32_t size = (uint32_t)__builtin_amdgcn_readfirstlane(ck[3]);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane(ck[4]) + n;
    if (cnt > P.span_budget) {
        span_give_up(P, ck, lane);
        return;
    }
    if (used + n > size) { // close this chunk, draw the next (twice as large, up to kChunkMax)
        surv_close(P, ck, lane);
        uint32_t next = size * 2 < kChunkMin ? kChunkMin : (size * 2 > kChunkMax ? kChunkMax : size * 2);
        next = next < n ? n : next;
        if (lane == 0) {
            const unsigned long long b = atomicAdd(&P.counters[1], (unsigned long long)next);
            ck[0] = (uint32_t)b;
            ck[1] = (uint32_t)(b >> 32);
            ck[2] = 0;
            ck[3] = next;
        }
        used = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const uint64_t cbase = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(ck[1]) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane(ck[0]);
    if (cbase + used + n > P.surv_cap) { // the survivor buffer is full
        span_give_up(P, ck, lane);
        return;
    }
    if (has) {
        survivor sv;
        sv.key = key;
        sv.t_lo = (uint32_t)t;
        sv.t_hi = (uint32_t)(t >> 32);
        sv.pad = P.pass;
        P.surv[cbase + used + __popcll(m & ((1ull << lane) - 1))] = sv;
    }
    if (lane == 0) {
        ck[2] = used + n;
        ck[4] = cnt;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Level 1 + level 2 on NWD 2-bit-packed words per lane (w[j] = 16 bases, prev[j] = the 16 bases before them).
// PK selects how word j maps to a text position (1-byte text vs. packed shadow).
// KM: keys shorter than 16 symbols (masked); only strides 1 and 2 ever carry such keys.
template <int S, int NWD, int HV, int SIG, bool PK, bool KM>
__device__ __forceinline__ void filter_words(const filter_params &P, const uint32_t (&w)[NWD],
                                             const uint32_t (&prev)[NWD], const uint32_t (&nv)[SIG != 4 ? NWD : 1],
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
    if constexpr (SIG != 4) {
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
    // survivors: recorded for resolve_kernel (one per wave step and lane until every lane's mask is empty)
    while (__ballot(pos_mask != 0) != 0) {
        if (__builtin_amdgcn_readfirstlane(cand_chunk_of(P, lds)[5]) != 0)
            break; // the span has given up
        uint64_t t = 0;
        uint32_t key = 0;
        bool has = false;
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
                has = true;
            }
        }
        emit_survivors(P, has, key, t, lane, lds);
    }
}


// ---- anchored passes (stride 1, needle sets of several passes) ------------------------------------------------------
// The keys of the pass all begin with its anchor dimer (the host chose each seed's key window that way; a few passes
// carry a pattern with a don't-care bit, i.e. two dimers), so only the text windows that begin with it -- 1 in 16 -- can
// equal a key.  Which of a lane's 16 windows those are is computed bit-parallel on the packed word (9 VALU for 16
// windows); the hash and the two LDS reads then run for the selected windows only, in a per-lane loop.  The unanchored
// stride-1 kernel spends 14 VALU and two conflict-ridden LDS reads on EVERY window and saturates both the VALU and the
// LDS.

// bit 2 (d - 1) set: window d (1..16) of this word -- it starts at symbol d of (prev, w) -- begins with the anchor
__device__ __forceinline__ uint32_t anchor_select(uint32_t w, uint32_t prev, uint32_t c, uint32_t cm)
{
    // (wave-uniform) the pattern's bits for the first and the second symbol, repeated for every symbol of a word
    const uint32_t C0 = (c & 3u) * 0x55555555u, M0 = (cm & 3u) * 0x55555555u;
    const uint32_t C1 = (c >> 2) * 0x55555555u, M1 = (cm >> 2) * 0x55555555u;
    const uint32_t X = alignbit(w, prev, 2); // symbols 1..16: where the windows start
    const uint32_t Y = alignbit(w, prev, 4); // symbols 2..17: their second symbols
    uint32_t t = ((X ^ C0) & M0) | ((Y ^ C1) & M1); // a set bit: that bit of that symbol differs from the pattern
    t |= t >> 1;
    return ~t & 0x55555555u;
}

template <int NWD>
__device__ __forceinline__ void filter_words_anchored(const filter_params &P, const uint32_t (&w)[NWD], const uint32_t (&prev)[NWD],
                                                      uint64_t gbase, uint32_t lane, const uint32_t *lds)
{
    static_assert(NWD % 2 == 0 || NWD == 1, "words are taken in pairs");
    const uint16_t *fp_tab = reinterpret_cast<const uint16_t *>(lds);
    const uint16_t *disp_tab = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(lds) + P.chd_disp_off);
#pragma unroll
    for (int j = 0; j < NWD; j += 2) {
        // the selected windows of two words in one mask: bit b = word j + (b & 1), window (b >> 1) + 1
        uint32_t todo = anchor_select(w[j], prev[j], P.anchor_c, P.anchor_cm);
        if (j + 1 < NWD)
            todo |= anchor_select(w[j + 1 < NWD ? j + 1 : j], prev[j + 1 < NWD ? j + 1 : j], P.anchor_c, P.anchor_cm) << 1;
        while (__ballot(todo != 0) != 0) {
            const bool act = todo != 0;
            const uint32_t b = act ? (uint32_t)__ffs(todo) - 1u : 0u;
            todo &= todo - 1;
            const uint32_t u = b & 1u, d = (b >> 1) + 1u;
            const uint32_t wu = (j + 1 < NWD && u) ? w[j + 1 < NWD ? j + 1 : j] : w[j];
            const uint32_t pu = (j + 1 < NWD && u) ? prev[j + 1 < NWD ? j + 1 : j] : prev[j];
            const uint32_t key = (uint32_t)((((uint64_t)wu << 32) | pu) >> (2 * d));
            const uint32_t x = chd_hash(key).x;
            const uint32_t dsp = disp_tab[x >> P.chd_bucket_shift];
            const uint32_t slot = ((x >> 3) + dsp * ((key | 1u) & 0xFFFFFFu)) & P.chd_slot_mask;
            const uint32_t f = fp_tab[slot];
            const bool hit = act && ((f ^ key ^ (key >> 16)) & 0xFFFFu) == 0;
            if (__ballot(hit) != 0) { // rare: a survivor for resolve_kernel
                if (__builtin_amdgcn_readfirstlane(cand_chunk_of(P, lds)[5]) != 0)
                    return; // the span has given up
                const uint64_t wpos = gbase + (uint64_t)(j + u) * 1024 + (uint64_t)lane * 16;
                const int64_t ts = (int64_t)wpos - 16 + (int64_t)d;
                const bool has = hit && ts >= (int64_t)P.lo && (uint64_t)ts + P.key_len <= P.hi;
                emit_survivors(P, has, key, has ? (uint64_t)ts : 0, lane, lds);
            }
        }
    }
}

// One group of UU consecutive 1-KiB chunks, already in registers.  chunk u of the group starts at text index
// gbase + 1024*u; this lane holds its bytes [16*lane, 16*lane+16).
template <int S, int UU, int HV, int SIG, bool KM, bool AN = false>
__device__ __forceinline__ void filter_group(const filter_params &P, const uint4 (&cur)[UU], uint64_t gbase,
                                             uint32_t &carry_in, uint32_t &carry_n, uint32_t lane,
                                             const uint32_t *lds, uint32_t idx_mask)
{
    uint32_t w[UU], prev[UU];
    uint32_t nv[SIG != 4 ? UU : 1]; // dna5 / dna15: (this lane's N mask << 16) | previous lane's N mask
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        uint32_t nm = 0;
        w[u] = pack16_sig<SIG>(cur[u], nm);
        prev[u] = __builtin_amdgcn_update_dpp(0u, w[u], 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
        if (lane == 0)
            prev[u] = carry_in;
        carry_in = __builtin_amdgcn_readlane(w[u], 63);
        if (SIG != 4) {
            uint32_t np = __builtin_amdgcn_update_dpp(0u, nm, 0x138, 0xF, 0xF, false);
            if (lane == 0)
                np = carry_n;
            carry_n = __builtin_amdgcn_readlane(nm, 63);
            nv[u] = (nm << 16) | np;
        }
    }
    if constexpr (AN)
        filter_words_anchored<UU>(P, w, prev, gbase, lane, lds);
    else
        filter_words<S, UU, HV, SIG, false, KM>(P, w, prev, nv, gbase, lane, lds, idx_mask);
}

// Streaming structure: a wave owns spans of consecutive 1-KiB chunks; per iteration it works on a GROUP of U chunks
// (U KiB contiguous per wave) while the unconditional 16-byte loads of the next group are already in flight, so
// each wave keeps 2*U KiB outstanding.  Only groups that lie fully inside the text take this path; the ragged end
// of the text goes through a guarded one-chunk loop (bytes past the end read as 0).
// 512 threads: 8 waves per CU is the measured best for the HBM-bound 1-byte text, and a bound of 1024 leaves 128 VGPRs,
// which made the masked-key stride-2 variant spill to scratch inside the streaming loop (+31 % kernel time).
// Stride 1 without key masks (C4-sized needle sets: LDS-bound at 16 windows per lane, see DESIGN) fits 128 VGPRs and may run
// 16 waves per CU.
template <int S, int U, bool NT, int HV, int SIG, bool KM, bool AN = false>
__global__ __launch_bounds__(((S == 1 && !KM && SIG == 4) || (S == 2 && U == 2)) ? 1024 : 512) void seed_filter_kernel(const filter_params P)
{
    extern __shared__ uint32_t lds[];
    // ---- stage the level-1 table in LDS (once per workgroup; the grid is persistent) ----
    for (uint32_t i = threadIdx.x; i < P.lds_words; i += blockDim.x)
        lds[i] = P.bitmap[i];
    if ((threadIdx.x & 63) == 0) { // this wave's candidate chunk: none drawn yet
        uint32_t *ck = lds + P.lds_words + kCandLdsSlot + kCandRec * (threadIdx.x >> 6);
        for (uint32_t i = 0; i < kCandRec; ++i)
            ck[i] = 0; // size 0: the first survivor draws a chunk; no span open yet
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
                carry_in = pack16_sig<SIG>(before, carry_n);
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
                filter_group<S, U, HV, SIG, KM, AN>(P, cur, base0 + ch * 1024, carry_in, carry_n, lane, lds, idx_mask);
            }
        }
        // ---- ragged end ----
        for (; ch < c_end; ++ch) {
            uint4 one[1];
            one[0] = load_text16(P.text, base0 + ch * 1024 + (uint64_t)lane * 16, P.hi);
            filter_group<S, 1, HV, SIG, KM, AN>(P, one, base0 + ch * 1024, carry_in, carry_n, lane, lds, idx_mask);
        }
        sp += n_waves;
    }
    surv_close(P, cand_chunk_of(P, lds), lane);
}

// ---------------------------------------------------------------------------------------------------
// Dense pass: every key of a needle set of any size in ONE pass over the text (filter_shared.hpp, index_build.hpp).
//   level 0   which of a lane's 16 windows per word begin with an anchor dimer: bit-parallel on the packed word
//             (dense_select: ~6 VALU per pattern and word); 1/8 .. 3/16 of the windows for the usual sets;
//   level 1   one presence bit per key in LDS (2^20 bits), looked up for the anchored windows in a per-lane loop over the
//             set bits of two words; about a third of them pass with 400 000 keys;
//   level 1b  those wait in a per-wave LDS queue {key, offset in the span} until 64 or more are there; then the wave takes
//             them one per lane, gathers each one's 16-byte fingerprint bucket from L2 (up to kDenseBatches batches in
//             flight together: the chip serves ~2.7e11 such gathers per second, tools/l2_gather_probe.hip, so the queue
//             exists to issue them 64 lanes wide and to keep several round trips overlapping), and compares the 15-bit
//             fingerprint with the bucket's eight slots;
//   what matches is a survivor like those of the sparse passes: recorded for resolve_kernel.
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t kDenseQueueCap = 192; // entries per wave
constexpr int kDenseBatches = 3;

template <int NP>
__device__ __forceinline__ uint32_t dense_select(const filter_params &P, uint32_t w, uint32_t prev)
{
    const uint32_t X = alignbit(w, prev, 2); // symbols 1..16: where the windows start
    const uint32_t Y = alignbit(w, prev, 4); // symbols 2..17: their second symbols
    uint32_t sel = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        // (wave-uniform) the pattern's bits for the first and the second symbol, repeated for every symbol of a word
        const uint32_t C0 = (P.pat_c[i] & 3u) * 0x55555555u, M0 = (P.pat_cm[i] & 3u) * 0x55555555u;
        const uint32_t C1 = (P.pat_c[i] >> 2) * 0x55555555u, M1 = (P.pat_cm[i] >> 2) * 0x55555555u;
        uint32_t t = ((X ^ C0) & M0) | ((Y ^ C1) & M1); // a set bit: that bit of that symbol differs from the pattern
        t |= t >> 1;
        sel |= ~t;
    }
    return sel & 0x55555555u; // bit 2 (d - 1): window d (1..16) of this word begins with an anchor
}

struct dense_queue // per wave, in LDS behind the survivor-chunk records
{
    uint32_t key[kDenseQueueCap];
    uint32_t off[kDenseQueueCap]; // window start - span begin + 16
};

// Take queued windows through level 1b: full batches of 64 (`all`: the rest too).  Wave-uniform call.
__device__ __forceinline__ void dense_drain(const filter_params &P, dense_queue &Q, uint32_t &qn, bool all, uint64_t span_base,
                                            uint32_t lane, const uint32_t *lds)
{
    while (qn >= 64 || (all && qn != 0)) {
        const uint32_t nb_full = qn >> 6;
        uint32_t nb = all ? (qn + 63) >> 6 : nb_full;
        nb = nb > (uint32_t)kDenseBatches ? (uint32_t)kDenseBatches : nb;
        uint32_t key[kDenseBatches], off[kDenseBatches];
        uint4 bk[kDenseBatches];
        bool v[kDenseBatches];
#pragma unroll
        for (int b = 0; b < kDenseBatches; ++b) { // entries [qn - 64 (b + 1), qn - 64 b), from the top of the queue
            const int32_t e = (int32_t)qn - 64 * b - 1 - (int32_t)lane;
            v[b] = (uint32_t)b < nb && e >= 0;
            key[b] = 0;
            off[b] = 0;
            bk[b] = make_uint4(0, 0, 0, 0);
            if (v[b]) {
                key[b] = Q.key[e];
                off[b] = Q.off[e];
                bk[b] = P.buckets[dense_bucket(key[b], P.bucket_shift) & ((P.dense_debug & 4u) ? 255u : 0xFFFFFFFFu)];
            }
        }
        qn = qn > 64 * nb ? qn - 64 * nb : 0;
#pragma unroll
        for (int b = 0; b < kDenseBatches; ++b) {
            if ((uint32_t)b >= nb)
                break; // (wave-uniform)
            const uint32_t f2 = dense_fp(key[b]) * 0x00010001u;
            // a 16-bit slot equal to the fingerprint <=> a zero halfword in slot ^ fingerprint
            const uint32_t x0 = bk[b].x ^ f2, x1 = bk[b].y ^ f2, x2 = bk[b].z ^ f2, x3 = bk[b].w ^ f2;
            const uint32_t z = ((x0 - 0x00010001u) & ~x0) | ((x1 - 0x00010001u) & ~x1) | ((x2 - 0x00010001u) & ~x2) |
                               ((x3 - 0x00010001u) & ~x3);
            const bool hit = v[b] && ((z & 0x80008000u) != 0 || (bk[b].w >> 16) == kDenseAcceptAll);
            if (__ballot(hit) != 0) { // rare: a survivor for resolve_kernel
                if (__builtin_amdgcn_readfirstlane(cand_chunk_of(P, lds)[5]) == 0) { // (unless the span has given up)
                    const int64_t ts = (int64_t)span_base + (int64_t)off[b] - 16;
                    const bool has = hit && ts >= (int64_t)P.lo && (uint64_t)ts + P.key_len <= P.hi;
                    emit_survivors(P, has, key[b], has ? (uint64_t)ts : 0, lane, lds);
                }
            }
        }
    }
}

// S == 0: the anchored windows of a dense pass (NP patterns).  S = 1, 2: EVERY S-th window of a sparse pass whose level 1 is
// the presence table (hash variant 4): the same number for every lane, at fixed places -- unrolled, no search for set bits.
template <int UU, int NP, int S, bool KM>
__device__ __forceinline__ void dense_group(const filter_params &P, const uint4 (&cur)[UU], uint64_t gbase, uint64_t span_base,
                                            uint32_t &carry_in, uint32_t lane, const uint32_t *lds, dense_queue &Q, uint32_t &qn)
{
    static_assert(UU % 2 == 0 || UU == 1, "words are taken in pairs");
    uint32_t w[UU], prev[UU];
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        w[u] = pack16(cur[u]);
        prev[u] = __builtin_amdgcn_update_dpp(0u, w[u], 0x138 /*wave_shr:1*/, 0xF, 0xF, false);
        if (lane == 0)
            prev[u] = carry_in;
        carry_in = __builtin_amdgcn_readlane(w[u], 63);
    }
    const uint32_t goff = (uint32_t)(gbase - span_base) + lane * 16; // (this lane's first base of chunk 0) - span begin
#pragma unroll
    for (int j = 0; j < UU; j += 2) {
        constexpr bool pair = UU > 1;
        const uint32_t w0 = w[j], p0 = prev[j], w1 = w[pair ? j + 1 : j], p1 = prev[pair ? j + 1 : j];
        const uint32_t kmask = KM ? P.key_mask : 0xFFFFFFFFu;
        static_assert(kDenseBloomBits == 20, "the word address below takes bits 5..19 of the index");
        // pm: bit b = word j + (b & 1), window (b >> 1) + 1 has its presence bit set
        uint32_t pm = 0;
        if constexpr (S == 0) {
            // the anchored windows of two words in one mask
            uint32_t todo = dense_select<NP>(P, w0, p0);
            if (pair)
                todo |= dense_select<NP>(P, w1, p1) << 1;
            // level 1, every lane for itself (a lane runs as long as it has anchored windows -- no wave-wide steps in
            // here, the loop is what this kernel spends its time in): which of them have their presence bit set
            if (P.dense_debug & 2u)
                todo = 0;
            while (todo != 0) {
                const uint32_t b = (uint32_t)__ffs(todo) - 1u;
                todo &= todo - 1;
                const bool odd = (b & 1u) != 0;
                const uint64_t both = ((uint64_t)(odd ? w1 : w0) << 32) | (odd ? p1 : p0);
                const uint32_t key = (uint32_t)(both >> ((b & ~1u) + 2u)); // window d = (b >> 1) + 1 starts 2 d bits in
                const uint32_t h = key ^ (key >> 13); // dense_bloom_index(key) = h & (2^20 - 1): word h >> 5, bit h & 31
                const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lds) + ((h >> 3) & 0x1FFFCu));
                pm |= __builtin_amdgcn_ubfe(word, h, 1) << b;
            }
        } else {
#pragma unroll
            for (int u = 0; u < (pair ? 2 : 1); ++u) {
#pragma unroll
                for (int d = S; d <= 16; d += S) {
                    const uint32_t wu = u ? w1 : w0, pu = u ? p1 : p0;
                    const uint32_t key = (d == 16 ? wu : alignbit(wu, pu, (2 * d) & 31)) & kmask;
                    const uint32_t h = key ^ (key >> 13);
                    const uint32_t word = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(lds) + ((h >> 3) & 0x1FFFCu));
                    pm |= __builtin_amdgcn_ubfe(word, h, 1) << (2 * (d - 1) + u);
                }
            }
        }
        // level 1b: those that do wait in the queue until the wave has a batch of them (wave-wide steps from here on)
        if (P.dense_debug & 1u)
            pm = 0;
        while (__ballot(pm != 0) != 0) {
            const bool pos = pm != 0;
            const uint32_t b = pos ? (uint32_t)__ffs(pm) - 1u : 0u;
            pm &= pm - 1;
            const bool odd = (b & 1u) != 0;
            const uint64_t both = ((uint64_t)(odd ? w1 : w0) << 32) | (odd ? p1 : p0);
            const uint32_t key = (uint32_t)(both >> ((b & ~1u) + 2u)) & kmask;
            const uint64_t mm = __ballot(pos);
            if (pos) {
                const uint32_t q = qn + __popcll(mm & ((1ull << lane) - 1));
                Q.key[q] = key;
                Q.off[q] = goff + (uint32_t)j * 1024u + (odd ? 1024u : 0u) + (b >> 1) + 1u; // = window start - span begin + 16
            }
            qn += __popcll(mm);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (qn > kDenseQueueCap - 64)
                dense_drain(P, Q, qn, false, span_base, lane, lds);
        }
    }
}

template <int U, int NP, int S = 0, bool KM = false>
__global__ __launch_bounds__(1024) void seed_filter_dense_kernel(const filter_params P)
{
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < P.lds_words; i += blockDim.x)
        lds[i] = P.bitmap[i];
    if ((threadIdx.x & 63) == 0) { // this wave's candidate chunk: none drawn yet
        uint32_t *ck = lds + P.lds_words + kCandLdsSlot + kCandRec * (threadIdx.x >> 6);
        for (uint32_t i = 0; i < kCandRec; ++i)
            ck[i] = 0;
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t wave_in_wg = (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;
    dense_queue &Q = *reinterpret_cast<dense_queue *>(lds + P.lds_words + kCandLdsSlot + kCandRec * 16 +
                                                      (sizeof(dense_queue) / 4) * wave_in_wg);
    uint32_t qn = 0;

    const uint64_t base0 = P.lo & ~1023ull; // chunks are 1 KiB aligned relative to text[0]
    const uint64_t n_chunks = (P.hi - base0 + 1023) / 1024;
    const uint64_t n_whole = (P.hi - base0) / 1024; // chunks that lie fully inside the text
    const uint64_t span = P.span_chunks;            // multiple of U (host guarantees)
    const uint64_t n_spans = (n_chunks + span - 1) / span;
    const uint8_t *lane_text = P.text + base0 + (uint64_t)lane * 16;

    uint64_t sp = wave_id;
    for (;;) { // (span scheduling as in seed_filter_kernel)
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
        const uint64_t span_base = base0 + c_begin * 1024;
        span_open(P, lds, lane, span_base);
        uint32_t carry_in = 0;
        {
            uint32_t nm = 0;
            if (span_base >= 16 && lane == 0)
                carry_in = pack16_sig<4>(load_text16(P.text, span_base - 16, P.hi), nm);
            carry_in = __builtin_amdgcn_readfirstlane(carry_in);
        }
        uint64_t ch = c_begin;
        const uint64_t whole_end = c_end < n_whole ? c_end : n_whole;
        const uint64_t fast_end = c_begin + (whole_end > c_begin ? (whole_end - c_begin) / U * U : 0);
        if (ch < fast_end) {
            uint4 nxt[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                nxt[u] = load16_stream<true>(lane_text + (ch + u) * 1024);
            for (; ch < fast_end; ch += U) {
                uint4 cur[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    cur[u] = nxt[u];
                const bool more = ch + U < fast_end;
                const uint64_t pf = more ? ch + U : ch;
                const uint64_t ustride = more ? 1024 : 0;
#pragma unroll
                for (int u = 0; u < U; ++u)
                    nxt[u] = load16_stream<true>(lane_text + pf * 1024 + (uint64_t)u * ustride);
                __builtin_amdgcn_sched_barrier(0); // the prefetch stays ahead of the group's work
                dense_group<U, NP, S, KM>(P, cur, base0 + ch * 1024, span_base, carry_in, lane, lds, Q, qn);
            }
        }
        for (; ch < c_end; ++ch) { // ragged end
            uint4 one[1];
            one[0] = load_text16(P.text, base0 + ch * 1024 + (uint64_t)lane * 16, P.hi);
            dense_group<1, NP, S, KM>(P, one, base0 + ch * 1024, span_base, carry_in, lane, lds, Q, qn);
        }
        dense_drain(P, Q, qn, true, span_base, lane, lds); // (offsets in the queue are relative to this span)
        sp += n_waves;
    }
    surv_close(P, cand_chunk_of(P, lds), lane);
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
        sp += n_waves;
    }
    surv_close(P, cand_chunk_of(P, lds), lane);
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
    const band_rec *bands;
    const unsigned long long *counters; // [band_counter] = entries of the band list
    uint32_t band_counter;              // 3: the list resolve_kernel wrote; 10: the list band_select_kernel kept
    uint32_t preselected;               // 1: the list holds only bands with enough seed hits, their table slots reset
    uint64_t band_cap;
    ulonglong2 *band_tab;               // band table (slots are given back as the bands are consumed), see band_value()
    uint32_t table_mask, band_bits;     // ... its size - 1 and key layout (band_runs_kernel looks neighbours up)
    uint32_t runs;                      // 1: band_runs_kernel has compacted the heads of runs; the list holds heads only
    const uint8_t *surplus; // per needle: seed hits a band needs (seeds - k); nullptr = 1 for every needle
    uint32_t Bw;            // diagonals per band
    uint32_t overlap;       // 1: bands extend k + 1 diagonals into the next one (sets with surplus seeds)
    uint32_t max_m;         // diagonals are counted from max_m before the haystack's first symbol
    const uint32_t *peq32; // the brute table: [group][sigma+1][nw_table][64], needles top-aligned
    uint32_t sigma;        // alphabet size; LDS holds sigma+1 rows per thread (row sigma = no match)
    uint32_t nw_table;     // words per needle in that table
    uint32_t max_k;        // largest k of the set: 2*max_k + 1 end-position slots per candidate
    const int32_t *m;      // per pattern
    const int32_t *k;
    uint32_t report_begin; // 1: exact matchers report begin = end - m
    uint32_t max_span;     // diagonals of a band beyond the first: Bw - 1 (+ max_k + 1 with overlap)
    uint32_t wave_text;    // wave-per-candidate kernel: bytes of LDS per group for the candidate's text window
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

// ---------------------------------------------------------------------------------------------------
// resolve: survivors -> (needle, offset) through the exact key directory -> whole-seed check, piece count -> diagonal bands
// ---------------------------------------------------------------------------------------------------
// One lane per survivor, a full grid: the L2 round trips of the table probes overlap across thousands of waves instead of
// stalling the streaming kernel.
//   * A survivor says: the key window at needle offset x matches the text at t.  The pigeonhole argument needs a seed
//     that occurs in the text UNCHANGED, and such a seed yields a survivor of its own -- so a (survivor, entry) pair whose
//     whole seed does not match exactly is dropped without losing an occurrence.  On long texts most pairs are chance
//     matches of the key (C4: 800 000 of 1 140 000; C3: 16 000 of 20 000); they fail after ~1.3 symbols.
//   * What is left is counted into a hash table keyed (needle, haystack, diagonal band); the first arrival of a band
//     appends it to the band list.  A band is verified ONCE, over every end position its diagonals can produce -- the 17
//     sampled windows of one needle inside one 136-base microsatellite are one or two verifications, not 17; the 65
//     seeds of a |P| = 1024, k = 64 occurrence one, not 65.
//   * Sets whose needles carry k + 2 seeds (k >= kMergeMinK): an occurrence keeps >= 2 seeds intact, on diagonals <= k
//     apart; bands overlap by k + 1 diagonals so that those seeds always share a band, and a band is verified only if it
//     collected as many seed hits as the needle has surplus seeds.
struct resolve_params
{
    const survivor *surv;
    unsigned long long *counters; // [1] survivor slots drawn (this pass), [3] band slots drawn, [5] candidates (stat),
                                  // [8] survivors over all passes, [9] largest survivor demand of a pass, [2] overflow
    uint64_t surv_cap;
    const pass_entry *passes; // key directories, one per pass
    const uint4 *entries;     // {val = needle << 11 | offset, seed signature, range code, key}: a key's entries side by side
    uint32_t key_len;
    uint32_t flank_check;     // 1: dna4 set without surplus seeds: seed signatures are checked against the packed text
    uint32_t pieces_check;    // 1: ... and the piece count (see pieces_visit)
    const uint8_t *text;
    uint64_t text_alloc;            // readable bytes from text
    const uint8_t *needle_ranks;    // the needles' symbols back to back, padded (nullptr: no whole-seed check)
    const uint32_t *needle_offsets; // start of every needle in needle_ranks
    const uint16_t *seed_q;         // seed length of every needle
    const int32_t *m, *k;
    uint64_t hay_begin, hay_end;    // unsegmented scans: the haystack
    const uint64_t *seg_offsets;
    uint64_t n_segments;
    uint32_t Bw, overlap, max_m;
    uint32_t band_bits;             // key = pattern << 43 | segment << band_bits | band
    ulonglong2 *band_tab;           // {key, value} per slot, see band_value()
    uint32_t table_mask;
    band_rec *bands;
    uint64_t band_cap;
    const uint32_t *needle_pk;      // dna4 sets: the needles 2 bits per symbol, 16 symbols per word (pack16's bit order)
    const uint32_t *pk_offsets;     // first word of every needle in needle_pk
    // Exact sets whose needles are their own (single) seed: a pair that passed the whole-seed check IS an occurrence and
    // is reported from here -- no band, no verification launch (C2: Shift-Or |P| = 32).
    uint32_t exact_hits;
    uint32_t report_begin;          // 1: report begin = end - |P| (the exact matchers)
    uint64_t scan_begin, scan_end;  // owned: last symbol index in [scan_begin, scan_end)
    uint64_t pos_offset;
    const uint32_t *seg_owned;
    unsigned long long *seen;
    uint32_t seen_mask;
    spm_hit *hits;
    unsigned long long *hit_counter, *overflow;
    uint64_t hit_cap;
    // The band table is shared by the scans of a context and must be EMPTY when a scan starts (a slot left behind by an
    // earlier scan makes this one take a band for "already listed" and never verify it).  A scan whose band list or table
    // overflows leaves such slots; the host empties the table before the next scan -- unless that scan was launched before
    // the host looked (deferred completion, SPM_SCAN_DEFER).  So the overflow is also recorded HERE, on the device, where it
    // sticks until the host has emptied the table: a scan that finds it set declares itself void.
    uint32_t *table_poison;
    uint32_t debug_stage;   // diagnostics (SPM_HIP_RESOLVE_DEBUG): cut the kernel short after stage 1..4 to time the stages (wrong results)
};

// Band table slot: .x = key (kBandEmpty = all ones: free), .y = value kept so that a free slot is ALL ONES (one memset
// clears the table, one 16-byte store gives a slot back): the complement of the mask of diagonals hit, or the number of
// seed hits minus one (mod 2^64).  Key and value share a slot because a table of 10^6..10^7 slots lives in HBM and a
// random access there costs a DRAM row, not bytes: the insert's compare-and-swap and the update of the value, and later
// the consumer's read and reset, touch one line each instead of two.
__device__ __forceinline__ unsigned long long band_value(unsigned long long stored, bool counting)
{
    return counting ? (unsigned long long)(uint32_t)(stored + 1ull) : ~stored;
}
__device__ __forceinline__ void band_release(ulonglong2 *tab, uint32_t slot)
{
    tab[slot] = make_ulonglong2(kBandEmpty, ~0ull);
}

// 16 bytes from an arbitrary address with ONE load: gfx950 under amdhsa runs with unaligned access enabled, so an
// align-1 copy of 16 bytes is a single global_load_dwordx4.  (Byte loads would cost the memory pipe 16 instructions, each
// as expensive as this one; the checks below run for 10^7..10^8 (survivor, entry) pairs on a repeat-rich text.)
__device__ __forceinline__ uint4 load_bytes16(const uint8_t *p)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

// the same near the end of a buffer: bytes at or beyond `limit` read as 0xFF (never equal to a symbol)
__device__ __forceinline__ uint4 load_bytes16_guarded(const uint8_t *base, uint64_t idx, uint64_t limit)
{
    if (idx + 16 <= limit)
        return load_bytes16(base + idx);
    uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    for (int b = 0; b < 16; ++b)
        if (idx + b < limit)
            w[b >> 2] = (w[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | ((uint32_t)base[idx + b] << (8 * (b & 3)));
    return make_uint4(w[0], w[1], w[2], w[3]);
}

__device__ __forceinline__ uint32_t diff16(const uint4 a, const uint4 b, uint32_t n) // n <= 16 leading bytes compared
{
    const uint32_t d[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
    uint32_t r = 0;
#pragma unroll
    for (uint32_t i = 0; i < 4; ++i) {
        const uint32_t have = n > 4 * i ? n - 4 * i : 0; // bytes of this dword that count
        const uint32_t mask = have >= 4 ? 0xFFFFFFFFu : (have == 0 ? 0u : (1u << (8 * have)) - 1);
        r |= d[i] & mask;
    }
    return r;
}

// r = offset of the key window inside its seed (the seeds of a needle with an N do not sit at multiples of q)
// q_whole: the seed's length when the entry spells it (kRngWhole: key + signature are the whole seed; the pieces of a
// dense pass differ in length), 0: all seeds of the needle are seed_q[needle] long
__device__ __forceinline__ bool seed_intact(const resolve_params &P, uint64_t t, uint32_t val, uint32_t r, uint32_t q_whole,
                                            int64_t hay_b, int64_t hay_e)
{
    if (!P.needle_ranks)
        return true;
    const uint32_t pat = val >> 11, x = val & 0x7FF;
    struct
    {
        uint32_t q;
    } sp{q_whole ? q_whole : P.seed_q[pat]};
    const uint32_t o = x - r; // start of the seed inside the needle
    const int64_t ts = (int64_t)t - (int64_t)r; // where the seed would start in the text
    if (ts < hay_b || ts + (int64_t)sp.q > hay_e)
        return false; // it would stick out of the haystack
    const uint8_t *nd = P.needle_ranks + P.needle_offsets[pat] + o; // (the needle buffer is padded: no guard)
    // (the key window itself already matched; comparing it again costs less than skipping it.)  16 symbols per round: a
    // chance match of the key fails in the first
    for (uint32_t i = 0; i < sp.q; i += 16) {
        const uint32_t n = sp.q - i < 16 ? sp.q - i : 16;
        if (diff16(load_bytes16_guarded(P.text, (uint64_t)ts + i, P.text_alloc), load_bytes16(nd + i), n))
            return false;
    }
    return true;
}

// Piece count (dna4 sets without surplus seeds, i.e. k < kMergeMinK) -- the q-gram lemma on aligned pieces.  The seed
// sits unchanged on diagonal d (needle position y <-> text index d + y).  Cut the needle into pieces of 8 symbols: an
// occurrence with <= k edits leaves all but k of them unchanged, each displaced by at most k from its place on the
// diagonal (the indels between it and the seed).  So if more than k pieces are found nowhere within +-k of their place,
// no occurrence contains this seed here -- and an occurrence always has an intact seed whose pieces pass.  What this
// removes: the seed hits of needles that merely END in a microsatellite or a poly-A run, in every stretch of the same
// unit (their unique part fails after two or three blocks); each would have cost a band and a verification of ~170
// columns.  Needles that ARE a repeat pass, as they must: they occur there.  A pair whose needle would stick out of the
// haystack is not checked.
//
// The count runs two rounds (32 needle symbols, four pieces) per visit, starting at the end of the needle that is far
// from the seed -- the seed sits in the part the needle shares with the text, what differs is at the other end -- and a
// pair that is neither rejected nor through goes back into the wave's queue: on a repeat-rich text 85 % of the pairs are
// rejected in the first visit, and the other lanes of the wave do not wait for the 15 % that need all the rounds.
// Progress lives in the pair's `rng` word: rounds done (bits 21..27), pieces missing (28..30).
constexpr uint32_t kPieceRoundsShift = 21, kPieceMissingShift = 28;

__device__ __forceinline__ bool pieces_apply(const resolve_params &P, uint64_t t, uint32_t val, uint32_t m, uint32_t k, int64_t hay_b,
                                             int64_t hay_e)
{
    if (!P.pieces_check || k == 0 || k > 7 || m < 32)
        return false;
    const int64_t d = (int64_t)t - (int64_t)(val & 0x7FF);
    return d - (int64_t)k >= hay_b && d + (int64_t)(m + k) <= hay_e;
}

// one visit; returns 0: rejected, 1: more rounds to go (rng updated), 2: through
__device__ __forceinline__ uint32_t pieces_visit(const resolve_params &P, uint64_t t, uint32_t val, uint32_t &rng, uint32_t m, uint32_t k)
{
    const uint32_t pat = val >> 11, x = val & 0x7FF;
    const uint32_t n_rounds = m >> 4; // two pieces per 16 needle symbols
    uint32_t done = (rng >> kPieceRoundsShift) & 0x7F, missing = (rng >> kPieceMissingShift) & 7;
    const bool ascending = 2 * x >= m; // seed in the right half: start at the left end
    const uint32_t todo = n_rounds - done < 2 ? n_rounds - done : 2;
    // rounds `done`, `done + 1` in visiting order are the needle words r_lo, r_lo + 1 (if todo == 2)
    const uint32_t r_lo = ascending ? done : n_rounds - done - todo;
    const uint32_t *pk = P.needle_pk + P.pk_offsets[pat] + r_lo;
    const uint32_t nw0 = pk[0], nw1 = todo == 2 ? pk[1] : 0;
    // text [d + 16 r_lo - k, ... + 48): 16 + 2k <= 30 symbols per round, the two rounds share the middle load
    const uint64_t w0 = (uint64_t)((int64_t)t - (int64_t)x + (int64_t)(16 * r_lo) - (int64_t)k);
    const uint32_t c0 = pack16(load_bytes16_guarded(P.text, w0, P.text_alloc));
    const uint32_t c1 = pack16(load_bytes16_guarded(P.text, w0 + 16, P.text_alloc));
    const uint32_t c2 = todo == 2 ? pack16(load_bytes16_guarded(P.text, w0 + 32, P.text_alloc)) : 0;
#pragma unroll
    for (uint32_t rr = 0; rr < 2; ++rr) {
        if (rr < todo) {
            const uint32_t two = rr ? nw1 : nw0;
            const uint64_t win = rr ? ((uint64_t)c1 | ((uint64_t)c2 << 32)) : ((uint64_t)c0 | ((uint64_t)c1 << 32));
#pragma unroll
            for (uint32_t j = 0; j < 2; ++j) {
                const uint32_t piece = (two >> (16 * j)) & 0xFFFFu;
                bool found = false;
                for (uint32_t sh = 0; sh <= 2 * k; ++sh) // displacement sh - k
                    found = found || (((uint32_t)(win >> (2 * (8 * j + sh))) & 0xFFFFu) == piece);
                missing += found ? 0u : 1u;
            }
        }
    }
    if (missing > k)
        return 0;
    done += todo;
    if (done >= n_rounds)
        return 2;
    rng = (rng & ~((0x7Fu << kPieceRoundsShift) | (7u << kPieceMissingShift))) | (done << kPieceRoundsShift) |
          (missing << kPieceMissingShift);
    return 1;
}

// The 64 text symbols around a survivor's key window, [t - 16, t + 48), 2 bits each: every entry of the key is checked
// against them in registers ("does the REST of the seed match too?") instead of with loads per entry -- 73 % of the
// (survivor, entry) pairs of a repeat-rich text die here.
struct text64
{
    uint64_t lo, hi; // symbols 0..31, 32..63
    bool ok;         // false: the window is not wholly inside the haystack (no signature check then)
};

__device__ __forceinline__ uint32_t syms16(const text64 &W, uint32_t first) // 16 symbols from index `first` (<= 48)
{
    if (first >= 32)
        return (uint32_t)(W.hi >> (2 * (first - 32)));
    return (uint32_t)(W.lo >> (2 * first)) | (first ? (uint32_t)(W.hi << (2 * (32 - first)) ) : 0u);
}

// the first n (<= 16) symbols of the seed's rest -- the (up to 16) symbols right before the key window, r of them if the
// window starts r < 16 symbols into its seed, then those after it -- against the text
__device__ __forceinline__ bool seed_sig_ok(const text64 &W, uint32_t sig, uint32_t r, uint32_t n, uint32_t H)
{
    if (n == 0)
        return true;
    const uint32_t rb = r < 16 ? r : 16; // symbols of the seed before the window that the signature starts with
    const uint32_t nl = rb < n ? rb : n; // ... of which it holds the first nl: text [t - rb, t - rb + nl)
    uint32_t got = nl ? syms16(W, 16 - rb) : 0u;
    if (nl < 16) {
        got &= (1u << (2 * nl)) - 1;
        got |= syms16(W, 16 + H) << (2 * nl);
    }
    const uint32_t mask = n >= 16 ? 0xFFFFFFFFu : ((1u << (2 * n)) - 1);
    return ((got ^ sig) & mask) == 0;
}

// Work waits in LDS until a wave has 64 items of a kind: a key shared by twenty needles, a pair that needs all its
// piece rounds, a seed hit that counts into four bands would otherwise keep one lane busy while 63 idle -- and the band
// table's atomics, whose round trips to HBM are the longest latency in this kernel, would be issued a few lanes at a time.
constexpr uint32_t kPairCap = 128;
struct pair_queue // per wave: (survivor, entry) pairs on their way through the checks
{
    uint32_t t_lo[kPairCap], t_hi[kPairCap], val[kPairCap], rng[kPairCap], seg[kPairCap];
};
struct band_queue // per wave: (band key, diagonals hit) on their way into the band table
{
    uint32_t key_lo[kPairCap], key_hi[kPairCap], run_lo[kPairCap], run_hi[kPairCap];
    uint32_t elect[128]; // insert_bands: which lane of the batch speaks for a band
};

struct band_chunk // this wave's chunk of the band list (wave-uniform registers)
{
    uint64_t base;
    uint32_t used, size, draws;
};
// The first draws of a wave take exactly what it needs, so a scan with few bands (the usual case; and the long-needle
// sets whose wave-per-band verification wants every wave busy) leaves a dense list; later draws take growing chunks.
constexpr uint32_t kDenseDraws = 4;

// n (<= 64) queued band hits, one per lane: into the band table; the first arrival of a band appends it to the list
__device__ __forceinline__ void insert_bands(const resolve_params &R, band_queue &B, uint32_t first, uint32_t n,
                                             uint32_t lane, band_chunk &C)
{
    bool act = lane < n;
    unsigned long long bkey = 0, run = 0;
    if (act)
        bkey = ((unsigned long long)B.key_hi[first + lane] << 32) | B.key_lo[first + lane];
    // Hits of the same band inside the batch (the sampled windows of one needle in one repeat stretch) go to the table
    // as ONE: every lane names itself in a small LDS table under its band's hash, whoever is left standing there speaks
    // for the lanes with the same band, which OR their diagonals into its queue entry.  (Worth 2 % on a text with many
    // needles inside repeats; the table's compare-and-swap round trips are the longest waits of this kernel.)
    if (!R.overlap) {
        const uint32_t h = (uint32_t)mix64(bkey) & 127u;
        if (act)
            B.elect[h] = lane;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t w = act ? B.elect[h] : lane;
        const unsigned long long wkey = ((unsigned long long)B.key_hi[first + w] << 32) | B.key_lo[first + w];
        const bool follower = act && w != lane && wkey == bkey;
        if (follower) {
            atomicOr(&B.run_lo[first + w], B.run_lo[first + lane]);
            atomicOr(&B.run_hi[first + w], B.run_hi[first + lane]);
            act = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (act)
        run = ((unsigned long long)B.run_hi[first + lane] << 32) | B.run_lo[first + lane];
    bool claimed = false;
    uint32_t s2 = (uint32_t)mix64(bkey) & R.table_mask;
    if (act) {
        bool placed = false;
        for (uint32_t tries = 0; tries < 8192 && !placed; ++tries) {
            // (a table that has overflowed is lost -- the host repeats the scan with a larger one: do not walk a full table
            // 8192 slots per band, which cost 5 s on a text with 5 % repeats whose first attempt was sized for 1 %)
            if ((tries & 31u) == 31u && __hip_atomic_load(&R.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                break;
            const unsigned long long o = atomicCAS(&R.band_tab[s2].x, kBandEmpty, bkey);
            if (o == kBandEmpty) {
                claimed = true;
                placed = true;
            } else if (o == bkey) {
                placed = true;
            } else {
                s2 = (s2 + 1) & R.table_mask;
            }
        }
        if (!placed) {
            atomicAdd(&R.counters[2], 1ull); // table full: the host starts over with more room
            *R.table_poison = 1u;
        }
        else if (R.overlap)
            atomicAdd(&R.band_tab[s2].y, 1ull);
        else
            atomicAnd(&R.band_tab[s2].y, ~run);
    }
    const uint64_t mm = __ballot(claimed);
    if (mm == 0)
        return;
    const uint32_t nn = __popcll(mm);
    if (C.used + nn > C.size) { // close this chunk (unused tail invalid), draw the next
        for (uint32_t q = C.used + lane; q < C.size; q += 64)
            if (C.base + q < R.band_cap)
                R.bands[C.base + q].val = kBandInvalid;
        uint32_t next = C.size * 2 < kChunkMin ? kChunkMin : (C.size * 2 > kChunkMax ? kChunkMax : C.size * 2);
        next = next < nn ? nn : next;
        if (C.draws < kDenseDraws) {
            next = nn;
            ++C.draws;
        }
        unsigned long long b = 0;
        if (lane == 0)
            b = atomicAdd(&R.counters[3], (unsigned long long)next);
        C.base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b);
        C.used = 0;
        C.size = next;
    }
    if (claimed) {
        const uint64_t idx = C.base + C.used + __popcll(mm & ((1ull << lane) - 1));
        if (idx < R.band_cap) {
            band_rec br;
            br.slot = s2;
            br.val = (uint32_t)(bkey >> 43) << 11;
            br.seg = (uint32_t)((bkey & ((1ull << 43) - 1)) >> R.band_bits);
            br.band = (uint32_t)(bkey & ((1ull << R.band_bits) - 1));
            R.bands[idx] = br;
        } else {
            atomicAdd(&R.counters[2], 1ull); // band list full: the slot this lane claimed is never given back
            *R.table_poison = 1u;
        }
    }
    C.used += nn;
}

// Reserve hit slots for a whole wave with ONE atomic: lane l gets `mine` consecutive slots starting at the returned index.
// (One atomic per end-position slot and wave, as the brute-force kernels do it, serialises on the counter's address at
// ~100/us: a repeat-rich text reports millions of hits.)  Call with the wave converged.
__device__ __forceinline__ unsigned long long wave_reserve_hits(unsigned long long *counter, uint32_t mine)
{
    const uint32_t lane = threadIdx.x & 63;
    uint32_t incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o)
            incl += up;
    }
    const uint32_t total = (uint32_t)__shfl((int)incl, 63);
    unsigned long long base = 0;
    if (total != 0) {
        if (lane == 0)
            base = atomicAdd(counter, (unsigned long long)total);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
    }
    return base + (incl - mine);
}

// one key per reported hit in the scan's dedupe set; true if this is the first report of (pattern, end)
__device__ __forceinline__ bool seen_insert_raw(unsigned long long *seen, uint32_t seen_mask, unsigned long long *overflow,
                                                uint32_t pat, int64_t e)
{
    const unsigned long long key = ((unsigned long long)pat << 40) | (unsigned long long)e;
    uint32_t slot = (uint32_t)(mix64(key)) & seen_mask;
    // the set holds at most one key per reportable hit at load <= 1/2; a probe sequence this long means more hits than
    // it was sized for: flag it, the host starts over with a larger one
    for (uint32_t tries = 0; tries < 64; ++tries) {
        // (a set that has overflowed once is lost -- the host repeats the scan: do not grind through 64 compare-and-swaps
        // per hit of a full table, 5 s for the 46 M hits of a text with 5 % repeats)
        if (tries == 4 && __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            return false;
        const unsigned long long old = atomicCAS(&seen[slot], ~0ull, key);
        if (old == ~0ull)
            return true;
        if (old == key)
            return false;
        slot = (slot + 1) & seen_mask;
    }
    atomicAdd(overflow, 1ull);
    return false;
}

struct resolve_wave // what a wave of resolve_kernel carries (wave-uniform)
{
    pair_queue *Q;
    band_queue *B;
    uint32_t qn, bn;
    band_chunk C;
    uint32_t n_cand;
};

__device__ __forceinline__ void queue_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// n (<= 64) pairs from the top of the queue, one per lane: one visit of the checks; pairs with rounds to go return to the
// queue, pairs that are through become band hits
__device__ __forceinline__ void check_pairs(const resolve_params &R, resolve_wave &S, uint32_t n, uint32_t lane)
{
    pair_queue &Q = *S.Q;
    band_queue &B = *S.B;
    S.qn -= n;
    const uint32_t first = S.qn;
    bool live = lane < n;
    uint64_t t = 0, seg = 0;
    uint32_t val = 0, rng = 0;
    int64_t sb = (int64_t)R.hay_begin, se = (int64_t)R.hay_end;
    if (live) {
        const uint32_t i = first + lane;
        t = ((uint64_t)Q.t_hi[i] << 32) | Q.t_lo[i];
        val = Q.val[i];
        rng = Q.rng[i];
        seg = Q.seg[i];
        if (R.seg_offsets) {
            sb = (int64_t)R.seg_offsets[seg];
            se = (int64_t)R.seg_offsets[seg + 1];
        }
    }
    queue_sync(); // (every lane holds its pair before anything is written back)
    bool through = live;
    const uint32_t pat = val >> 11;
    if (live && !(rng & kRngRun)) {
        if (!(rng & kSeedChecked)) {
            live = seed_intact(R, t, val, (rng >> 16) & 0x1F, (rng & kRngWhole) ? R.key_len + ((rng >> 5) & 0x1F) : 0u, sb, se);
            rng |= kSeedChecked;
        }
        if (R.debug_stage == 3)
            live = false;
        through = live;
        if (live && R.pieces_check) {
            const uint32_t m = (uint32_t)R.m[pat], k = (uint32_t)R.k[pat];
            if (pieces_apply(R, t, val, m, k, sb, se)) {
                const uint32_t v = pieces_visit(R, t, val, rng, m, k);
                live = v != 0;
                through = v == 2;
            }
        }
    }
    if (R.debug_stage == 4 && through)
        live = false;
    // back into the queue: alive, not through
    {
        const bool back = live && !through;
        const uint64_t mm = __ballot(back);
        if (back) {
            const uint32_t q = S.qn + __popcll(mm & ((1ull << lane) - 1));
            Q.t_lo[q] = (uint32_t)t;
            Q.t_hi[q] = (uint32_t)(t >> 32);
            Q.val[q] = val;
            Q.rng[q] = rng;
            Q.seg[q] = (uint32_t)seg;
        }
        S.qn += __popcll(mm);
    }
    // bands this pair counts into: those holding a diagonal of [d_lo, d_hi]; with overlapping bands also the one
    // before, if d_lo still lies in its k-wide extension
    int64_t b_cur = 0, b_last = -1, d_lo = 0, d_hi = 0;
    if (R.exact_hits) { // (wave-uniform; such sets have no merged run entries)
        bool fresh = false;
        int64_t e = 0;
        if (live && through) {
            ++S.n_cand;
            e = (int64_t)t - (int64_t)(val & 0x7FF) + (int64_t)R.m[pat]; // exclusive end of the occurrence
            int64_t own_b = (int64_t)R.scan_begin, own_e = (int64_t)R.scan_end;
            if (R.seg_offsets) {
                own_b = R.seg_owned ? sb + (int64_t)R.seg_owned[seg] : sb;
                own_e = se;
            }
            // (R.seen == nullptr: every occurrence has exactly one sampled window and one entry, so nothing is reported
            // twice -- unless spans give up and the brute-force kernel re-scans them: then the host runs the scan again
            // with the dedupe set)
            fresh = e >= own_b + 1 && e <= own_e && (!R.seen || seen_insert_raw(R.seen, R.seen_mask, R.overflow, pat, e));
        }
        const unsigned long long idx = wave_reserve_hits(R.hit_counter, fresh ? 1u : 0u);
        if (fresh && idx < R.hit_cap) {
            spm_hit h;
            h.pos = (R.report_begin ? (uint64_t)(e - (int64_t)R.m[pat]) : (uint64_t)e) + R.pos_offset;
            h.pattern = pat;
            h.score = 0;
            R.hits[idx] = h;
        }
        queue_sync();
        return;
    }
    if (live && through) {
        ++S.n_cand;
        d_hi = (int64_t)t - (int64_t)(val & 0x7FF) - sb + (int64_t)R.max_m;
        d_lo = d_hi - (int64_t)((rng & kRngRun) ? (rng & 0x7FF) : 0u);
        b_cur = d_lo / (int64_t)R.Bw;
        b_last = d_hi / (int64_t)R.Bw;
        if (R.overlap && b_cur > 0 && d_lo - b_cur * (int64_t)R.Bw <= (int64_t)R.k[pat])
            --b_cur;
        if ((uint64_t)b_last >> R.band_bits) {
            atomicAdd(&R.counters[2], 1ull); // a haystack too long for the key layout: the host falls back
            b_last = b_cur - 1;
        }
    }
    for (;;) {
        const bool have = b_cur <= b_last;
        const uint64_t mm = __ballot(have);
        if (mm == 0)
            break;
        if (have) {
            const unsigned long long bkey =
                ((unsigned long long)pat << 43) | ((unsigned long long)seg << R.band_bits) | (unsigned long long)b_cur;
            // which diagonals of the band (Bw <= 64) are hit: the verification covers just those
            const int64_t b0 = b_cur * (int64_t)R.Bw;
            const uint32_t o_lo = (uint32_t)(d_lo > b0 ? d_lo - b0 : 0);
            const uint32_t o_hi = (uint32_t)(d_hi < b0 + (int64_t)R.Bw - 1 ? d_hi - b0 : (int64_t)R.Bw - 1);
            const unsigned long long run = (o_hi - o_lo >= 63 ? ~0ull : ((1ull << (o_hi - o_lo + 1)) - 1)) << o_lo;
            const uint32_t q = S.bn + __popcll(mm & ((1ull << lane) - 1));
            B.key_lo[q] = (uint32_t)bkey;
            B.key_hi[q] = (uint32_t)(bkey >> 32);
            B.run_lo[q] = (uint32_t)run;
            B.run_hi[q] = (uint32_t)(run >> 32);
        }
        S.bn += __popcll(mm);
        queue_sync();
        if (S.bn >= 64) {
            S.bn -= 64;
            insert_bands(R, B, S.bn, 64, lane, S.C);
            queue_sync();
        }
        ++b_cur;
    }
    queue_sync();
}

__global__ __launch_bounds__(256) void resolve_kernel(const resolve_params R)
{
    __shared__ pair_queue pair_queues[4];
    __shared__ band_queue band_queues[4];
    unsigned long long n = R.counters[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        atomicAdd(&R.counters[8], n);
        atomicMax(&R.counters[9], n);
    }
    if (n > R.surv_cap)
        n = R.surv_cap;
    if (!R.exact_hits && __hip_atomic_load(R.table_poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        // an earlier scan left slots in the band table and the host has not emptied it yet: this scan is void
        if (blockIdx.x == 0 && threadIdx.x == 0)
            atomicAdd(&R.counters[2], 1ull);
        return;
    }
    const uint32_t lane = threadIdx.x & 63;
    resolve_wave S;
    S.Q = &pair_queues[threadIdx.x >> 6];
    S.B = &band_queues[threadIdx.x >> 6];
    S.qn = 0;
    S.bn = 0;
    S.C.base = 0;
    S.C.used = 0;
    S.C.size = 0;
    S.C.draws = 0;
    S.n_cand = 0;
    pair_queue &Q = *S.Q;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride; // wave-uniform trip count: the queue and the appends are wave-collective
    for (uint64_t r = 0; r < rounds; ++r) {
        // a list or table of this attempt has overflowed: its results are void (the host starts over), stop feeding it
        if ((r & 7u) == 7u &&
            __builtin_amdgcn_readfirstlane((uint32_t)(__hip_atomic_load(&R.counters[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)))
            break;
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool probing = false;
        uint32_t key = 0;
        uint64_t t = 0, seg = 0;
        pass_entry T = R.passes[0];
        if (i < n) {
            const survivor sv = R.surv[i];
            if (sv.t_hi != kSurvInvalid) {
                probing = true;
                key = sv.key;
                t = ((uint64_t)sv.t_hi << 32) | sv.t_lo;
                if (sv.pad != 0)
                    T = R.passes[sv.pad];
            }
        }
        if (probing && R.seg_offsets) {
            uint64_t lo = 0, hi = R.n_segments; // invariant: seg_offsets[lo] <= t < seg_offsets[hi]
            while (hi - lo > 1) {
                const uint64_t mid = (lo + hi) >> 1;
                if (R.seg_offsets[mid] <= t)
                    lo = mid;
                else
                    hi = mid;
            }
            seg = lo;
            if ((int64_t)t + (int64_t)R.key_len > (int64_t)R.seg_offsets[lo + 1])
                probing = false; // the key window straddles two haystacks
        }
        // the text around the key window, once per survivor
        text64 W;
        W.lo = 0;
        W.hi = 0;
        W.ok = false;
        if (probing && R.flank_check) { // (dna4 sets: 2-bit codes)
            int64_t sb = (int64_t)R.hay_begin, se = (int64_t)R.hay_end;
            if (R.seg_offsets) {
                sb = (int64_t)R.seg_offsets[seg];
                se = (int64_t)R.seg_offsets[seg + 1];
            }
            if ((int64_t)t - 16 >= sb && (int64_t)t + 48 <= se && t + 48 <= R.text_alloc) {
                const uint8_t *p = R.text + (t - 16);
                W.lo = (uint64_t)pack16(load_bytes16(p)) | ((uint64_t)pack16(load_bytes16(p + 16)) << 32);
                W.hi = (uint64_t)pack16(load_bytes16(p + 32)) | ((uint64_t)pack16(load_bytes16(p + 48)) << 32);
                W.ok = true;
            }
        }
        // the key's entries: one short directory probe per survivor ...
        uint32_t first = 0, cnt = 0;
        {
            uint32_t slot = ht_hash(key) & T.ht_mask;
            while (__ballot(probing) != 0) {
                if (probing) {
                    const uint4 d = T.ht[slot];
                    if (d.z == 0) {
                        probing = false; // (a level-1 false positive: no such key)
                    } else if (d.x == key) {
                        first = d.y;
                        cnt = d.z;
                        probing = false;
                    } else {
                        slot = (slot + 1) & T.ht_mask;
                    }
                }
            }
        }
        if (R.debug_stage == 1)
            cnt = 0;
        // ... then the (survivor, entry) pairs of the whole wave are dealt to its lanes, 64 at a time: pair i belongs to the
        // lane whose inclusive prefix sum of `cnt` is the first one above i
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
            if (lane >= (uint32_t)o)
                incl += up;
        }
        const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)incl, 63));
        for (uint32_t p0 = 0; p0 < total; p0 += 64) {
            const uint32_t i = p0 + lane;
            bool have = i < total;
            uint32_t owner = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1) {
                const uint32_t v = (uint32_t)__shfl((int)incl, (int)(owner + step - 1));
                if (v <= i)
                    owner += step;
            }
            owner = owner > 63 ? 63 : owner;
            const uint32_t o_incl = (uint32_t)__shfl((int)incl, (int)owner), o_cnt = (uint32_t)__shfl((int)cnt, (int)owner);
            const uint32_t o_first = (uint32_t)__shfl((int)first, (int)owner);
            const uint64_t o_t = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(t >> 32), (int)owner) << 32) |
                                 (uint32_t)__shfl((int)(uint32_t)t, (int)owner);
            const uint32_t o_seg = (uint32_t)__shfl((int)(uint32_t)seg, (int)owner);
            text64 OW;
            OW.lo = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(W.lo >> 32), (int)owner) << 32) |
                    (uint32_t)__shfl((int)(uint32_t)W.lo, (int)owner);
            OW.hi = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(W.hi >> 32), (int)owner) << 32) |
                    (uint32_t)__shfl((int)(uint32_t)W.hi, (int)owner);
            OW.ok = __shfl((int)W.ok, (int)owner) != 0;
            uint32_t val = 0, rng = 0;
            if (have) {
                const uint4 e = R.entries[o_first + (i - (o_incl - o_cnt))];
                val = e.x;
                rng = e.z;
                if (!(rng & kRngRun)) {
                    const uint32_t r0 = rng & 0x1F, ns = (rng >> 5) & 0x1F;
                    const bool whole = (rng & kRngWhole) != 0;
                    // (queued with the pair: where the key window sits in its seed, and the seed's length if the entry knows it)
                    rng = (r0 << 16) | (whole ? (ns << 5) | kRngWhole : 0u);
                    if (OW.ok) { // does the rest of the seed match?  (registers only)
                        if (!seed_sig_ok(OW, e.y, r0, ns, R.key_len))
                            have = false;
                        else if (whole)
                            rng |= kSeedChecked;
                    }
                }
            }
            if (R.debug_stage == 2)
                have = false;
            // queue the pairs that are left; 64 waiting pairs are resolved at once
            const uint64_t mm = __ballot(have);
            if (mm != 0) {
                if (have) {
                    const uint32_t q = S.qn + __popcll(mm & ((1ull << lane) - 1));
                    Q.t_lo[q] = (uint32_t)o_t;
                    Q.t_hi[q] = (uint32_t)(o_t >> 32);
                    Q.val[q] = val;
                    Q.rng[q] = rng;
                    Q.seg[q] = o_seg;
                }
                S.qn += __popcll(mm);
                queue_sync();
                while (S.qn >= 64)
                    check_pairs(R, S, 64, lane);
            }
        }
    }
    while (S.qn != 0)
        check_pairs(R, S, S.qn < 64 ? S.qn : 64, lane);
    if (S.bn != 0)
        insert_bands(R, *S.B, 0, S.bn, lane, S.C);
    for (uint32_t q = S.C.used + lane; q < S.C.size; q += 64)
        if (S.C.base + q < R.band_cap)
            R.bands[S.C.base + q].val = kBandInvalid;
    wave_count_add(R.counters + 5, S.n_cand);
}

// one key per reported hit in the scan's dedupe set; true if this is the first report of (pattern, end)
__device__ __forceinline__ bool seen_insert(const verify_params &P, uint32_t pat, int64_t e)
{
    return seen_insert_raw(P.seen, P.seen_mask, P.overflow, pat, e);
}

// Sets with surplus seeds (k >= kMergeMinK): most bands hold a single chance match of a short key and are NOT verified.
// This pass keeps the bands that collected enough seed hits -- a dense list, so the wave-per-band verification (one
// ~0.2 ms serial chain per band) gets one band per wave instead of two on some and none on most -- and gives every table
// slot back.
__global__ __launch_bounds__(256) void band_select_kernel(const verify_params P, band_rec *out, unsigned long long *out_count)
{
    unsigned long long n = P.counters[3];
    if (n > P.band_cap)
        n = P.band_cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n + stride - 1) / stride;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t i = r * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
        bool keep = false;
        band_rec c;
        c.val = kBandInvalid;
        if (i < n) {
            c = P.bands[i];
            if (c.val != kBandInvalid) {
                const uint32_t cnt = (uint32_t)band_value(P.band_tab[c.slot].y, true);
                band_release(P.band_tab, c.slot);
                keep = cnt >= (P.surplus ? (uint32_t)P.surplus[c.val >> 11] : 1u);
            }
        }
        const uint64_t m = __ballot(keep);
        if (m != 0) {
            unsigned long long base = 0;
            const int leader = __ffsll((unsigned long long)m) - 1;
            if ((int)lane == leader)
                base = atomicAdd(out_count, (unsigned long long)__popcll(m));
            base = __shfl(base, leader);
            if (keep)
                out[base + __popcll(m & ((1ull << lane) - 1))] = c; // (<= the entries of the input list)
        }
    }
}

// ---- runs of adjacent bands ------------------------------------------------------------------------------------------
// A repeat stretch of the text that a needle (nearly) matches is hit on a hundred diagonals in a row: three, four bands of
// 32, each of which would be verified on its own -- |P| + k columns of cold start each time for 32 + 2k end positions --
// and every one of its hits would go through the dedupe set (a compare-and-swap in a table of a gigabyte: memory-side
// atomics are what the verification of a repeat-rich text runs out of first).  band_runs_kernel (sets without surplus
// seeds, long band lists only) looks every band's neighbours up in the band table BEFORE the verification gives the slots
// back: a band whose predecessor exists is a FOLLOWER (not verified on its own), a band without one -- or at a multiple of
// kRunMax, so that no lane gets more than kRunMax bands (a wave waits for its longest lane: four bands are 237 columns
// against the 141 of one, and already save 58 % of four cold starts) -- is the HEAD of a run and records how many bands
// follow it and the last hit diagonal of the last one.  The heads are compacted into a list of their own; the verification
// runs through the diagonals of a whole run with one cold start.  In band_rec::val (low 11 bits, 0 when the kernel did not run):
constexpr uint32_t kRunAnalysed = 0x400u; // band_runs_kernel has looked at this band
constexpr uint32_t kRunFollower = 0x200u; // ... and found its predecessor: the head of the run covers it
constexpr uint32_t kRunLenShift = 5, kRunLenMask = 0x3u, kRunHiMask = 0x1Fu; // head: followers (0..3), last hit diagonal (0..31)
// head: the band right before / right behind the run exists too (the run was cut at a multiple of kRunMax): only there can
// another verification report the same (needle, end) -- the seeds of one occurrence lie within k diagonals of each other,
// i.e. in one band or in two adjacent ones, and adjacent bands belong to one run unless the cut falls between them.  So a
// head asks the dedupe set only about the end positions within reach of such a side, and reports the rest as they come.
constexpr uint32_t kRunTouchLeft = 0x80u, kRunTouchRight = 0x100u;
constexpr uint32_t kRunMax = 4;
constexpr uint32_t kHitStage = 192; // hit records a wave of verify_kernel collects before it moves them out

__device__ __forceinline__ uint32_t band_lookup(const verify_params &P, unsigned long long bkey)
{
    uint32_t s2 = (uint32_t)mix64(bkey) & P.table_mask;
    for (uint32_t tries = 0; tries < 8192; ++tries) {
        const unsigned long long o = P.band_tab[s2].x;
        if (o == bkey)
            return s2;
        if (o == kBandEmpty)
            return 0xFFFFFFFFu;
        s2 = (s2 + 1) & P.table_mask;
    }
    return 0xFFFFFFFFu;
}

// One pass over the band list: classify every band (head / follower), write the code into its record, and compact the heads
// into `heads` (a workgroup takes 2048 records at a time and reserves room for its heads with ONE atomic -- a wave-level
// reservation would be 500 000 atomics on one word for a 30 M list).  The verification then runs over `heads` only and
// leaves the table alone; band_release_kernel gives every slot of the original list back afterwards.
__global__ __launch_bounds__(256) void band_runs_kernel(const verify_params P, band_rec *bands, band_rec *heads,
                                                        unsigned long long *head_count)
{
    __shared__ uint32_t wave_tot01[4], wave_tot23[4];
    __shared__ unsigned long long chunk_base;
    unsigned long long n = P.counters[P.band_counter];
    if (n > P.band_cap)
        n = P.band_cap;
    constexpr uint32_t PER = 8;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint64_t c0 = (uint64_t)blockIdx.x * (256 * PER); c0 < n; c0 += (uint64_t)gridDim.x * (256 * PER)) {
        band_rec mine[PER];
        uint32_t head_mask = 0, cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j) {
            const uint64_t i = c0 + (uint64_t)j * 256 + tid;
            mine[j].val = kBandInvalid;
            if (i >= n)
                continue;
            band_rec c = bands[i];
            if (c.val == kBandInvalid)
                continue;
            const unsigned long long base = ((unsigned long long)(c.val >> 11) << 43) | ((unsigned long long)c.seg << P.band_bits);
            uint32_t code = kRunAnalysed;
            const bool pred = c.band > 0 && band_lookup(P, base | (unsigned long long)(c.band - 1)) != 0xFFFFFFFFu;
            if (c.band % kRunMax != 0 && pred) {
                code |= kRunFollower;
            } else {
                uint32_t followers = 0, last = c.slot;
                bool cut = true; // the run ends at a multiple of kRunMax (then the band behind it may exist)
                for (uint32_t r = 1; r < kRunMax && (c.band + r) % kRunMax != 0; ++r) {
                    const uint32_t s2 = band_lookup(P, base | (unsigned long long)(c.band + r));
                    if (s2 == 0xFFFFFFFFu) {
                        cut = false;
                        break;
                    }
                    followers = r;
                    last = s2;
                }
                const bool succ = cut && band_lookup(P, base | (unsigned long long)(c.band + followers + 1)) != 0xFFFFFFFFu;
                const unsigned long long v = band_value(P.band_tab[last].y, false);
                const uint32_t hi = v ? (uint32_t)(63 - __clzll((long long)v)) : 0u;
                code |= (followers << kRunLenShift) | (hi & kRunHiMask) | (pred ? kRunTouchLeft : 0u) | (succ ? kRunTouchRight : 0u);
                head_mask |= 1u << j;
                ++cnt;
            }
            c.val = (c.val & ~0x7FFu) | code;
            mine[j] = c;
        }
        // where this thread's heads go: one reservation per chunk, and inside the chunk the heads are laid out BY RUN LENGTH
        // (class = followers, 0..3): the verification takes 64 consecutive heads per wave and runs as long as its longest
        // run -- 231 columns for four bands against 135 for one --, so waves of one class waste nothing.  Four 16-bit counters
        // (<= 2048 heads per chunk) travel in two words through the prefix sums.
        uint32_t c01 = 0, c23 = 0; // this thread's heads per class: classes 0, 1 in c01 (low, high half), 2, 3 in c23
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j)
            if (head_mask & (1u << j)) {
                const uint32_t cls = (mine[j].val >> kRunLenShift) & kRunLenMask;
                const uint32_t one = 1u << (16 * (cls & 1u));
                c01 += cls < 2 ? one : 0u;
                c23 += cls < 2 ? 0u : one;
            }
        uint32_t i01 = c01, i23 = c23;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u01 = (uint32_t)__shfl_up((int)i01, o), u23 = (uint32_t)__shfl_up((int)i23, o);
            if (lane >= (uint32_t)o) {
                i01 += u01;
                i23 += u23;
            }
        }
        if (lane == 63) {
            wave_tot01[wave] = i01;
            wave_tot23[wave] = i23;
        }
        __syncthreads();
        uint32_t off01 = 0, off23 = 0, tot01 = 0, tot23 = 0;
        for (uint32_t w = 0; w < 4; ++w) {
            off01 += w < wave ? wave_tot01[w] : 0u;
            off23 += w < wave ? wave_tot23[w] : 0u;
            tot01 += wave_tot01[w];
            tot23 += wave_tot23[w];
        }
        const uint32_t tot[4] = {tot01 & 0xFFFFu, tot01 >> 16, tot23 & 0xFFFFu, tot23 >> 16};
        const uint32_t total = tot[0] + tot[1] + tot[2] + tot[3];
        if (tid == 0)
            chunk_base = total ? atomicAdd(head_count, (unsigned long long)total) : 0ull;
        __syncthreads();
        // class c starts behind the classes before it; inside it: the waves before this one, then the lanes before this one
        const uint32_t mine_before[4] = {(off01 & 0xFFFFu) + ((i01 - c01) & 0xFFFFu), (off01 >> 16) + ((i01 - c01) >> 16),
                                         (off23 & 0xFFFFu) + ((i23 - c23) & 0xFFFFu), (off23 >> 16) + ((i23 - c23) >> 16)};
        uint64_t pos[4];
        pos[0] = chunk_base + mine_before[0];
        pos[1] = chunk_base + tot[0] + mine_before[1];
        pos[2] = chunk_base + tot[0] + tot[1] + mine_before[2];
        pos[3] = chunk_base + tot[0] + tot[1] + tot[2] + mine_before[3];
#pragma unroll
        for (uint32_t j = 0; j < PER; ++j)
            if (head_mask & (1u << j)) {
                const uint32_t cls = (mine[j].val >> kRunLenShift) & kRunLenMask;
                // (a select chain instead of pos[cls]: a dynamically indexed array would live in scratch memory)
                const uint64_t at = cls == 0 ? pos[0] : cls == 1 ? pos[1] : cls == 2 ? pos[2] : pos[3];
                heads[at] = mine[j]; // (<= the entries of the input list)
                pos[0] += cls == 0 ? 1u : 0u;
                pos[1] += cls == 1 ? 1u : 0u;
                pos[2] += cls == 2 ? 1u : 0u;
                pos[3] += cls == 3 ? 1u : 0u;
            }
        __syncthreads();
    }
}

// ... and after the verification of the heads: every band of the original list gives its table slot back
__global__ __launch_bounds__(256) void band_release_kernel(const verify_params P, const band_rec *bands, uint32_t counter)
{
    unsigned long long n = P.counters[counter];
    if (n > P.band_cap)
        n = P.band_cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const band_rec c = bands[i];
        if (c.val != kBandInvalid)
            band_release(P.band_tab, c.slot);
    }
}

// What a band record stands for (verification side).  Returns false if the band has too few seed hits, lies outside the
// owned range, or is an unused list slot.  Consumes the band's table slot (the table is empty again after the scan).
struct band_geom
{
    uint32_t pat;
    int64_t m, k;
    int64_t e_lo, e_hi; // exclusive end positions it answers for
    int64_t ws;         // cold start
    int64_t dd_lo, dd_hi; // end positions <= dd_lo or >= dd_hi go through the dedupe set (heads of runs: the rest cannot
                          // be reported by anybody else); everything, unless band_runs_kernel has analysed the band
};

__device__ __forceinline__ bool decode_band(const verify_params &P, const band_rec &c, bool consume, band_geom &g)
{
    if (c.val == kBandInvalid)
        return false;
    g.pat = c.val >> 11;
    unsigned long long v = 1;
    if (!P.preselected) {
        v = band_value(P.band_tab[c.slot].y, P.overlap != 0);
        if (consume)
            band_release(P.band_tab, c.slot);
        if (c.val & kRunFollower)
            return false; // the head of its run verifies this band's diagonals too
        if (P.overlap ? (uint32_t)v < (P.surplus ? (uint32_t)P.surplus[g.pat] : 1u) : v == 0)
            return false;
    }
    g.m = P.m[g.pat];
    g.k = P.k[g.pat];
    int64_t own_b = (int64_t)P.scan_begin, own_e = (int64_t)P.scan_end, hay_b = (int64_t)P.ctx_begin;
    if (P.seg_offsets) { // every segment is a haystack of its own
        const int64_t sb = (int64_t)P.seg_offsets[c.seg], se = (int64_t)P.seg_offsets[c.seg + 1];
        own_b = P.seg_owned ? sb + (int64_t)P.seg_owned[c.seg] : sb;
        own_e = se;
        hay_b = sb;
    }
    // diagonals d .. d + span: needle position 0 <-> text index d.  Bands that do not overlap know which of their
    // diagonals were hit: an isolated seed hit is verified over its own diagonal, a repeat stretch over the whole band
    const int64_t band_base = hay_b - (int64_t)P.max_m + (int64_t)c.band * (int64_t)P.Bw;
    int64_t d = band_base;
    int64_t span = (int64_t)P.Bw - 1 + (P.overlap ? g.k + 1 : 0);
    if (!P.overlap) {
        const int lo = __ffsll((long long)v) - 1;
        int hi = 63 - __clzll((long long)v);
        if (c.val & kRunAnalysed) // the head of a run of adjacent bands: through the last hit diagonal of its last band
            hi = (int)(((c.val >> kRunLenShift) & kRunLenMask) * P.Bw + (c.val & kRunHiMask));
        d += lo;
        span = hi - lo;
    }
    g.e_lo = d + g.m - g.k;
    g.e_hi = d + g.m + g.k + span;
    g.dd_lo = INT64_MAX; // (every end position is <= this: all of them go through the dedupe set)
    g.dd_hi = INT64_MIN;
    if (!P.overlap && (c.val & kRunAnalysed) && !P.seg_offsets) {
        const int64_t next_base = band_base + (int64_t)(((c.val >> kRunLenShift) & kRunLenMask) + 1) * (int64_t)P.Bw;
        // the run before ends on a diagonal < band_base: its end positions reach band_base - 1 + m + k; the run behind
        // starts on a diagonal >= next_base: its end positions begin at next_base + m - k
        g.dd_lo = (c.val & kRunTouchLeft) ? band_base + g.m + g.k : INT64_MIN;
        g.dd_hi = (c.val & kRunTouchRight) ? next_base + g.m - g.k - 1 : INT64_MAX;
    }
    // ownership: last symbol e-1 in [own_b, own_e)
    if (g.e_lo < own_b + 1)
        g.e_lo = own_b + 1;
    if (g.e_hi > own_e)
        g.e_hi = own_e;
    if (g.e_lo > g.e_hi)
        return false;
    // cold start m+k symbols before the first end position (or at the haystack start)
    g.ws = g.e_lo - (g.m + g.k);
    if (g.ws < hay_b)
        g.ws = hay_b;
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
    // [sigma + 1][NWN][blockDim.x] words, then [16][blockDim.x] uint16: the hits of one text block, then per wave kHitStage
    // hit records on their way out
    extern __shared__ uint32_t vlds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t nthr = blockDim.x;
    const uint32_t rows = P.sigma + 1;
    uint16_t *hitbuf = reinterpret_cast<uint16_t *>(vlds + (size_t)rows * NWN * nthr);
    // Hits leave through a per-wave staging area: a repeat-rich text reports tens of millions of them, and one atomic on
    // the hit counter per wave and text block would be a million atomics on one word (~100 per microsecond).  The wave
    // appends to its stage without atomics and moves kHitStage records out at a time: one atomic, coalesced stores.
    spm_hit *stage = reinterpret_cast<spm_hit *>(hitbuf + (size_t)16 * nthr) + (size_t)(tid >> 6) * kHitStage;
    uint32_t n_staged = 0; // (wave-uniform)
    const uint32_t lane = tid & 63;
    auto flush = [&]() {
        if (n_staged == 0)
            return;
        unsigned long long base = 0;
        if (lane == 0)
            base = atomicAdd(P.hit_counter, (unsigned long long)n_staged);
        base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
               (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base);
        for (uint32_t i = lane; i < n_staged; i += 64)
            if (base + i < P.hit_cap)
                P.hits[base + i] = stage[i];
        n_staged = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    unsigned long long n_cand = P.counters[P.band_counter];
    if (n_cand > P.band_cap)
        n_cand = P.band_cap; // overflow is handled by the host
    const uint64_t stride = (uint64_t)gridDim.x * nthr;
    uint32_t n_valid = 0;
    // wave-uniform trip count (the emission below is wave-collective); a lane without a band idles through the round
    const uint64_t rounds = (n_cand + stride - 1) / stride;
    for (uint64_t rd = 0; rd < rounds; ++rd) {
        const uint64_t ci = rd * stride + (uint64_t)blockIdx.x * nthr + tid;
        band_geom g;
        g.pat = 0;
        g.m = 1;
        g.k = 0;
        g.e_lo = 0;
        g.e_hi = -1;
        g.ws = 0;
        // (false: an unused list slot, too few seed hits, or nothing owned)
        // (runs: the list holds heads only and band_release_kernel gives the slots back afterwards)
        const bool active = ci < n_cand && decode_band(P, P.bands[ci], P.runs == 0, g);
        if (active)
            ++n_valid;
        const uint32_t pat = g.pat;
        const int64_t m = g.m, k = g.k, e_lo = g.e_lo, e_hi = active ? g.e_hi : g.ws, ws = g.ws;
        // stage this needle's rows: brute table [group][row][word][lane], top NWN words
        const uint32_t *src = P.peq32 + (((size_t)(pat >> 6) * rows) * P.nw_table + (P.nw_table - NWN)) * 64 + (pat & 63);
        if (active)
            for (uint32_t r = 0; r < rows; ++r)
#pragma unroll
                for (int w = 0; w < NWN; ++w)
                    vlds[((size_t)r * NWN + w) * nthr + tid] = src[((size_t)r * P.nw_table + w) * 64];
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
        uint4 nxt = make_uint4(0, 0, 0, 0);
        if (active)
            nxt = load_text16(P.text, (uint64_t)blk0, P.text_alloc);
        // columns in 32-bit terms relative to the cold start: rel = p - ws in [0, n_cols); end positions from rel_lo on
        const uint32_t n_cols = (uint32_t)(e_hi - ws);
        const uint32_t rel_lo = (uint32_t)(e_lo - 1 - ws);
        const uint32_t sigma = P.sigma;
        uint32_t rel0 = (uint32_t)(blk0 - ws); // (wraps below zero for the symbols of the first block before ws)
        // The text goes by in 16-symbol blocks; the wave steps through them together (trip count: its longest lane -- the
        // head of a run of bands has up to 1.7 times the columns of a single band) and emits the hits of a block right
        // behind it, all lanes at once: one CAS per hit that needs the dedupe set, then into the wave's stage.
        const uint32_t my_blocks = active ? (uint32_t)((e_hi - blk0 + 15) >> 4) : 0u;
        uint32_t max_blocks = my_blocks;
        for (int o = 32; o > 0; o >>= 1)
            max_blocks = max(max_blocks, (uint32_t)__shfl_xor((int)max_blocks, o));
        max_blocks = (uint32_t)__builtin_amdgcn_readfirstlane(max_blocks);
        for (uint32_t bi = 0; bi < max_blocks; ++bi, rel0 += 16) {
            const int64_t blk = blk0 + 16 * (int64_t)bi;
            uint32_t hitmask = 0;
            if (bi < my_blocks) {
                const uint4 cur = nxt;
                if (bi + 1 < my_blocks)
                    nxt = load_text16(P.text, (uint64_t)(blk + 16), P.text_alloc);
                const uint32_t words[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t rel = rel0 + (uint32_t)i;
                    if (rel >= n_cols)
                        continue;
                    uint32_t sym = (words[i >> 2] >> (8 * (i & 3))) & 0xFF;
                    sym = sym < sigma ? sym : sigma;
                    L.step_strided(vlds + ((size_t)sym * NWN) * nthr + tid, nthr);
                    if (L.score <= (int32_t)k && rel >= rel_lo) {
                        hitbuf[(size_t)i * nthr + tid] = (uint16_t)(L.score + 1);
                        hitmask |= 1u << i;
                    }
                }
            }
            if (__ballot(hitmask != 0) == 0)
                continue;
            // pass 1: dedupe across the seeds / bands of one occurrence; what stays in `keep` is new
            uint32_t fresh = 0, keep = 0;
            for (uint32_t mm = hitmask; mm != 0; mm &= mm - 1) {
                const uint32_t i = (uint32_t)__ffs(mm) - 1u;
                const int64_t e = blk + (int64_t)i + 1; // (exclusive end of an occurrence ending at symbol blk + i)
                if ((e > g.dd_lo && e < g.dd_hi) || seen_insert(P, pat, e)) {
                    ++fresh;
                    keep |= 1u << i;
                }
            }
            // pass 2: into the wave's stage (every lane behind the lanes before it)
            uint32_t incl = fresh;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
                if (lane >= (uint32_t)o)
                    incl += up;
            }
            const uint32_t total = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl((int)incl, 63));
            if (total == 0)
                continue;
            if (n_staged + total > kHitStage)
                flush();
            unsigned long long direct = 0; // more hits in one block than the stage holds: straight to the hit list
            const bool staged = total <= kHitStage;
            if (!staged) {
                if (lane == 0)
                    direct = atomicAdd(P.hit_counter, (unsigned long long)total);
                direct = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(direct >> 32)) << 32) |
                         (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)direct);
            }
            uint32_t at = n_staged + incl - fresh;
            for (uint32_t mm = keep; mm != 0; mm &= mm - 1) {
                const uint32_t i = (uint32_t)__ffs(mm) - 1u;
                const int64_t e = blk + (int64_t)i + 1;
                spm_hit h;
                h.pos = (P.report_begin ? (uint64_t)(e - m) : (uint64_t)e) + P.pos_offset;
                h.pattern = pat;
                h.score = (int32_t)hitbuf[(size_t)i * nthr + tid] - 1;
                if (staged)
                    stage[at] = h;
                else if (direct + at < P.hit_cap)
                    P.hits[direct + at] = h;
                ++at;
            }
            if (staged)
                n_staged += total;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    flush();
    wave_count_add(P.hit_counter + 7, n_valid); // bands verified
}


// ---- wave-per-candidate verification for long needles ----------------------------------------------------------
// One lane per 32-row block of the needle instead of one lane per candidate: G lanes form a systolic array that runs
// the block-based Myers recurrence (Myers 1999 / Hyyro 2003: every block takes the horizontal delta hin of the block
// above and hands its own hout down).  At step t lane b processes text column t - b; the text symbol and hout travel
// one lane per step through DPP wave_shr:1.  A column of a 1024-row needle costs one step of ~40 instructions
// instead of 32 blocks x 13 on a single lane, so the latency of one verification drops from ~1.5 ms to ~0.1 ms --
// what matters when a scan leaves a few hundred long bands to verify.  Reads the bottom-aligned table of the cut-off
// kernel ([group][row][nw_table][64]); each lane keeps the <= 5 match masks of its block in registers.
// NB: 32-row blocks per lane.  A scan leaves a few thousand long bands; with one block per lane a |P| = 1024 band takes 32
// lanes, a wave holds two bands, and 2 800 bands are 1 400 busy waves on 1 024 SIMDs: the SIMDs that got two of them run
// every step twice, and they set the kernel's duration.  Two blocks per lane (the second takes the first one's hout in the
// same step) put four bands into a wave -- 700 busy waves, at most one per SIMD -- at ~1.7x the instructions of a step.
template <int G, int NB>
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
    unsigned long long n_cand = P.counters[P.band_counter];
    if (n_cand > P.band_cap)
        n_cand = P.band_cap;
    const uint64_t stride = (uint64_t)gridDim.x * waves * GPW;
    const uint32_t rows = P.sigma + 1;
    uint32_t n_valid = 0;
    for (uint64_t base = ((uint64_t)blockIdx.x * waves + wave) * GPW; base < n_cand; base += stride) {
        const uint64_t ci = base + gw;
        bool active = ci < n_cand;
        uint32_t pat = 0;
        int64_t m = 1, k = 0, e_lo = 0, e_hi = -1, ws = 0;
        band_rec c;
        c.val = kBandInvalid;
        c.slot = 0;
        if (active)
            c = P.bands[ci];
        {
            band_geom g;
            g.pat = 0;
            g.m = 1;
            g.k = 0;
            g.e_lo = 0;
            g.e_hi = -1;
            g.ws = 0;
            active = active && decode_band(P, c, false, g); // every lane of the group reads the band's count ...
            pat = g.pat;
            m = g.m;
            k = g.k;
            e_lo = g.e_lo;
            e_hi = g.e_hi;
            ws = g.ws;
        }
        __builtin_amdgcn_wave_barrier();
        if (gl == 0 && c.val != kBandInvalid && !P.preselected) // ... then one of them gives the table slot back
            band_release(P.band_tab, c.slot);
        if (active && gl == 0)
            ++n_valid;
        const uint32_t nbk = (uint32_t)((m + 31) >> 5);       // blocks of this needle
        const uint32_t nb = (nbk + NB - 1) / NB;              // lanes that hold them
        const uint32_t n_cols = active ? (uint32_t)(e_hi - ws) : 0u;
        const bool mine = active && gl < nb;
        const bool lane_last = gl + 1 == nb;                  // holds the needle's last block, at index j_last
        const uint32_t j_last = (nbk - 1) % NB;
        uint32_t out_bit[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j)
            out_bit[j] = (gl * NB + (uint32_t)j + 1 == nbk) ? (uint32_t)((m - 1) & 31) : 31u; // where block j's hout is read
        // ---- stage the text window [ws, e_hi) into LDS: 16-byte blocks, the group's lanes side by side ----
        const uint32_t skew = (uint32_t)(ws & 15);
        uint32_t beyond4 = 0; // a byte >= 4 somewhere in the window (never in a validated dna4 text)
        if (active) {
            const uint32_t n_bytes = (skew + n_cols + 15) & ~15u;
            for (uint32_t off = gl * 16; off < n_bytes; off += G * 16) {
                const uint4 v = load_text16(P.text, ((uint64_t)ws & ~15ull) + off, P.text_alloc);
                *reinterpret_cast<uint4 *>(tw + off) = v;
                beyond4 |= (v.x | v.y | v.z | v.w) & 0xFCFCFCFCu;
            }
        }
        const bool plain4 = P.sigma == 4 && __ballot(beyond4 != 0) == 0; // (wave-uniform) every symbol selects a match mask
        uint32_t e0[NB], e1[NB], e2[NB], e3[NB], e4[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            e0[j] = e1[j] = e2[j] = e3[j] = e4[j] = 0;
            const uint32_t gb = gl * NB + (uint32_t)j; // (a block beyond the needle matches nothing)
            if (mine && gb < nbk) {
                const uint32_t *src = peq_bot + (((size_t)(pat >> 6) * rows) * P.nw_table + gb) * 64 + (pat & 63);
                const size_t rs = (size_t)P.nw_table * 64;
                e0[j] = src[0];
                e1[j] = src[rs];
                e2[j] = P.sigma > 2 ? src[2 * rs] : 0u;
                e3[j] = P.sigma > 3 ? src[3 * rs] : 0u;
                e4[j] = P.sigma > 4 ? src[4 * rs] : 0u;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t Pv[NB], Mv[NB], ho = 0;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            Pv[j] = 0xFFFFFFFFu;
            Mv[j] = 0;
        }
        int32_t score = (int32_t)m;
        bool any_hit = false;
        const uint32_t first_slot_col = (uint32_t)(e_lo - ws) - 1; // column whose end position is e_lo
        const uint32_t t_end = n_cols + nb - 1;                    // steps until the last block has seen the last column
        // wave-uniform trip count: the longest group decides
        uint32_t t_wave = active ? t_end : 0u;
        // ... and the steps in which EVERY block of every band of the wave has a column of its window: [t_lo, t_hi)
        uint32_t t_lo = active ? nb - 1 : 0u, t_hi = active ? n_cols : 0xFFFFFFFFu;
        // ... and the first step at which a band of the wave can report an end position (its last block reaches the column
        // of e_lo): the steps before it -- three in four for |P| = 1024, k = 64 -- do not look for hits at all
        uint32_t t_hit = active ? first_slot_col + nb - 1 : 0xFFFFFFFFu;
        for (int o = 32; o >= (int)G; o >>= 1) {
            t_wave = max(t_wave, (uint32_t)__shfl_xor((int)t_wave, o));
            t_lo = max(t_lo, (uint32_t)__shfl_xor((int)t_lo, o));
            t_hi = min(t_hi, (uint32_t)__shfl_xor((int)t_hi, o));
            t_hit = min(t_hit, (uint32_t)__shfl_xor((int)t_hit, o));
        }
        t_wave = (uint32_t)__builtin_amdgcn_readfirstlane(t_wave);
        t_lo = (uint32_t)__builtin_amdgcn_readfirstlane(t_lo);
        t_hi = (uint32_t)__builtin_amdgcn_readfirstlane(t_hi);
        t_hit = (uint32_t)__builtin_amdgcn_readfirstlane(t_hit);
        t_hi = t_hi > t_wave ? t_wave : t_hi;
        t_lo = t_lo > t_hi ? t_hi : t_lo;
        t_hit = t_hit < t_lo ? t_lo : (t_hit > t_hi ? t_hi : t_hit);
        uint32_t sym_next = mine ? tw[skew] : 0u; // column 0 (clamped reads below keep every index inside the window)
        // One step of one lane's blocks.  CHECKED: the lane may have no column at this step (the pipeline fills and drains,
        // or a shorter band shares the wave).  The steps in between -- nearly all -- run without exec-mask changes and with
        // the match mask picked by bit selects instead of compare / cndmask chains: a band is a serial chain of ~1500
        // steps, so the scan pays for every instruction and every VALU -> SALU hand-over of a step.
        // (unchecked steps: a lane that holds a block reads its own column, any other lane anything inside its group's LDS;
        // only the lane of the last block can meet score <= k_hit)
        const uint8_t *my_text = mine ? tw + skew - gl : tw;
        const int32_t k_hit = (lane_last && mine) ? (int32_t)k : -1;
        const uint32_t n_rel = n_cols - first_slot_col; // end-position slots of this band
        auto step = [&](uint32_t t, auto checked, auto dna4, auto hits) {
            const uint32_t ho_up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ho, 0x138, 0xF, 0xF, false);
            const uint32_t col = t - gl; // wraps for t < gl: then col >= n_cols
            const uint32_t sym = sym_next;
            if constexpr (decltype(checked)::value) { // next column's symbol, one step ahead of its use
                uint32_t nc = col + 1;
                nc = nc < n_cols ? nc : 0u;
                sym_next = tw[skew + nc];
            } else {
                sym_next = my_text[t + 1];
            }
            if (!decltype(checked)::value || (mine && col < n_cols)) {
                uint32_t hin = gl == 0 ? 0u : ho_up;
                uint32_t s0 = 0, s1 = 0;
                if constexpr (decltype(dna4)::value) {
                    s0 = (uint32_t)((int32_t)(sym << 31) >> 31);
                    s1 = (uint32_t)((int32_t)(sym << 30) >> 31);
                }
                int32_t d_last = 0; // op - on of the needle's last block (meaningful on its lane only)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    uint32_t Eq;
                    if constexpr (decltype(dna4)::value) {
                        const uint32_t lo2 = (e1[j] & s0) | (e0[j] & ~s0), hi2 = (e3[j] & s0) | (e2[j] & ~s0);
                        Eq = (hi2 & s1) | (lo2 & ~s1); // (plain4: no symbol beyond 3 in the window)
                    } else {
                        const uint32_t lo2 = (sym & 1u) ? e1[j] : e0[j], hi2 = (sym & 1u) ? e3[j] : e2[j];
                        Eq = (sym & 2u) ? hi2 : lo2;
                        Eq = sym < 4 ? Eq : (sym == 4 ? e4[j] : 0u);
                        Eq = sym < P.sigma ? Eq : 0u;
                    }
                    const uint32_t hp = hin & 1u, hn = hin >> 1;
                    const uint32_t Xv = Eq | Mv[j];
                    Eq |= hn;
                    const uint32_t Xh = (((Eq & Pv[j]) + Pv[j]) ^ Pv[j]) | Eq;
                    uint32_t Ph = Mv[j] | ~(Xh | Pv[j]);
                    uint32_t Mh = Pv[j] & Xh;
                    const uint32_t op = (Ph >> out_bit[j]) & 1u, on = (Mh >> out_bit[j]) & 1u;
                    hin = op | (on << 1); // this block's hout: the next block's hin, in this very step
                    Ph = (Ph << 1) | hp;
                    Mh = (Mh << 1) | hn;
                    Pv[j] = Mh | ~(Xv | Ph);
                    Mv[j] = Ph & Xv;
                    if (NB == 1 || (uint32_t)j == j_last)
                        d_last = (int32_t)op - (int32_t)on;
                }
                ho = hin;
                score += d_last;
                if constexpr (decltype(hits)::value) {
                    const uint32_t rel = col - first_slot_col;
                    if (score <= k_hit && rel < n_rel) {
                        hb[rel] = (uint16_t)(score + 1);
                        any_hit = true;
                    }
                }
            }
        };
        uint32_t t = 0;
        for (; t < t_lo; ++t)
            step(t, std::true_type{}, std::false_type{}, std::true_type{});
        if (plain4) {
            for (; t < t_hit; ++t)
                step(t, std::false_type{}, std::true_type{}, std::false_type{});
            for (; t < t_hi; ++t)
                step(t, std::false_type{}, std::true_type{}, std::true_type{});
        } else {
            for (; t < t_hi; ++t)
                step(t, std::false_type{}, std::false_type{}, std::true_type{});
        }
        for (; t < t_wave; ++t)
            step(t, std::true_type{}, std::false_type{}, std::true_type{});
        // ---- emission: the lanes of a group share its slots; wave-converged appends ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (__ballot(any_hit) != 0) {
            // pass 1: every lane of a group takes the slots r = gl, gl + G, ..: dedupe, keep what is new
            const uint32_t nr = active ? (uint32_t)(e_hi - e_lo) + 1 : 0u;
            uint32_t fresh = 0;
            for (uint32_t r = gl; r < nr; r += G) {
                const uint32_t sc1 = hb[r];
                if (sc1) {
                    if (seen_insert(P, pat, e_lo + (int64_t)r))
                        ++fresh;
                    else
                        hb[r] = 0;
                }
            }
            // pass 2: one reservation for the wave, then every lane writes its run
            unsigned long long idx = wave_reserve_hits(P.hit_counter, fresh);
            for (uint32_t r = gl; r < nr; r += G) {
                const uint32_t sc1 = hb[r];
                hb[r] = 0;
                if (sc1) {
                    if (idx < P.hit_cap) {
                        const int64_t e = e_lo + (int64_t)r;
                        spm_hit h;
                        h.pos = (P.report_begin ? (uint64_t)(e - m) : (uint64_t)e) + P.pos_offset;
                        h.pattern = pat;
                        h.score = (int32_t)sc1 - 1;
                        P.hits[idx] = h;
                    }
                    ++idx;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    wave_count_add(P.hit_counter + 7, n_valid); // bands verified
}

} // namespace spm_hip
