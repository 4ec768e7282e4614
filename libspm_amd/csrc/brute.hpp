// brute.hpp -- brute-force scan kernels: ONE LANE PER PATTERN, every text symbol through the recurrence.
//
// This is the kernel shape BASELINE.json's north_star names: the 64 lanes of a wavefront hold 64 different
// needles; the text symbol is wave-uniform; the per-needle match bit-masks (Peq) sit in LDS laid out
// [symbol][word][lane] so that a wave reads one conflict-free 256-byte row per word; the text is streamed with
// coalesced dword loads and broadcast through v_readlane; hits are compacted with ballot/popc.
//
// Recurrence (replaces [upstream] _findMyersSmallPatterns/_findMyersLargePatterns, reached from
// /root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:60-65 and myers_matcher_restorable.hpp:50-54):
//   X = Peq[c] | VN;  D0 = ((VP + (X & VP)) ^ VP) | X;  HN = VP & D0;  HP = VN | ~(VP | D0)
//   X = HP << 1;  VN = X & D0;  VP = (HN << 1) | ~(X | D0);  score += HP[top] - HN[top]
// on NW 32-bit words with the add carry and the two shift carries chained across words.
//
// Layout trick: every needle is TOP-ALIGNED in its NW*32-bit vector (row j of the needle sits at bit off+j,
// off = 32*NW - |P|).  Bits below `off` are wildcard rows (Peq = 1 for every symbol, VP = 0 initially); they hold
// D = 0 in every column, exactly like the DP's row 0, so the needle's rows see the same boundary as in the
// unshifted formulation -- and the score delta is always bit 31 of the top word for every lane, whatever |P|.
//
// Tiling (SURVEY.md 8(e) / spm::window_size): the scan range is cut into tiles; a tile starts its recurrence cold
// W-1 symbols early (W = max |P|+k) and reports only hits whose last symbol it owns.  A cold start at s' computes
// the DP of the suffix text[s'..]; every cell with D <= k has an optimal alignment that spans <= |P|+k symbols, so
// value and <= k membership agree with the sequential scan for every owned position.
#pragma once

#include "common.hpp"

namespace spm_hip
{

struct brute_params
{
    const uint8_t *text;   // text[0] = first symbol; 4-byte aligned
    uint64_t text_alloc;   // readable bytes from text
    uint64_t scan_begin;   // owned range: hits whose LAST symbol index is in [scan_begin, scan_end)
    uint64_t scan_end;
    uint64_t ctx_begin;    // lowest symbol index that may be consumed as warm-up
    uint64_t pos_offset;   // added to reported positions
    uint32_t tile;         // owned symbols per tile (multiple of 4)
    uint32_t n_tiles;
    uint32_t n_groups;     // ceil(n_patterns / 64)
    uint32_t warm;         // warm-up symbols = max window - 1
    uint32_t sigma;        // rows 0..sigma-1 real symbols, row sigma = "no match" (invalid symbol)
    uint32_t has_state;    // 1: tile 0 continues from state_in instead of a cold start
    const uint64_t *tile_tab; // segmented scans: per tile {scan_lo, own_lo, own_hi}; nullptr = regular tiling
    const uint32_t *peq;   // [group][sigma+1][NW][64]
    const uint32_t *hp0;   // prefix mode: [group][NW][64] carry-in mask (bit `off`), else nullptr
    const int32_t *m;      // [group*64] needle lengths (0 = padding lane)
    const int32_t *k;      // [group*64]
    const uint32_t *state_in; // [group][2*NW+1][64] internal layout (VP words, VN words, score) or nullptr
    uint32_t *state_out;      // same layout, or nullptr
    spm_hit *hits;
    unsigned long long *counters; // [0] = hit count
    uint64_t hit_cap;
    // span-local fallback of the filter engine: this launch re-scans the spans whose candidates overflowed, and must not
    // report a hit the filter's verification already reported.  seen = the scan's dedupe set (nullptr otherwise).
    unsigned long long *seen;
    uint32_t seen_mask;
    unsigned long long *overflow;
};

__device__ __forceinline__ uint32_t alignbit(uint32_t hi, uint32_t lo, uint32_t sh)
{
    return __builtin_amdgcn_alignbit(hi, lo, sh);
}

// a value that is the same in every lane, moved to SGPRs (the compiler cannot prove uniformity of a loaded value)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) |
           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
}

// append the hits of one wave: ballot + popc prefix, one atomic per wave
__device__ __forceinline__ void wave_append_hits(bool is_hit, uint64_t pos, uint32_t pattern, int32_t score,
                                                 spm_hit *hits, unsigned long long *counter, uint64_t cap)
{
    const uint64_t mask = __ballot(is_hit);
    if (mask == 0)
        return;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t n = __popcll(mask);
    unsigned long long base = 0;
    if (lane == (uint32_t)__ffsll((unsigned long long)mask) - 1)
        base = atomicAdd(counter, (unsigned long long)n);
    base = __shfl(base, __ffsll((unsigned long long)mask) - 1);
    if (is_hit) {
        const uint64_t idx = base + __popcll(mask & ((1ull << lane) - 1));
        if (idx < cap) {
            spm_hit h;
            h.pos = pos;
            h.pattern = pattern;
            h.score = score;
            hits[idx] = h;
        }
    }
}

__host__ __device__ inline uint64_t seen_hash(uint64_t z) // splitmix64 finaliser (same function as synth.hpp's mix64)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Hit of the brute-force kernels.  `e` = exclusive end of the occurrence in text coordinates (the dedupe key of the
// filter engine's verification), `pos` = what is reported.  Call with the wave converged.
__device__ __forceinline__ void brute_report(const brute_params &P, bool hit, uint64_t e, uint64_t pos, uint32_t pattern,
                                             int32_t score)
{
    if (P.seen) {
        if (hit) {
            const unsigned long long key = ((unsigned long long)pattern << 40) | (unsigned long long)e;
            uint32_t slot = (uint32_t)seen_hash(key) & P.seen_mask;
            bool placed = false, fresh = false;
            for (uint32_t tries = 0; tries < 64 && !placed; ++tries) { // (as seen_insert, filter.hpp)
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
            hit = fresh;
        }
        if (__ballot(hit) == 0)
            return;
    }
    wave_append_hits(hit, pos, pattern, score, P.hits, P.counters, P.hit_cap);
}

// Load the dword holding text[idx..idx+4) for idx % 4 == 0, never touching bytes >= alloc.
__device__ __forceinline__ uint32_t load_text_dword(const uint8_t *text, uint64_t idx, uint64_t alloc)
{
    if (idx + 4 <= alloc)
        return *reinterpret_cast<const uint32_t *>(text + idx);
    uint32_t v = 0;
    for (int b = 0; b < 4; ++b)
        if (idx + b < alloc)
            v |= (uint32_t)text[idx + b] << (8 * b);
    return v;
}

// ---------------------------------------------------------------------------------------------------
// Myers, NW 32-bit words per needle, one lane per needle.  PREFIX = MyersUkkonenGlobal carry-in.
// Everything except the needle state is wave-uniform and is forced into SGPRs (readfirstlane), so the per-symbol
// VALU work is the recurrence itself: per word 1 LDS read, ~9 VALU (v_bitop3 folds the 3-input boolean terms).
// ---------------------------------------------------------------------------------------------------
template <int NW, bool PREFIX>
struct myers_lane
{
    uint32_t VP[NW], VN[NW];
    uint32_t hp0[PREFIX ? NW : 1];
    int32_t score;

    __device__ __forceinline__ void step(const uint32_t *row /* &peq[c][0][lane] */) { step_strided(row, 64); }

    // word w of the row sits at row[w * stride] (64 in the brute kernels, blockDim.x in the verify kernel)
    __device__ __forceinline__ void step_strided(const uint32_t *row, uint32_t stride)
    {
        uint32_t carry = 0, hp_prev = 0, hn_prev = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            // 10 VALU per word: v_bitop3_b32 evaluates any 3-input boolean function (truth table with
            // a = 0xF0, b = 0xCC, c = 0xAA), which folds every AND/OR/NOT pair of the recurrence.
            const uint32_t eq = row[(size_t)w * stride];
            const uint32_t t = __builtin_amdgcn_bitop3_b32(eq, VN[w], VP[w], 0xA8); // (eq | VN) & VP
            uint32_t cout;
            const uint32_t sum = __builtin_addc(VP[w], t, carry, &cout);
            carry = cout;
            const uint32_t X = eq | VN[w];
            const uint32_t D0 = __builtin_amdgcn_bitop3_b32(sum, VP[w], X, 0xBE);    // (sum ^ VP) | X
            const uint32_t HN = VP[w] & D0;
            const uint32_t HP = __builtin_amdgcn_bitop3_b32(VN[w], VP[w], D0, 0xF1); // VN | ~(VP | D0)
            uint32_t Xs = alignbit(HP, hp_prev, 31);
            if (PREFIX)
                Xs |= hp0[w];
            const uint32_t Ts = alignbit(HN, hn_prev, 31);
            hp_prev = HP;
            hn_prev = HN;
            VN[w] = Xs & D0;
            VP[w] = __builtin_amdgcn_bitop3_b32(Ts, Xs, D0, 0xF1);                   // Ts | ~(Xs | D0)
        }
        score += (int32_t)(hp_prev >> 31) - (int32_t)(hn_prev >> 31);
    }
};

template <int NW, bool PREFIX>
__global__ __launch_bounds__(256) void myers_brute_kernel(const brute_params P)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t rows = P.sigma + 1;
    uint32_t *my_peq = lds + (size_t)wave_in_wg * rows * NW * 64; // this wave's table: [row][word][lane]

    const uint64_t n_items = (uint64_t)P.n_tiles * P.n_groups;
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;

    uint32_t loaded_group = 0xFFFFFFFFu;
    int32_t my_m = 0, my_k = -1;
    myers_lane<NW, PREFIX> L;

    for (uint64_t item = wave_id; item < n_items; item += n_waves) {
        const uint32_t group = (uint32_t)(item % P.n_groups);
        const uint32_t tile = (uint32_t)(item / P.n_groups);
        if (group != loaded_group) {
            const uint32_t *src = P.peq + (size_t)group * rows * NW * 64;
            for (uint32_t i = lane; i < rows * NW * 64; i += 64)
                my_peq[i] = src[i];
            my_m = P.m[group * 64 + lane];
            my_k = my_m > 0 ? P.k[group * 64 + lane] : -1;
            if (PREFIX) {
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    L.hp0[w] = P.hp0[((size_t)group * NW + w) * 64 + lane];
            }
            loaded_group = group;
            __builtin_amdgcn_wave_barrier();
        }
        // ---- tile geometry (wave-uniform, SGPRs) ----
        uint64_t own_lo = P.scan_begin + (uint64_t)tile * P.tile;
        uint64_t own_hi = own_lo + P.tile;
        if (own_hi > P.scan_end)
            own_hi = P.scan_end;
        uint64_t scan_lo = own_lo >= P.ctx_begin + P.warm ? own_lo - P.warm : P.ctx_begin;
        if (P.tile_tab) { // segmented haystacks: geometry comes from the host's tile table
            scan_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile]);
            own_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 1]);
            own_hi = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 2]);
        }
        const bool resume = P.has_state && tile == 0;
        if (resume)
            scan_lo = own_lo;

        // ---- initial state ----
        if (resume) {
            const uint32_t *st = P.state_in + (size_t)group * (2 * NW + 1) * 64 + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                L.VP[w] = st[w * 64];
                L.VN[w] = st[(NW + w) * 64];
            }
            L.score = (int32_t)st[2 * NW * 64];
        } else {
            const int32_t off = NW * 32 - my_m;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int32_t lo = off - w * 32; // bits >= lo are needle rows
                L.VP[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
                L.VN[w] = 0;
            }
            L.score = my_m;
        }
        const uint32_t *lane_peq = my_peq + lane;
        const uint32_t sigma = P.sigma;

        const uint64_t a0 = scan_lo & ~3ull;
        for (uint64_t cbase = a0; cbase < own_hi; cbase += 256) {
            const uint64_t my_idx = cbase + (uint64_t)lane * 4;
            uint32_t v = 0;
            if (my_idx < own_hi)
                v = load_text_dword(P.text, my_idx, P.text_alloc);
            const uint64_t rem = own_hi - cbase;
            const bool full = (cbase >= scan_lo) && (rem >= 256);
            if (full && cbase + 256 <= own_lo) {
                // warm-up chunk: recurrence only
                for (uint32_t j = 0; j < 64; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        c = c < sigma ? c : sigma;
                        L.step(lane_peq + (size_t)c * NW * 64);
                    }
                }
            } else if (full && cbase >= own_lo) {
                // owned interior chunk: recurrence + hit test, no range checks (batching the test over 4 symbols, as the
                // Shift-Or kernel does, measured 4 % slower here: the four kept scores cost more than three ballots)
                for (uint32_t j = 0; j < 64; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        c = c < sigma ? c : sigma;
                        L.step(lane_peq + (size_t)c * NW * 64);
                        const bool hit = L.score <= my_k;
                        if (__ballot(hit) != 0)
                            brute_report(P, hit, cbase + (uint64_t)j * 4 + s + 1,
                                         cbase + (uint64_t)j * 4 + s + 1 + P.pos_offset, group * 64 + lane, L.score);
                    }
                }
            } else {
                // edge chunk: every symbol checked against [scan_lo, own_hi) and own_lo
                const uint32_t n_dw = rem >= 256 ? 64u : (uint32_t)((rem + 3) / 4);
                for (uint32_t j = 0; j < n_dw; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
                    for (int s = 0; s < 4; ++s) {
                        const uint64_t p = cbase + (uint64_t)j * 4 + s;
                        if (p < scan_lo || p >= own_hi)
                            continue;
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        c = c < sigma ? c : sigma;
                        L.step(lane_peq + (size_t)c * NW * 64);
                        const bool hit = L.score <= my_k;
                        if (p >= own_lo && __ballot(hit) != 0)
                            brute_report(P, hit, p + 1, p + 1 + P.pos_offset, group * 64 + lane, L.score);
                    }
                }
            }
        }
        // The state after the last symbol is exact only when the recurrence ran over every symbol since the
        // matcher was constructed or 2*|P| symbols from a cold start (host arranges a single-tile pass for it).
        if (P.state_out && P.n_tiles == 1) {
            uint32_t *st = P.state_out + (size_t)group * (2 * NW + 1) * 64 + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                st[w * 64] = L.VP[w];
                st[(NW + w) * 64] = L.VN[w];
            }
            st[2 * NW * 64] = (uint32_t)L.score;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Myers with Ukkonen cut-off, one lane per needle (stateless scans: the throughput case).
//
// The reference's large-pattern path only computes the blocks inside the band of cells that can still be <= k
// ([upstream] _findMyersLargePatterns, reached from myers_matcher_restorable.hpp:53-54); this is the same idea at
// 32-row granularity, per lane:
//   a    = number of active words (rows 0 .. 32a-1), S = D at the bottom row of word a-1 in the current column
//   grow   before a column, if S <= k and a < nw: word a enters with vertical deltas +1 (an over-estimate of cells
//          that are all > k, which keeps every cell whose true value is <= k exact -- Myers 1999, sec. 4)
//   shrink after a column, while a > 1 and S >= k + rows(a-1): the whole last word is > k; S moves up by the word's
//          vertical deltas (popc(VP) - popc(VN))
//   hit    a == nw and S <= k
// Needles are BOTTOM-aligned here (row j at bit j) so that the band starts in word 0.  On random text with k <= 3
// every lane sits at a = 1 almost always; the wave takes a one-word fast path then (amax == 1) and the general
// masked multi-word path otherwise.  tests: brute-cutoff == brute-full == oracle.
// ---------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void myers_cutoff_kernel(const brute_params P)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t rows = P.sigma + 1;
    uint32_t *my_peq = lds + (size_t)wave_in_wg * rows * NW * 64; // [row][word][lane], bottom-aligned

    const uint64_t n_items = (uint64_t)P.n_tiles * P.n_groups;
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;

    uint32_t loaded_group = 0xFFFFFFFFu;
    int32_t my_m = 0, my_k = -1, my_nw = 1, my_lb = 0;

    for (uint64_t item = wave_id; item < n_items; item += n_waves) {
        const uint32_t group = (uint32_t)(item % P.n_groups);
        const uint32_t tile = (uint32_t)(item / P.n_groups);
        if (group != loaded_group) {
            const uint32_t *src = P.peq + (size_t)group * rows * NW * 64;
            for (uint32_t i = lane; i < rows * NW * 64; i += 64)
                my_peq[i] = src[i];
            my_m = P.m[group * 64 + lane];
            my_k = my_m > 0 ? P.k[group * 64 + lane] : -1;
            my_nw = my_m > 0 ? (my_m + 31) >> 5 : 1;
            my_lb = my_m > 0 ? (my_m - 1) & 31 : 31;
            loaded_group = group;
            __builtin_amdgcn_wave_barrier();
        }
        uint64_t own_lo = P.scan_begin + (uint64_t)tile * P.tile;
        uint64_t own_hi = own_lo + P.tile;
        if (own_hi > P.scan_end)
            own_hi = P.scan_end;
        uint64_t scan_lo = own_lo >= P.ctx_begin + P.warm ? own_lo - P.warm : P.ctx_begin;
        if (P.tile_tab) {
            scan_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile]);
            own_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 1]);
            own_hi = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 2]);
        }

        // ---- cold start: D[i][0] = i, band = rows with D <= k ----
        uint32_t VP[NW], VN[NW];
        int32_t a = my_k >= 0 ? (my_k >> 5) + 1 : 1;
        if (a > my_nw)
            a = my_nw;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            VP[w] = 0xFFFFFFFFu;
            VN[w] = 0;
        }
        int32_t S = my_m > 0 ? (a == my_nw ? my_m : 32 * a) : 0x3FFFFFFF;
        // derived per-lane values, refreshed whenever `a` changes
        int32_t bp = a == my_nw ? my_lb : 31;                                  // bit of the band's bottom row
        int32_t shrink_at = a > 1 ? my_k + (bp + 1) : 0x7FFFFFFF;              // S >= this: last word is all > k
        int32_t amax = 1;                                                      // wave maximum of a (SGPR)
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (__ballot(a > w) != 0)
                amax = w + 1;
        const uint32_t *lane_peq = my_peq + lane;
        const uint32_t sigma = P.sigma;

        auto adjust = [&](bool report, uint64_t pos) {
            // slow path, entered when some lane has S <= k or S >= shrink_at
            bool hit = (a == my_nw) && (S <= my_k);
            if (report && __ballot(hit) != 0)
                brute_report(P, hit, pos - P.pos_offset, pos, group * 64 + lane, S);
            // shrink
            while (a > 1 && S >= my_k + (bp + 1)) {
                uint32_t vp = 0, vn = 0;
#pragma unroll
                for (int w = 1; w < NW; ++w)
                    if (w == a - 1) {
                        vp = VP[w];
                        vn = VN[w];
                    }
                const uint32_t rm = bp == 31 ? 0xFFFFFFFFu : ((2u << bp) - 1);
                S -= (int32_t)__popc(vp & rm) - (int32_t)__popc(vn & rm);
                --a;
                bp = 31;
            }
            // grow (for the next column)
            if (a < my_nw && S <= my_k) {
#pragma unroll
                for (int w = 1; w < NW; ++w)
                    if (w == a) {
                        VP[w] = 0xFFFFFFFFu;
                        VN[w] = 0;
                    }
                ++a;
                bp = a == my_nw ? my_lb : 31;
                S += bp + 1;
            }
            shrink_at = a > 1 ? my_k + (bp + 1) : 0x7FFFFFFF;
            amax = 1;
#pragma unroll
            for (int w = 1; w < NW; ++w)
                if (__ballot(a > w) != 0)
                    amax = w + 1;
        };

        auto step = [&](uint32_t c, bool report, uint64_t pos) {
            const uint32_t *row = lane_peq + (size_t)c * NW * 64;
            // word 0 is inside every lane's band (a >= 1)
            uint32_t hpl, hnl;
            uint32_t carry;
            {
                const uint32_t eq = row[0];
                const uint32_t t = __builtin_amdgcn_bitop3_b32(eq, VN[0], VP[0], 0xA8);
                const uint32_t sum = __builtin_addc(VP[0], t, 0u, &carry);
                const uint32_t X = eq | VN[0];
                const uint32_t D0 = __builtin_amdgcn_bitop3_b32(sum, VP[0], X, 0xBE);
                hnl = VP[0] & D0;
                hpl = __builtin_amdgcn_bitop3_b32(VN[0], VP[0], D0, 0xF1);
                const uint32_t Xs = hpl << 1;
                const uint32_t Ts = hnl << 1;
                VN[0] = Xs & D0;
                VP[0] = __builtin_amdgcn_bitop3_b32(Ts, Xs, D0, 0xF1);
            }
            if (amax > 1) {
                // some lane has a wider band: the remaining words, masked per lane
                uint32_t hp_prev = hpl, hn_prev = hnl;
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    if (w < amax) {
                        const uint32_t eq = row[w * 64];
                        const uint32_t t = __builtin_amdgcn_bitop3_b32(eq, VN[w], VP[w], 0xA8);
                        uint32_t cout;
                        const uint32_t sum = __builtin_addc(VP[w], t, carry, &cout);
                        const uint32_t X = eq | VN[w];
                        const uint32_t D0 = __builtin_amdgcn_bitop3_b32(sum, VP[w], X, 0xBE);
                        const uint32_t HN = VP[w] & D0;
                        const uint32_t HP = __builtin_amdgcn_bitop3_b32(VN[w], VP[w], D0, 0xF1);
                        const uint32_t Xs = alignbit(HP, hp_prev, 31);
                        const uint32_t Ts = alignbit(HN, hn_prev, 31);
                        if (w < a) { // lanes whose band ends earlier keep their (inactive) words
                            VN[w] = Xs & D0;
                            VP[w] = __builtin_amdgcn_bitop3_b32(Ts, Xs, D0, 0xF1);
                            carry = cout;
                            hp_prev = HP;
                            hn_prev = HN;
                        }
                        if (w == a - 1) {
                            hpl = HP;
                            hnl = HN;
                        }
                    }
                }
            }
            S += (int32_t)((hpl >> bp) & 1) - (int32_t)((hnl >> bp) & 1);
            if (__ballot(S <= my_k || S >= shrink_at) != 0)
                adjust(report, pos);
        };

        const uint64_t a0 = scan_lo & ~3ull;
        for (uint64_t cbase = a0; cbase < own_hi; cbase += 256) {
            const uint64_t my_idx = cbase + (uint64_t)lane * 4;
            uint32_t v = 0;
            if (my_idx < own_hi)
                v = load_text_dword(P.text, my_idx, P.text_alloc);
            const uint64_t rem = own_hi - cbase;
            const bool full = (cbase >= scan_lo) && (rem >= 256);
            if (full && (cbase + 256 <= own_lo || cbase >= own_lo)) {
                const bool report = cbase >= own_lo;
                for (uint32_t j = 0; j < 64; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        c = c < sigma ? c : sigma;
                        step(c, report, cbase + (uint64_t)j * 4 + s + 1 + P.pos_offset);
                    }
                }
            } else {
                const uint32_t n_dw = rem >= 256 ? 64u : (uint32_t)((rem + 3) / 4);
                for (uint32_t j = 0; j < n_dw; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
                    for (int s = 0; s < 4; ++s) {
                        const uint64_t p = cbase + (uint64_t)j * 4 + s;
                        if (p < scan_lo || p >= own_hi)
                            continue;
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        c = c < sigma ? c : sigma;
                        step(c, p >= own_lo, p + 1 + P.pos_offset);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Shift-Or, NW 32-bit words per needle, one lane per needle
// (replaces [upstream] _findShiftOrSmallNeedle/_findShiftOrLargeNeedle, shiftor_matcher_restorable.hpp:38-41).
//   R = (R << 1) | mask[c];  occurrence ends here iff bit |P|-1 of R is 0.
// Same top-aligned layout: wildcard bits below `off` have mask = 0 and R = 0, so bit off-1 always shifts a 0
// ("empty prefix matches") into the needle's row 0.  Reports begin = end - |P| + 1.
// state layout: [group][NW][64].
// ---------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(256) void shiftor_brute_kernel(const brute_params P)
{
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t waves_per_wg = blockDim.x >> 6;
    const uint32_t rows = P.sigma + 1;
    uint32_t *my_mask = lds + (size_t)wave_in_wg * rows * NW * 64;

    const uint64_t n_items = (uint64_t)P.n_tiles * P.n_groups;
    const uint64_t wave_id = (uint64_t)blockIdx.x * waves_per_wg + wave_in_wg;
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_wg;

    uint32_t loaded_group = 0xFFFFFFFFu;
    int32_t my_m = 0;

    for (uint64_t item = wave_id; item < n_items; item += n_waves) {
        const uint32_t group = (uint32_t)(item % P.n_groups);
        const uint32_t tile = (uint32_t)(item / P.n_groups);
        if (group != loaded_group) {
            const uint32_t *src = P.peq + (size_t)group * rows * NW * 64;
            for (uint32_t i = lane; i < rows * NW * 64; i += 64)
                my_mask[i] = src[i];
            my_m = P.m[group * 64 + lane];
            loaded_group = group;
            __builtin_amdgcn_wave_barrier();
        }
        uint64_t own_lo = P.scan_begin + (uint64_t)tile * P.tile;
        uint64_t own_hi = own_lo + P.tile;
        if (own_hi > P.scan_end)
            own_hi = P.scan_end;
        uint64_t scan_lo = own_lo >= P.ctx_begin + P.warm ? own_lo - P.warm : P.ctx_begin;
        if (P.tile_tab) { // segmented haystacks: geometry comes from the host's tile table
            scan_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile]);
            own_lo = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 1]);
            own_hi = uniform_u64(P.tile_tab[3 * (uint64_t)tile + 2]);
        }
        const bool resume = P.has_state && tile == 0;
        if (resume)
            scan_lo = own_lo;

        uint32_t R[NW];
        if (resume) {
            const uint32_t *st = P.state_in + (size_t)group * NW * 64 + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                R[w] = st[w * 64];
        } else {
            const int32_t off = NW * 32 - my_m;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int32_t lo = off - w * 32;
                R[w] = lo <= 0 ? 0xFFFFFFFFu : (lo >= 32 ? 0u : (0xFFFFFFFFu << lo));
            }
        }
        const bool active_lane = my_m > 0;
        const uint32_t *lane_mask = my_mask + lane;
        const uint32_t sigma = P.sigma;

        auto step = [&](uint32_t c) {
            const uint32_t *row = lane_mask + (size_t)c * NW * 64;
            uint32_t prev = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const uint32_t cur = R[w];
                R[w] = alignbit(cur, prev, 31) | row[w * 64];
                prev = cur;
            }
        };

        const uint64_t a0 = scan_lo & ~3ull;
        for (uint64_t cbase = a0; cbase < own_hi; cbase += 256) {
            const uint64_t my_idx = cbase + (uint64_t)lane * 4;
            uint32_t v = 0;
            if (my_idx < own_hi)
                v = load_text_dword(P.text, my_idx, P.text_alloc);
            const uint64_t rem = own_hi - cbase;
            const bool full = (cbase >= scan_lo) && (rem >= 256);
            if (full && cbase + 256 <= own_lo) {
                for (uint32_t j = 0; j < 64; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        step(c < sigma ? c : sigma);
                    }
                }
            } else if (full && cbase >= own_lo) {
                // owned interior chunk.  Hits are tested once per 4 symbols: an occurrence ends at one of them iff
                // bit 31 is clear in the AND of the four top words; only then are the four looked at one by one.
                for (uint32_t j = 0; j < 64; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
                    uint32_t top[4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        step(c < sigma ? c : sigma);
                        top[s] = R[NW - 1];
                    }
                    const bool any = active_lane && ((int32_t)(top[0] & top[1] & top[2] & top[3]) >= 0);
                    if (__ballot(any) != 0) {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const bool hit = active_lane && ((int32_t)top[s] >= 0); // bit 31 clear
                            if (__ballot(hit) != 0)
                                brute_report(P, hit, cbase + (uint64_t)j * 4 + s + 1,
                                             cbase + (uint64_t)j * 4 + s + 1 - (uint64_t)my_m + P.pos_offset,
                                             group * 64 + lane, 0);
                        }
                    }
                }
            } else {
                const uint32_t n_dw = rem >= 256 ? 64u : (uint32_t)((rem + 3) / 4);
                for (uint32_t j = 0; j < n_dw; ++j) {
                    const uint32_t w4 = __builtin_amdgcn_readlane(v, j);
                    for (int s = 0; s < 4; ++s) {
                        const uint64_t p = cbase + (uint64_t)j * 4 + s;
                        if (p < scan_lo || p >= own_hi)
                            continue;
                        uint32_t c = (w4 >> (8 * s)) & 0xFF;
                        step(c < sigma ? c : sigma);
                        const bool hit = active_lane && ((int32_t)R[NW - 1] >= 0);
                        // an occurrence that starts before the haystack's first symbol cannot exist; with a
                        // restored state it can start in an earlier chunk, whose coordinates the caller owns
                        if (p >= own_lo && __ballot(hit) != 0)
                            brute_report(P, hit, p + 1, p + 1 - (uint64_t)my_m + P.pos_offset, group * 64 + lane, 0);
                    }
                }
            }
        }
        if (P.state_out && P.n_tiles == 1) {
            uint32_t *st = P.state_out + (size_t)group * NW * 64 + lane;
#pragma unroll
            for (int w = 0; w < NW; ++w)
                st[w * 64] = R[w];
        }
    }
}

} // namespace spm_hip
