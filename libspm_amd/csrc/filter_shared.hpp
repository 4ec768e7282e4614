// filter_shared.hpp -- what the host-side index build (index_build.hpp, compilable by plain g++) and the seed-filter
// kernels (filter.hpp) must agree on: seed plan, key hashes, table layouts, entry codes.
#pragma once

#include "hd.hpp"

namespace spm_hip
{

constexpr uint32_t kKeyMax = 16; // symbols per key: 16 whenever the seeds allow it, down to kKeyMin for short seeds
// Short keys match by chance (windows x keys / 4^H survivors).  The streaming kernel only records a survivor (16 bytes)
// and resolve_kernel disposes of it in ~0.2 ns of GPU time, so keys down to 9 symbols pay: |P| = 32, k = 2 (seeds of 10)
// with 1000 needles leaves 0.3 % of the windows -- against a ~3000x slower brute-force scan.  A set whose keys would let
// more than kMaxSurvivorShare of all windows through stays with the brute-force engine.
constexpr uint32_t kKeyMin = 9;
constexpr double kMaxSurvivorShare = 0.08;

// Seeds of one needle: n pieces of q symbols at offsets j*q.  k+1 pieces guarantee one intact piece per occurrence;
// needles with many errors get k+2 (two intact pieces on nearby diagonals), which lets the verification stage count seed
// hits per diagonal band and skip bands with a single one (candidate merging, filter.hpp).
constexpr uint32_t kMergeMinK = 8;
struct seed_plan
{
    uint32_t n, q;
};
SPM_HD inline seed_plan plan_seeds(uint32_t m, uint32_t k)
{
    const uint32_t surplus = (k >= kMergeMinK && k <= 1000 && m / (k + 2) >= kKeyMin) ? 2u : 1u;
    return {k + surplus, m / (k + surplus)};
}

template <int HV>
SPM_HD inline uint32_t bloom_hash(uint32_t key, uint32_t i)
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

SPM_HD inline chd_hashes chd_hash(uint32_t key)
{
    chd_hashes h;
    h.x = key * 0x9E3779B1u; // (two 24-bit multiplies instead of this quarter-rate one measured 8 % SLOWER on C4: the
                             // weaker mixing costs more in LDS bank conflicts than the multiply saves)
    h.s2 = (key | 1u) & 0xFFFFFFu;
    h.f = (key ^ (key >> 16)) & 0xFFFFu;
    return h;
}

SPM_HD inline uint32_t chd_slot(const chd_hashes &h, uint32_t d, uint32_t slot_mask)
{
    return ((h.x >> 3) + d * h.s2) & slot_mask;
}

SPM_HD inline uint32_t ht_hash(uint32_t key)
{
    uint32_t x = key ^ (key >> 16);
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    return x ^ (x >> 16);
}

// ---- dense passes (needle sets whose keys do not fit a few LDS fingerprint tables) ---------------------------------
// ONE pass over the text for ALL keys of the pass (hundreds of thousands): level 1 is a one-bit presence table in LDS
// (2^20 bits: ~3 bits per key, a third of the looked-up windows pass), level 1b a bucketed fingerprint table that lives
// in L2 (one 16-byte gather per window that passed the bit test: 8 slots of {1, 15-bit fingerprint}), consulted from
// inside the streaming kernel in batches of 64.  What makes that affordable is looking up few windows: every key begins
// with an ANCHOR dimer (a union of <= kDensePatterns dimer patterns, 1/8 .. of all dimers), chosen on the host so that
// every needle still has its disjoint intact-able pieces (index_build.hpp); a text window that begins with any other
// dimer cannot equal a key.
constexpr uint32_t kDensePatterns = 3;
constexpr uint32_t kDenseBloomBits = 20;          // 2^20 bits = 128 KiB of LDS
constexpr uint32_t kDenseSlots = 8;               // 16-bit slots per 16-byte bucket
constexpr uint32_t kDenseAcceptAll = 0x7FFFu;     // last slot of a bucket that overflowed at build time: every window passes

SPM_HD inline uint32_t dense_bloom_index(uint32_t key) // bit index in the presence table
{
    // every key bit takes part: bits 0..19 as they are, bits 13..31 folded onto 0..18.  (The kernel never forms this index:
    // its word address is (h >> 3) & 0x1FFFC and its bit h & 31, with h = key ^ key >> 13 -- five VALU per window.)
    return (key ^ (key >> 13)) & ((1u << kDenseBloomBits) - 1);
}
SPM_HD inline uint32_t dense_bucket(uint32_t key, uint32_t bucket_shift) { return (key * 0x9E3779B1u) >> bucket_shift; }
SPM_HD inline uint32_t dense_fp(uint32_t key) { return 0x8000u | ((key ^ (key >> 15) ^ (key >> 23)) & 0x7FFFu); }

// dimer d = sym0 | sym1 << 2 begins an anchored window iff (d ^ c) & cm == 0 for one of the patterns
SPM_HD inline bool dense_anchored(uint32_t d, const uint32_t *c, const uint32_t *cm, uint32_t n_pat)
{
    for (uint32_t i = 0; i < n_pat; ++i)
        if (((d ^ c[i]) & cm[i]) == 0)
            return true;
    return false;
}

// Range code of an exact-table entry (.z):
//   kRngSingle | r (5 bits) | ns << 5 [| kRngWhole]
//                 the key sits at this one offset of the needle; its window starts r symbols into its seed; .y = signature:
//                 the first ns (<= 16) symbols of the rest of the seed; kRngWhole: that IS the whole rest (the seed is
//                 key_len + ns symbols long)
//   kRngRun | span
//                 MANY offsets of the needle share the key (the needle IS a repeat there): no per-offset checks -- they
//                 would pass wherever the text carries the same repeat --, one pair counted into the bands of the
//                 diagonals t - offset - span .. t - offset
constexpr uint32_t kRngRun = 0x8000u;
constexpr uint32_t kRngSingle = 0x2000u;
constexpr uint32_t kRngWhole = 0x1000u;
constexpr uint32_t kSeedChecked = 0x4000u; // (queue only) the signature already showed the whole seed in the text

} // namespace spm_hip
