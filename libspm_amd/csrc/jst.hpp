// jst.hpp -- journaled-sequence (pan-genome) search on the device (config C5).
//
// Contract: the hit set equals the union over haplotypes of a linear scan of each materialised haplotype
// (haplotype, position in haplotype coordinates).  The reference holds no traversal code, only the journaled-sequence
// design (specs/journaled_sequence_class_diagram.drawio) and the matcher hooks (matcher/concept.hpp:26-161).
//
// Scheme (everything below runs on the MI355X; the host only launches):
//   1. the reference axis is cut into blocks of L positions; block j owns the haplotype symbols that stem from
//      reference positions [jL, (j+1)L) (an alt symbol belongs to its allele's position);
//   2. hap_start[j][h] = haplotype coordinate of the first symbol block j owns in haplotype h (column-wise prefix sum
//      of the allele length deltas each haplotype carries);
//   3. a context = the symbols a (block, haplotype) pair owns plus window-1 symbols of left context.  Its signature is
//      exact: where the left context starts (reference position, or allele + offset inside an alt), its length, and
//      the bit set of the alleles the haplotype carries over that stretch.  Equal signatures <=> byte-identical
//      contexts, so one workgroup per (block, group of <= 1024 haplotypes) groups them by signature in LDS;
//   4. one representative per group is spelled out into the context buffer (wave-cooperative copies of reference
//      runs and alt runs), the buffer is scanned as independent segments by the ordinary engines, and each hit whose
//      last symbol lies in the owned part is fanned out to the haplotypes of its group.
// By the window property (a hit depends only on the window_size symbols ending at it) this is exact; work on the
// device is proportional to the distinct sequence content, not to haplotypes x length.
#pragma once

#include <hipcub/hipcub.hpp>

// the ctypes binding (libspm_amd/capi.py) mirrors these layouts
static_assert(sizeof(spm_jst_allele) == 24 && sizeof(spm_jst_hit) == 24 && sizeof(spm_jst_stats) == 104, "C ABI layout");

namespace spm_hip
{

constexpr uint32_t kJstGroup = 1024;    // haplotypes whose contexts are compared with each other (signature table in LDS);
                                        // larger trees are cut into groups of this size, contexts are shared within a group
constexpr uint32_t kJstMaxHap = 65535;  // haplotype indices are reported as uint32, group-local ids are 15 bits
constexpr uint32_t kJstMaskWords = 8;   // 256 alleles per context signature; denser stretches are not shared
constexpr uint32_t kJstSigWords = 4 + kJstMaskWords;
constexpr uint16_t kJstNone = 0xFFFF;   // (block, haplotype) owns no symbol

struct jst_dev
{
    const uint8_t *ref;
    uint64_t n_ref;
    const uint64_t *pos;
    const uint32_t *rlen, *alen;
    const uint64_t *aoff;
    const uint8_t *alt;
    const uint64_t *cov;
    uint64_t n_alleles;
    uint32_t cw, n_hap, max_rlen, window;
    uint32_t n_groups;    // ceil(n_hap / kJstGroup)
    uint64_t L, n_blocks; // all blocks of the reference
    uint64_t jb, je;      // indexed blocks
    const uint64_t *a_lo; // [n_blocks + 1]: first allele with pos >= j * L
    uint64_t *hap_start;  // [(n_blocks + 1) * n_hap]
};

__device__ __forceinline__ bool jst_carried(const jst_dev &J, uint64_t i, uint32_t h)
{
    return (J.cov[i * J.cw + (h >> 6)] >> (h & 63)) & 1ull;
}

__global__ void jst_alo_kernel(const uint64_t *pos, uint64_t n_alleles, uint64_t L, uint64_t n_blocks, uint64_t *a_lo)
{
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j > n_blocks)
        return;
    const uint64_t key = j * L;
    uint64_t lo = 0, hi = n_alleles;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (pos[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    a_lo[j] = lo;
}

__global__ void jst_widen_kernel(const uint32_t *in, uint64_t n, uint64_t *out)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        out[t] = in[t];
}

// delta[j][h] = sum over the alleles of block j that h carries of (alt_len - ref_len); row n_blocks = 0
__global__ void jst_delta_kernel(jst_dev J, int64_t *delta)
{
    const uint64_t j = blockIdx.x;
    const uint64_t a0 = J.a_lo[j], a1 = J.a_lo[j + 1];
    for (uint32_t h = threadIdx.x; h < J.n_hap; h += blockDim.x) {
        int64_t d = 0;
        for (uint64_t i = a0; i < a1; ++i)
            if (jst_carried(J, i, h))
                d += (int64_t)J.alen[i] - (int64_t)J.rlen[i];
        delta[j * J.n_hap + h] = d;
    }
}

// column-wise exclusive scan over blocks, chunks of `chunk` rows
__global__ void jst_chunk_sum_kernel(const int64_t *delta, uint64_t n_rows, uint32_t n_hap, uint32_t chunk, int64_t *csum)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_chunks = (n_rows + chunk - 1) / chunk;
    if (t >= n_chunks * n_hap)
        return;
    const uint64_t c = t / n_hap, h = t % n_hap;
    int64_t s = 0;
    for (uint64_t j = c * chunk; j < std::min<uint64_t>(n_rows, (c + 1) * chunk); ++j)
        s += delta[j * n_hap + h];
    csum[c * n_hap + h] = s;
}

__global__ void jst_chunk_scan_kernel(int64_t *csum, uint64_t n_chunks, uint32_t n_hap)
{
    const uint32_t h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= n_hap)
        return;
    int64_t run = 0;
    for (uint64_t c = 0; c < n_chunks; ++c) {
        const int64_t t = csum[c * n_hap + h];
        csum[c * n_hap + h] = run;
        run += t;
    }
}

// delta -> exclusive prefix (in place), rows 0..n_rows inclusive (row n_rows receives the total)
__global__ void jst_chunk_apply_kernel(int64_t *delta, uint64_t n_rows, uint32_t n_hap, uint32_t chunk, const int64_t *csum)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n_chunks = (n_rows + chunk - 1) / chunk;
    if (t >= n_chunks * n_hap)
        return;
    const uint64_t c = t / n_hap, h = t % n_hap;
    int64_t run = csum[c * n_hap + h];
    const uint64_t j1 = std::min<uint64_t>(n_rows, (c + 1) * chunk);
    for (uint64_t j = c * chunk; j < j1; ++j) {
        const int64_t d = delta[j * n_hap + h];
        delta[j * n_hap + h] = run;
        run += d;
    }
    if (j1 == n_rows)
        delta[n_rows * n_hap + h] = run;
}

// first reference position at or after jL that haplotype h still reads (a carried deletion may span the block border)
__device__ __forceinline__ uint64_t jst_first_ref(const jst_dev &J, uint64_t j, uint32_t h)
{
    const uint64_t r0 = j * J.L;
    if (j >= J.n_blocks)
        return J.n_ref;
    for (int64_t i = (int64_t)J.a_lo[j] - 1; i >= 0 && J.pos[i] + J.max_rlen > r0; --i)
        if (jst_carried(J, (uint64_t)i, h))
            return std::max<uint64_t>(r0, J.pos[i] + J.rlen[i]);
    return r0;
}

// shift (in place) -> hap_start
__global__ void jst_start_kernel(jst_dev J)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (J.n_blocks + 1) * J.n_hap)
        return;
    const uint64_t j = t / J.n_hap;
    const uint32_t h = (uint32_t)(t % J.n_hap);
    const int64_t shift = (int64_t)J.hap_start[t];
    J.hap_start[t] = (uint64_t)((int64_t)jst_first_ref(J, j, h) + shift);
}

// Where the context of (block j, haplotype h) starts and which alleles it spans.
struct jst_walk
{
    uint32_t sig[kJstSigWords]; // [0..1] start point, [2] first reference position - jL, [3] length, [4..] allele bits
    uint64_t start_ref;         // kind 0: reference position of the first symbol
    uint64_t start_allele;      // kind 1: allele whose alt holds the first symbol ...
    uint32_t start_off;         // ... at this offset
    uint32_t kind;              // 0 reference run, 1 inside an alt
    uint64_t next_allele;       // first allele index the forward walk examines
    uint64_t lo;                // haplotype coordinate of the first symbol
    uint32_t len, owned_from;   // context length, offset of the first owned symbol
    bool empty;
};

__device__ inline void jst_left_walk(const jst_dev &J, uint64_t j, uint32_t h, jst_walk &W)
{
    const uint64_t a = J.hap_start[j * J.n_hap + h], b = J.hap_start[(j + 1) * J.n_hap + h];
    W.empty = b <= a;
    for (uint32_t w = 0; w < kJstSigWords; ++w)
        W.sig[w] = 0;
    if (W.empty)
        return;
    const uint64_t a_hi = J.a_lo[j + 1];
    bool overflow = false;
    auto set_bit = [&](uint64_t idx) {
        const uint64_t t = a_hi - 1 - idx;
        if (t >= 32ull * kJstMaskWords) {
            overflow = true;
        } else {
            const uint32_t bit = 1u << (t & 31);
#pragma unroll
            for (uint32_t w = 0; w < kJstMaskWords; ++w) // static indices: the signature stays in registers
                if ((uint32_t)(t >> 5) == w)
                    W.sig[4 + w] |= bit;
        }
    };
    for (uint64_t i = J.a_lo[j]; i < a_hi; ++i)
        if (jst_carried(J, i, h))
            set_bit(i);
    uint64_t r = jst_first_ref(J, j, h);
    const uint64_t r_first = r;
    uint64_t rem = std::min<uint64_t>(J.window ? J.window - 1 : 0, a);
    W.owned_from = (uint32_t)rem;
    W.lo = a - rem;
    W.len = (uint32_t)(rem + (b - a));
    int64_t i = (int64_t)J.a_lo[j] - 1;
    W.kind = 0;
    W.start_off = 0;
    W.start_allele = 0;
    while (true) {
        if (rem == 0) {
            W.start_ref = r;
            W.next_allele = (uint64_t)(i + 1);
            break;
        }
        while (i >= 0 && !jst_carried(J, (uint64_t)i, h))
            --i;
        if (i < 0) { // reference prefix (a >= rem guarantees r >= rem)
            W.start_ref = r - rem;
            W.next_allele = 0;
            break;
        }
        const uint64_t e = J.pos[i] + J.rlen[i];
        const uint64_t run = r - e;
        if (rem <= run) {
            W.start_ref = r - rem;
            W.next_allele = (uint64_t)(i + 1);
            break;
        }
        rem -= run;
        set_bit((uint64_t)i);
        const uint64_t al = J.alen[i];
        if (rem <= al) {
            W.kind = 1;
            W.start_allele = (uint64_t)i;
            W.start_off = (uint32_t)(al - rem);
            W.next_allele = (uint64_t)i;
            break;
        }
        rem -= al;
        r = J.pos[i];
        --i;
    }
    const uint64_t sp = W.kind ? ((1ull << 63) | (W.start_allele << 24) | W.start_off) : W.start_ref;
    W.sig[0] = (uint32_t)sp;
    W.sig[1] = (uint32_t)(sp >> 32);
    W.sig[2] = (uint32_t)(r_first - j * J.L);
    W.sig[3] = W.len;
    if (overflow) { // too many alleles for the bit set: this context is not shared
        W.sig[0] = h;
        W.sig[1] = 0xC0000000u;
    }
}

struct jst_index_out
{
    uint16_t *local_id;           // [(je - jb) * n_hap]: group id | 0x8000 for the group's representative
    uint32_t *n_uniq;             // [je - jb]
    unsigned long long *bytes;    // [je - jb]
    unsigned long long *totals;   // [0] non-empty contexts, [1] owned symbols
};

// One workgroup per block: signatures in LDS, grouped through an LDS hash table with exact comparison.
__global__ __launch_bounds__(256) void jst_dedupe_kernel(jst_dev J, jst_index_out O)
{
    extern __shared__ uint32_t lds[];
    // this workgroup: block jr of the indexed range, haplotypes [h0, h0 + H)
    const uint64_t jr = blockIdx.x / J.n_groups;
    const uint32_t h0 = (uint32_t)(blockIdx.x % J.n_groups) * kJstGroup;
    const uint32_t H = min(kJstGroup, J.n_hap - h0);
    uint32_t n_slots = 64;
    while (n_slots < 2 * H)
        n_slots <<= 1;
    uint32_t *sig = lds;                       // [H][kJstSigWords]
    uint32_t *slots = sig + H * kJstSigWords;  // [n_slots] owner haplotype
    uint32_t *uid = slots + n_slots;           // [n_slots] group id of the slot
    uint32_t *slot_of = uid + n_slots;         // [H]
    __shared__ uint32_t s_n;
    __shared__ unsigned long long s_bytes, s_ctx, s_owned;
    const uint64_t j = J.jb + jr;
    if (threadIdx.x == 0) {
        s_n = 0;
        s_bytes = 0;
        s_ctx = 0;
        s_owned = 0;
    }
    for (uint32_t s = threadIdx.x; s < n_slots; s += blockDim.x)
        slots[s] = 0xFFFFFFFFu;
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < H; h += blockDim.x) {
        jst_walk W;
        jst_left_walk(J, j, h0 + h, W);
        for (uint32_t w = 0; w < kJstSigWords; ++w)
            sig[h * kJstSigWords + w] = W.sig[w];
        slot_of[h] = W.empty ? 0xFFFFFFFFu : 0;
        if (!W.empty) {
            atomicAdd(&s_ctx, 1ull);
            atomicAdd(&s_owned, (unsigned long long)(W.len - W.owned_from));
        }
    }
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < H; h += blockDim.x) {
        if (slot_of[h] == 0xFFFFFFFFu)
            continue;
        uint32_t x = 0x9E3779B9u;
        for (uint32_t w = 0; w < kJstSigWords; ++w) {
            x ^= sig[h * kJstSigWords + w] + 0x7F4A7C15u + (x << 6) + (x >> 2);
            x *= 0x85EBCA6Bu;
        }
        uint32_t s = (x ^ (x >> 15)) & (n_slots - 1);
        while (true) {
            uint32_t owner = atomicCAS(&slots[s], 0xFFFFFFFFu, h);
            if (owner == 0xFFFFFFFFu)
                owner = h;
            bool same = true;
            if (owner != h)
                for (uint32_t w = 0; w < kJstSigWords; ++w)
                    same = same && sig[owner * kJstSigWords + w] == sig[h * kJstSigWords + w];
            if (same) {
                slot_of[h] = s;
                break;
            }
            s = (s + 1) & (n_slots - 1);
        }
    }
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < H; h += blockDim.x) {
        const uint32_t s = slot_of[h];
        if (s != 0xFFFFFFFFu && slots[s] == h) {
            uid[s] = atomicAdd(&s_n, 1u);
            atomicAdd(&s_bytes, (unsigned long long)sig[h * kJstSigWords + 3]);
        }
    }
    __syncthreads();
    for (uint32_t h = threadIdx.x; h < H; h += blockDim.x) {
        const uint32_t s = slot_of[h];
        uint16_t v = kJstNone;
        if (s != 0xFFFFFFFFu)
            v = (uint16_t)(uid[s] | (slots[s] == h ? 0x8000u : 0u));
        O.local_id[jr * J.n_hap + h0 + h] = v;
    }
    if (threadIdx.x == 0) {
        O.n_uniq[blockIdx.x] = s_n;
        O.bytes[blockIdx.x] = s_bytes;
        atomicAdd(&O.totals[0], s_ctx);
        atomicAdd(&O.totals[1], s_owned);
    }
}

struct jst_emit_out
{
    const uint16_t *local_id;
    const uint64_t *ctx_base;  // [je - jb + 1] exclusive scan of n_uniq
    const uint64_t *byte_base; // [je - jb + 1] exclusive scan of bytes
    uint64_t *ctx_off;         // [n_ctx + 1]
    uint32_t *ctx_block;       // [n_ctx] block index - jb
    uint32_t *ctx_owned;       // [n_ctx] offset of the first owned symbol
    uint8_t *buffer;
};

// One workgroup per block: the representatives' contexts are laid out in group-id order and spelled out, one wave per
// context, lanes copying each reference / alt run side by side.
__global__ __launch_bounds__(256) void jst_emit_kernel(jst_dev J, jst_emit_out O)
{
    extern __shared__ uint32_t lds[];
    const uint64_t cb = blockIdx.x; // (block, haplotype group) cell
    const uint64_t jr = cb / J.n_groups, j = J.jb + jr;
    const uint32_t h0 = (uint32_t)(cb % J.n_groups) * kJstGroup;
    const uint32_t H = min(kJstGroup, J.n_hap - h0);
    uint32_t *rep = lds;        // [n_uniq] representative haplotype of group id
    uint32_t *len = rep + kJstGroup;    // [n_uniq]
    uint32_t *off = len + kJstGroup;    // [n_uniq] byte offset inside the cell's stretch (fits: <= H * (L + window + alts))
    const uint32_t n_uniq = (uint32_t)(O.ctx_base[cb + 1] - O.ctx_base[cb]);
    if (n_uniq == 0)
        return;
    for (uint32_t h = threadIdx.x; h < H; h += blockDim.x) {
        const uint16_t v = O.local_id[jr * J.n_hap + h0 + h];
        if (v != kJstNone && (v & 0x8000u))
            rep[v & 0x7FFFu] = h0 + h;
    }
    __syncthreads();
    for (uint32_t u = threadIdx.x; u < n_uniq; u += blockDim.x) {
        jst_walk W;
        jst_left_walk(J, j, rep[u], W);
        len[u] = W.len;
        O.ctx_block[O.ctx_base[cb] + u] = (uint32_t)cb;
        O.ctx_owned[O.ctx_base[cb] + u] = W.owned_from;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t u = 0; u < n_uniq; ++u) {
            off[u] = run;
            run += len[u];
        }
    }
    __syncthreads();
    for (uint32_t u = threadIdx.x; u < n_uniq; u += blockDim.x)
        O.ctx_off[O.ctx_base[cb] + u] = O.byte_base[cb] + off[u];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    for (uint32_t u = wave; u < n_uniq; u += n_waves) {
        const uint32_t h = rep[u];
        jst_walk W; // wave-uniform: every lane walks the same haplotype
        jst_left_walk(J, j, h, W);
        uint8_t *out = O.buffer + O.byte_base[cb] + off[u];
        uint64_t remaining = W.len;
        uint64_t i = W.next_allele, r = W.start_ref;
        bool in_alt = W.kind == 1;
        uint32_t aoff_in = W.start_off;
        while (remaining > 0) {
            if (in_alt && i >= J.n_alleles)
                break; // cannot happen for a validated allele table; never read past it
            if (in_alt) {
                const uint64_t n = std::min<uint64_t>(remaining, (uint64_t)J.alen[i] - aoff_in);
                const uint8_t *src = J.alt + J.aoff[i] + aoff_in;
                for (uint64_t x = lane; x < n; x += 64)
                    out[x] = src[x];
                out += n;
                remaining -= n;
                r = J.pos[i] + J.rlen[i];
                ++i;
                in_alt = false;
                aoff_in = 0;
            } else {
                while (i < J.n_alleles && !jst_carried(J, i, h))
                    ++i;
                const uint64_t run_end = i < J.n_alleles ? J.pos[i] : J.n_ref;
                const uint64_t n = std::min<uint64_t>(remaining, run_end - r);
                const uint8_t *src = J.ref + r;
                for (uint64_t x = lane; x < n; x += 64)
                    out[x] = src[x];
                out += n;
                remaining -= n;
                r += n;
                in_alt = true; // only entered when remaining > 0, i.e. the run ended at a carried allele
            }
        }
    }
}

struct jst_fan_params
{
    const spm_hit *hits;
    uint64_t n_hits;                         // segment hits, if the host knows their number ...
    const unsigned long long *n_hits_dev;    // ... else where the scan counts them (launched before the host has read it)
    uint64_t hit_cap;                        //     and the capacity of `hits`
    const uint64_t *ctx_off;
    uint64_t n_ctx;
    const uint32_t *ctx_block, *ctx_owned;
    const uint64_t *ctx_base;
    const uint16_t *local_id;
    const int32_t *m; // needle lengths
    uint32_t report_begin;
    spm_jst_hit *out;
    unsigned long long *out_count;
    uint64_t out_cap;
};

// the context ids of 8 consecutive haplotypes of one block: one 16-byte load when the row allows it
__device__ __forceinline__ void jst_load_ids8(const uint16_t *p, uint32_t n_valid, uint16_t (&v)[8])
{
    if (n_valid >= 8 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const uint4 q = *reinterpret_cast<const uint4 *>(p);
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = (uint16_t)(w[i >> 1] >> (16 * (i & 1)));
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            v[i] = (uint32_t)i < n_valid ? p[i] : kJstNone;
    }
}

// One thread per segment hit: drop it if its last symbol lies in the left context, else report it for every haplotype
// of the context's group, in haplotype coordinates.  Output slots are drawn with one atomic per wave (a per-record
// atomic on the single counter cost 0.6 ms for 10^6 records).
__global__ void jst_fanout_kernel(jst_dev J, jst_fan_params F)
{
    const uint32_t lane = threadIdx.x & 63;
    uint64_t n_hits = F.n_hits;
    if (F.n_hits_dev) {
        const unsigned long long n = *F.n_hits_dev;
        n_hits = n < F.hit_cap ? n : F.hit_cap;
    }
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t rounds = (n_hits + stride - 1) / stride; // wave-uniform trip count: the slot reservation is wave-collective
    for (uint64_t rd = 0; rd < rounds; ++rd) {
    const uint64_t t = rd * stride + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // pass 1: which context, and how many haplotypes share it
    bool live = t < n_hits;
    spm_hit hit{};
    uint64_t c = 0, local = 0, jr = 0;
    uint32_t id = 0, members = 0, h_lo = 0, h_hi = 0;
    if (live) {
        hit = F.hits[t];
        const uint64_t probe = F.report_begin ? hit.pos : hit.pos - 1; // a symbol of the hit's own context
        uint64_t lo = 0, hi = F.n_ctx;                                  // ctx_off[lo] <= probe < ctx_off[hi]
        while (hi - lo > 1) {
            const uint64_t mid = (lo + hi) >> 1;
            if (F.ctx_off[mid] <= probe)
                lo = mid;
            else
                hi = mid;
        }
        c = lo;
        local = hit.pos - F.ctx_off[c];
        const uint64_t last = F.report_begin ? local + (uint64_t)F.m[hit.pattern] - 1 : local - 1;
        live = last >= F.ctx_owned[c];
    }
    if (live) {
        const uint64_t cb = F.ctx_block[c]; // (block, haplotype group) cell of the context
        jr = cb / J.n_groups;
        h_lo = (uint32_t)(cb % J.n_groups) * kJstGroup;
        h_hi = min(J.n_hap, h_lo + kJstGroup);
        id = (uint32_t)(c - F.ctx_base[cb]);
        for (uint32_t h = h_lo; h < h_hi; h += 8) {
            uint16_t v[8];
            jst_load_ids8(F.local_id + jr * J.n_hap + h, h_hi - h, v);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                members += (v[i] != kJstNone && (uint32_t)(v[i] & 0x7FFFu) == id) ? 1u : 0u;
        }
    }
    // one atomic per wave: exclusive prefix of the member counts over the lanes
    uint32_t incl = members;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(incl, o);
        if ((int)lane >= o)
            incl += up;
    }
    const uint32_t total = __shfl(incl, 63);
    unsigned long long base = 0;
    if (lane == 0 && total)
        base = atomicAdd(F.out_count, (unsigned long long)total);
    base = __shfl(base, 0);
    if (!live || members == 0)
        continue;
    // pass 2: report the hit for every haplotype of the group, in haplotype coordinates
    unsigned long long slot = base + (incl - members);
    const uint64_t j = J.jb + jr;
    // (eight haplotypes at a time: their start coordinates are loaded together, then the records are stored -- one at a
    // time the compiler has to keep every load behind the previous record's store, a dependent round trip per member)
    for (uint32_t h = h_lo; h < h_hi; h += 8) {
        uint16_t v[8];
        jst_load_ids8(F.local_id + jr * J.n_hap + h, h_hi - h, v);
        bool mem[8];
        uint64_t a[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mem[i] = v[i] != kJstNone && (uint32_t)(v[i] & 0x7FFFu) == id;
            a[i] = mem[i] ? J.hap_start[j * J.n_hap + h + i] : 0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!mem[i])
                continue;
            const uint64_t ctx_lo = a[i] - std::min<uint64_t>(J.window ? J.window - 1 : 0, a[i]);
            if (slot < F.out_cap) {
                spm_jst_hit o;
                o.pos = ctx_lo + local;
                o.haplotype = h + i;
                o.pattern = hit.pattern;
                o.score = hit.score;
                o.reserved = 0;
                F.out[slot] = o;
            }
            ++slot;
        }
    }
    }
}

} // namespace spm_hip

// ----------------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------------
struct spm_jst
{
    spm_ctx *ctx = nullptr;
    const spm_text *ref = nullptr;
    std::vector<spm_jst_allele> al;
    std::vector<uint8_t> alt;
    std::vector<uint64_t> cov;
    uint32_t H = 0, cw = 0, max_rlen = 0;
    // device allele table
    uint64_t *d_pos = nullptr, *d_aoff = nullptr, *d_cov = nullptr;
    uint32_t *d_rlen = nullptr, *d_alen = nullptr;
    uint8_t *d_alt = nullptr;
    // index
    bool indexed = false;
    uint32_t window = 0;
    uint64_t L = 0, n_blocks = 0, jb = 0, je = 0;
    uint64_t *d_alo = nullptr, *d_hap_start = nullptr;
    uint16_t *d_local_id = nullptr;
    uint64_t n_ctx = 0, ctx_bytes = 0;
    uint64_t *d_ctx_off = nullptr, *d_ctx_base = nullptr, *d_byte_base = nullptr;
    uint32_t *d_ctx_block = nullptr, *d_ctx_owned = nullptr;
    spm_text *ctx_text = nullptr;
    spm_jst_stats stats{};
    // (record buffers of the searches are recycled through the context: a 200 MB hipMalloc / hipFree pair per search
    // cost ~0.3 ms)
    unsigned long long *d_fan_count = nullptr;
    uint64_t seg_hit_hint = 0; // most segment hits a search has seen (sizes the grid of the fan-out launched behind the scan)
    hipEvent_t fan_ev[2] = {nullptr, nullptr};

    spm_hip::jst_dev dev() const
    {
        spm_hip::jst_dev J{};
        J.ref = ref->d;
        J.n_ref = ref->n;
        J.pos = d_pos;
        J.rlen = d_rlen;
        J.alen = d_alen;
        J.aoff = d_aoff;
        J.alt = d_alt;
        J.cov = d_cov;
        J.n_alleles = al.size();
        J.cw = cw;
        J.n_hap = H;
        J.n_groups = (H + spm_hip::kJstGroup - 1) / spm_hip::kJstGroup;
        J.max_rlen = max_rlen;
        J.window = window;
        J.L = L;
        J.n_blocks = n_blocks;
        J.jb = jb;
        J.je = je;
        J.a_lo = d_alo;
        J.hap_start = d_hap_start;
        return J;
    }
    void free_index()
    {
        hipFree(d_alo);
        hipFree(d_hap_start);
        hipFree(d_local_id);
        hipFree(d_ctx_off);
        hipFree(d_ctx_base);
        hipFree(d_byte_base);
        hipFree(d_ctx_block);
        hipFree(d_ctx_owned);
        d_alo = d_hap_start = d_ctx_off = d_ctx_base = d_byte_base = nullptr;
        d_local_id = nullptr;
        d_ctx_block = d_ctx_owned = nullptr;
        if (ctx_text)
            spm_hip_text_destroy(ctx_text);
        ctx_text = nullptr;
        indexed = false;
    }
};

struct spm_jst_hits
{
    spm_ctx *ctx = nullptr;
    spm_jst_hit *d = nullptr;
    uint64_t cap = 0;
    uint64_t n = 0;
    bool sorted = false;
    std::vector<spm_jst_hit> host;
};

extern "C" int spm_hip_jst_create(spm_ctx *ctx, const spm_text *reference, const spm_jst_allele *alleles,
                                  uint64_t n_alleles, const uint8_t *alt_pool, uint64_t alt_pool_len,
                                  const uint64_t *coverage, uint32_t n_haplotypes, spm_jst **out)
{
    if (!ctx || !reference || !out || (n_alleles && (!alleles || !coverage)) || n_haplotypes == 0) {
        SPM_SET_ERR(ctx, "spm_hip_jst_create: invalid argument");
        return SPM_E_INVALID;
    }
    if (n_haplotypes > spm_hip::kJstMaxHap) {
        SPM_SET_ERR(ctx, "spm_hip_jst_create: %u haplotypes, at most %u", n_haplotypes, spm_hip::kJstMaxHap);
        return SPM_E_UNSUPPORTED;
    }
    const uint32_t cw = (n_haplotypes + 63) / 64;
    std::unique_ptr<spm_jst> J(new spm_jst);
    J->ctx = ctx;
    J->ref = reference;
    J->H = n_haplotypes;
    J->cw = cw;
    J->al.assign(alleles, alleles + n_alleles);
    J->cov.assign(coverage, coverage + n_alleles * cw);
    if (alt_pool_len)
        J->alt.assign(alt_pool, alt_pool + alt_pool_len);
    for (uint64_t i = 0; i < n_alleles; ++i) {
        spm_jst_allele &a = J->al[i];
        if (a.pos > reference->n || (i && a.pos < J->al[i - 1].pos) || a.alt_off + a.alt_len > alt_pool_len ||
            a.alt_len >= (1u << 24)) {
            SPM_SET_ERR(ctx, "spm_hip_jst_create: allele %llu is out of order, out of range or too long",
                        (unsigned long long)i);
            return SPM_E_INVALID;
        }
        a.ref_len = (uint32_t)std::min<uint64_t>(a.ref_len, reference->n - a.pos); // clamp to the reference
        J->max_rlen = std::max(J->max_rlen, a.ref_len);
        for (uint32_t x = 0; x < a.alt_len; ++x)
            if (J->alt[a.alt_off + x] >= reference->sigma) {
                SPM_SET_ERR(ctx, "spm_hip_jst_create: allele %llu holds a symbol that is not a rank < sigma",
                            (unsigned long long)i);
                return SPM_E_INVALID;
            }
        if (n_haplotypes & 63)
            J->cov[i * cw + cw - 1] &= (1ull << (n_haplotypes & 63)) - 1;
    }
    // alleles that overlap on the reference must not share a haplotype
    for (uint64_t i = 0; i < n_alleles; ++i) {
        const uint64_t e = J->al[i].pos + J->al[i].ref_len;
        for (uint64_t j = i + 1; j < n_alleles && J->al[j].pos < e; ++j)
            for (uint32_t w = 0; w < cw; ++w)
                if (J->cov[i * cw + w] & J->cov[j * cw + w]) {
                    SPM_SET_ERR(ctx, "spm_hip_jst_create: alleles %llu and %llu overlap on a shared haplotype",
                                (unsigned long long)i, (unsigned long long)j);
                    return SPM_E_UNSUPPORTED;
                }
    }
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t na = std::max<uint64_t>(n_alleles, 1);
    std::vector<uint64_t> pos(na, 0), aoff(na, 0);
    std::vector<uint32_t> rlen(na, 0), alen(na, 0);
    for (uint64_t i = 0; i < n_alleles; ++i) {
        pos[i] = J->al[i].pos;
        aoff[i] = J->al[i].alt_off;
        rlen[i] = J->al[i].ref_len;
        alen[i] = J->al[i].alt_len;
    }
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_pos, na * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_aoff, na * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_rlen, na * 4));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_alen, na * 4));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_cov, na * cw * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_alt, std::max<uint64_t>(alt_pool_len, 1) + 64));
    SPM_HIP_CHECK(ctx, hipMemcpy(J->d_pos, pos.data(), na * 8, hipMemcpyHostToDevice));
    SPM_HIP_CHECK(ctx, hipMemcpy(J->d_aoff, aoff.data(), na * 8, hipMemcpyHostToDevice));
    SPM_HIP_CHECK(ctx, hipMemcpy(J->d_rlen, rlen.data(), na * 4, hipMemcpyHostToDevice));
    SPM_HIP_CHECK(ctx, hipMemcpy(J->d_alen, alen.data(), na * 4, hipMemcpyHostToDevice));
    if (n_alleles)
        SPM_HIP_CHECK(ctx, hipMemcpy(J->d_cov, J->cov.data(), n_alleles * cw * 8, hipMemcpyHostToDevice));
    if (alt_pool_len)
        SPM_HIP_CHECK(ctx, hipMemcpy(J->d_alt, J->alt.data(), alt_pool_len, hipMemcpyHostToDevice));
    *out = J.release();
    return SPM_OK;
}

extern "C" void spm_hip_jst_destroy(spm_jst *J)
{
    if (!J)
        return;
    hipSetDevice(J->ctx->device);
    hipStreamSynchronize(J->ctx->stream);
    J->free_index();
    hipFree(J->d_fan_count);
    if (J->fan_ev[0]) {
        hipEventDestroy(J->fan_ev[0]);
        hipEventDestroy(J->fan_ev[1]);
    }
    hipFree(J->d_pos);
    hipFree(J->d_aoff);
    hipFree(J->d_rlen);
    hipFree(J->d_alen);
    hipFree(J->d_cov);
    hipFree(J->d_alt);
    delete J;
}

extern "C" uint64_t spm_hip_jst_haplotype_length(const spm_jst *J, uint32_t h)
{
    if (!J || h >= J->H)
        return 0;
    int64_t len = (int64_t)J->ref->n;
    for (size_t i = 0; i < J->al.size(); ++i)
        if ((J->cov[i * J->cw + (h >> 6)] >> (h & 63)) & 1)
            len += (int64_t)J->al[i].alt_len - (int64_t)J->al[i].ref_len;
    return (uint64_t)len;
}

extern "C" int spm_hip_jst_extract(spm_jst *J, uint32_t h, uint64_t begin, uint64_t n, uint8_t *out)
{
    if (!J || h >= J->H || (n && !out)) {
        SPM_SET_ERR(J ? J->ctx : nullptr, "spm_hip_jst_extract: invalid argument");
        return SPM_E_INVALID;
    }
    spm_ctx *ctx = J->ctx;
    const uint64_t end = begin + n;
    uint64_t r = 0, hp = 0; // reference cursor, haplotype coordinate of reference[r]
    auto ref_piece = [&](uint64_t hp0, uint64_t r0, uint64_t len) -> int { // haplotype [hp0, hp0+len) = reference[r0..]
        const uint64_t lo = std::max(hp0, begin), hi = std::min(hp0 + len, end);
        if (lo < hi)
            return spm_hip_text_download(ctx, J->ref, r0 + (lo - hp0), hi - lo, out + (lo - begin));
        return SPM_OK;
    };
    for (size_t i = 0; i < J->al.size() && hp < end; ++i) {
        if (!((J->cov[i * J->cw + (h >> 6)] >> (h & 63)) & 1))
            continue;
        const spm_jst_allele &a = J->al[i];
        const uint64_t run = a.pos - r;
        int rc = ref_piece(hp, r, run);
        if (rc != SPM_OK)
            return rc;
        hp += run;
        const uint64_t lo = std::max(hp, begin), hi = std::min(hp + a.alt_len, end);
        for (uint64_t x = lo; x < hi; ++x)
            out[x - begin] = J->alt[a.alt_off + (x - hp)];
        hp += a.alt_len;
        r = a.pos + a.ref_len;
    }
    if (hp < end) {
        if (hp + (J->ref->n - r) < end) {
            SPM_SET_ERR(ctx, "spm_hip_jst_extract: range beyond the end of haplotype %u", h);
            return SPM_E_INVALID;
        }
        int rc = ref_piece(hp, r, J->ref->n - r);
        if (rc != SPM_OK)
            return rc;
    }
    return SPM_OK;
}

extern "C" int spm_hip_jst_index(spm_jst *J, uint32_t window, uint32_t block_len, uint64_t block_begin,
                                 uint64_t block_end)
{
    using namespace spm_hip;
    if (!J || window == 0) {
        SPM_SET_ERR(J ? J->ctx : nullptr, "spm_hip_jst_index: invalid argument");
        return SPM_E_INVALID;
    }
    spm_ctx *ctx = J->ctx;
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    J->free_index();
    const uint64_t L = block_len ? block_len : std::max<uint64_t>(256, ((uint64_t)window + 255) & ~255ull);
    const uint64_t n_blocks = std::max<uint64_t>(1, (J->ref->n + L - 1) / L);
    const uint64_t jb = std::min(block_begin, n_blocks), je = block_end ? std::min(block_end, n_blocks) : n_blocks;
    if (jb > je || (uint64_t)window + L + J->max_rlen >= (1ull << 31)) {
        SPM_SET_ERR(ctx, "spm_hip_jst_index: bad block range or window");
        return SPM_E_INVALID;
    }
    J->window = window;
    J->L = L;
    J->n_blocks = n_blocks;
    J->jb = jb;
    J->je = je;
    const uint32_t H = J->H;
    const uint64_t nb = je - jb;
    hipStream_t st = ctx->stream;
    dev_scratch tmp_bufs; // temporaries of the build, released on every return path
    struct event_pair
    {
        hipEvent_t a = nullptr, b = nullptr;
        ~event_pair()
        {
            if (a)
                hipEventDestroy(a);
            if (b)
                hipEventDestroy(b);
        }
    } ev;
    SPM_HIP_CHECK(ctx, hipEventCreate(&ev.a));
    SPM_HIP_CHECK(ctx, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    SPM_HIP_CHECK(ctx, hipEventRecord(e0, st));

    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_alo, (n_blocks + 2) * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_hap_start, (n_blocks + 1) * (uint64_t)H * 8));
    hipLaunchKernelGGL(jst_alo_kernel, dim3((unsigned)((n_blocks + 1 + 255) / 256)), dim3(256), 0, st, J->d_pos,
                       (uint64_t)J->al.size(), L, n_blocks, J->d_alo);
    jst_dev D = J->dev();
    int64_t *delta = reinterpret_cast<int64_t *>(J->d_hap_start);
    hipLaunchKernelGGL(jst_delta_kernel, dim3((unsigned)n_blocks), dim3(std::min<uint32_t>(256, (H + 63) & ~63u)), 0, st,
                       D, delta);
    {
        const uint32_t chunk = 256;
        const uint64_t n_chunks = (n_blocks + chunk - 1) / chunk;
        int64_t *csum = nullptr;
        SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&csum, n_chunks * H * 8));
        const unsigned g = (unsigned)((n_chunks * H + 255) / 256);
        hipLaunchKernelGGL(jst_chunk_sum_kernel, dim3(g), dim3(256), 0, st, delta, n_blocks, H, chunk, csum);
        hipLaunchKernelGGL(jst_chunk_scan_kernel, dim3((H + 63) / 64), dim3(64), 0, st, csum, n_chunks, H);
        hipLaunchKernelGGL(jst_chunk_apply_kernel, dim3(g), dim3(256), 0, st, delta, n_blocks, H, chunk, csum);
        hipLaunchKernelGGL(jst_start_kernel, dim3((unsigned)(((n_blocks + 1) * H + 255) / 256)), dim3(256), 0, st, D);
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    SPM_HIP_CHECK(ctx, hipGetLastError());

    // group the haplotypes of every (block, haplotype group) cell by context signature
    const uint32_t Hg = std::min<uint32_t>(H, kJstGroup); // haplotypes a workgroup compares
    const uint64_t n_groups = D.n_groups;
    const uint64_t nc = nb * n_groups;                    // cells
    if (nc >= (1ull << 31)) {
        SPM_SET_ERR(ctx, "spm_hip_jst_index: too many (block, haplotype group) cells");
        return SPM_E_UNSUPPORTED;
    }
    uint32_t n_slots = 64;
    while (n_slots < 2 * Hg)
        n_slots <<= 1;
    const size_t lds_dedupe = ((size_t)Hg * kJstSigWords + 2 * n_slots + Hg) * 4;
    uint32_t *d_nuniq = nullptr;
    unsigned long long *d_bytes = nullptr, *d_totals = nullptr;
    const uint64_t ncx = std::max<uint64_t>(nc, 1);
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_local_id, std::max<uint64_t>(nb, 1) * H * 2));
    SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&d_nuniq, ncx * 4));
    SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&d_bytes, ncx * 8));
    SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&d_totals, 16));
    SPM_HIP_CHECK(ctx, hipMemsetAsync(d_totals, 0, 16, st));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_ctx_base, (nc + 1) * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_byte_base, (nc + 1) * 8));
    unsigned long long totals[2] = {0, 0};
    uint64_t n_ctx = 0, ctx_bytes = 0;
    if (nc) {
        jst_index_out O{J->d_local_id, d_nuniq, d_bytes, d_totals};
        hipFuncSetAttribute((const void *)jst_dedupe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dedupe);
        hipLaunchKernelGGL(jst_dedupe_kernel, dim3((unsigned)nc), dim3(256), lds_dedupe, st, D, O);
        SPM_HIP_CHECK(ctx, hipGetLastError());
        // exclusive scans: contexts and bytes per cell
        uint64_t *d_n64 = nullptr;
        SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&d_n64, (nc + 1) * 8));
        SPM_HIP_CHECK(ctx, hipMemsetAsync(d_n64, 0, (nc + 1) * 8, st));
        hipLaunchKernelGGL(jst_widen_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, st, d_nuniq, nc, d_n64);
        void *tmp = nullptr;
        size_t tmp_bytes = 0;
        hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, d_n64, J->d_ctx_base, (int)(nc + 1), st);
        SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&tmp, std::max<size_t>(tmp_bytes, 16)));
        hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, d_n64, J->d_ctx_base, (int)(nc + 1), st);
        uint64_t *d_b64 = nullptr;
        SPM_HIP_CHECK(ctx, tmp_bufs.alloc(&d_b64, (nc + 1) * 8));
        SPM_HIP_CHECK(ctx, hipMemsetAsync(d_b64, 0, (nc + 1) * 8, st));
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(d_b64, d_bytes, nc * 8, hipMemcpyDeviceToDevice, st));
        hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, d_b64, J->d_byte_base, (int)(nc + 1), st);
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(&n_ctx, J->d_ctx_base + nc, 8, hipMemcpyDeviceToHost, st));
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(&ctx_bytes, J->d_byte_base + nc, 8, hipMemcpyDeviceToHost, st));
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(totals, d_totals, 16, hipMemcpyDeviceToHost, st));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    }
    J->n_ctx = n_ctx;
    J->ctx_bytes = ctx_bytes;
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_ctx_off, (n_ctx + 1) * 8));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_ctx_block, std::max<uint64_t>(n_ctx, 1) * 4));
    SPM_HIP_CHECK(ctx, hipMalloc(&J->d_ctx_owned, std::max<uint64_t>(n_ctx, 1) * 4));
    int rc = text_alloc(ctx, ctx_bytes, J->ref->sigma, &J->ctx_text);
    if (rc != SPM_OK)
        return rc;
    SPM_HIP_CHECK(ctx, hipMemcpyAsync(J->d_ctx_off + n_ctx, &ctx_bytes, 8, hipMemcpyHostToDevice, st));
    if (n_ctx) {
        jst_emit_out E{J->d_local_id, J->d_ctx_base, J->d_byte_base, J->d_ctx_off, J->d_ctx_block, J->d_ctx_owned,
                       J->ctx_text->d};
        const size_t lds_emit = (size_t)kJstGroup * 3 * 4;
        hipLaunchKernelGGL(jst_emit_kernel, dim3((unsigned)nc), dim3(256), lds_emit, st, D, E);
        SPM_HIP_CHECK(ctx, hipGetLastError());
    }
    SPM_HIP_CHECK(ctx, hipEventRecord(e1, st));
    SPM_HIP_CHECK(ctx, hipEventSynchronize(e1));
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    J->stats = spm_jst_stats{};
    J->stats.haplotype_symbols = totals[1];
    J->stats.context_symbols = ctx_bytes;
    J->stats.contexts = totals[0];
    J->stats.unique_contexts = n_ctx;
    J->stats.n_blocks = nb;
    J->stats.block_len = (uint32_t)L;
    J->stats.window = window;
    J->stats.ms_index = ms;
    J->indexed = true;
    if (spm_trace_on())
        fprintf(stderr, "[spm_hip] jst_index: window %u, %llu blocks of %u: %llu contexts, %llu unique, %llu symbols laid out "
                        "for %llu haplotype symbols; %.3f ms\n", window, (unsigned long long)nb, (unsigned)L,
                (unsigned long long)totals[0], (unsigned long long)n_ctx, (unsigned long long)ctx_bytes,
                (unsigned long long)totals[1], ms);
    return SPM_OK;
}

extern "C" int spm_hip_jst_stats(const spm_jst *J, spm_jst_stats *out)
{
    if (!J || !out)
        return SPM_E_INVALID;
    *out = J->stats;
    return SPM_OK;
}

extern "C" void spm_hip_jst_hits_destroy(spm_jst_hits *h)
{
    if (!h)
        return;
    if (h->d) {
        if (h->ctx && h->ctx->jst_pool.size() < 4)
            h->ctx->jst_pool.push_back({h->d, h->cap});
        else
            hipFree(h->d);
    }
    delete h;
}

extern "C" int spm_hip_jst_search(spm_jst *J, const spm_patterns *patterns, const spm_scan_opts *opts_in,
                                  spm_jst_hits **out)
{
    using namespace spm_hip;
    if (!J || !patterns || !out) {
        SPM_SET_ERR(J ? J->ctx : nullptr, "spm_hip_jst_search: invalid argument");
        return SPM_E_INVALID;
    }
    spm_ctx *ctx = J->ctx;
    if (!J->indexed) {
        SPM_SET_ERR(ctx, "spm_hip_jst_search: call spm_hip_jst_index first");
        return SPM_E_INVALID;
    }
    uint64_t need = 0;
    for (uint32_t p = 0; p < patterns->n; ++p)
        need = std::max<uint64_t>(need, spm_hip_patterns_window_size(patterns, p));
    if (need > J->window) {
        SPM_SET_ERR(ctx, "spm_hip_jst_search: needle window %llu exceeds the indexed window %u",
                    (unsigned long long)need, J->window);
        return SPM_E_INVALID;
    }
    if (patterns->algo == SPM_ALGO_MYERS_PREFIX) {
        SPM_SET_ERR(ctx, "spm_hip_jst_search: the prefix matcher has no meaning over haplotype contexts");
        return SPM_E_UNSUPPORTED;
    }
    SPM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<spm_jst_hits, void (*)(spm_jst_hits *)> R(new spm_jst_hits, spm_hip_jst_hits_destroy);
    R->ctx = ctx;
    spm_scan_opts o{};
    if (opts_in)
        o = *opts_in;
    o.left_context = 0;
    o.pos_offset = 0;
    const uint64_t out_cap = o.max_hits ? o.max_hits : (1ull << 22);
    if (o.max_hits == 0)
        o.max_hits = 1ull << 22;
    if (J->n_ctx == 0 || patterns->n == 0) {
        *out = R.release();
        return SPM_OK;
    }
    spm_hits *seg = nullptr;
    for (size_t i = 0; i < ctx->jst_pool.size(); ++i)
        if (ctx->jst_pool[i].second == out_cap) {
            R->d = static_cast<spm_jst_hit *>(ctx->jst_pool[i].first);
            ctx->jst_pool.erase(ctx->jst_pool.begin() + (long)i);
            break;
        }
    if (!R->d)
        SPM_HIP_CHECK(ctx, hipMalloc(&R->d, out_cap * sizeof(spm_jst_hit)));
    R->cap = out_cap;
    if (!J->d_fan_count) {
        SPM_HIP_CHECK(ctx, hipMalloc(&J->d_fan_count, 8));
        SPM_HIP_CHECK(ctx, hipEventCreate(&J->fan_ev[0]));
        SPM_HIP_CHECK(ctx, hipEventCreate(&J->fan_ev[1]));
    }
    hipEvent_t e0 = J->fan_ev[0], e1 = J->fan_ev[1];
    jst_fan_params F{};
    F.ctx_off = J->d_ctx_off;
    F.n_ctx = J->n_ctx;
    F.ctx_block = J->d_ctx_block;
    F.ctx_owned = J->d_ctx_owned;
    F.ctx_base = J->d_ctx_base;
    F.local_id = J->d_local_id;
    F.m = patterns->d_m;
    F.report_begin = patterns->is_myers() ? 0 : 1;
    F.out = R->d;
    F.out_cap = out_cap;
    // The fan-out is launched right behind the scan's kernels, before the host has read the scan's counters: it takes the
    // number of segment hits from the device and counts its records into a spare slot of the scan's counter block, so one
    // read-back serves both (a second host round trip per search cost ~35 us of an idle GPU).  If the scan has to go on
    // after that read-back (overflow, fallback), the fan-out is simply run again below.
    const std::function<int(spm_hits *)> fan_hook = [&](spm_hits *h) -> int {
        jst_fan_params G = F;
        G.hits = h->d_hits;
        G.n_hits = 0;
        G.n_hits_dev = h->d_count;
        G.hit_cap = h->cap;
        G.out_count = h->d_count + 12;
        const uint64_t expect = std::max<uint64_t>(1u << 16, 2 * J->seg_hit_hint);
        SPM_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(jst_fanout_kernel, dim3((unsigned)std::min<uint64_t>((expect + 255) / 256, (uint64_t)ctx->n_cu * 64)),
                           dim3(256), 0, ctx->stream, J->dev(), G);
        SPM_HIP_CHECK(ctx, hipGetLastError());
        SPM_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
        return SPM_OK;
    };
    // the verification stage skips end positions inside a context's left context (the fan-out would drop them)
    int rc = scan_impl(ctx, J->ctx_text, 0, J->ctx_bytes, patterns, &o, nullptr, nullptr, nullptr, J->n_ctx, &seg,
                       J->d_ctx_off, J->d_ctx_owned, &fan_hook);
    if (rc != SPM_OK)
        return rc;
    std::unique_ptr<spm_hits, void (*)(spm_hits *)> S(seg, spm_hip_hits_destroy);
    const void *d_rec = nullptr;
    uint64_t n_seg_hits = 0;
    rc = spm_hip_hits_device(seg, &d_rec, &n_seg_hits);
    if (rc != SPM_OK)
        return rc;
    spm_scan_stats ss{};
    spm_hip_hits_stats(seg, &ss);
    J->stats.ms_scan = ss.ms_total;
    J->stats.ms_main = ss.ms_main;
    J->stats.ms_verify = ss.ms_verify;
    J->stats.engine_used = ss.engine_used;
    J->stats.main_launches = ss.main_launches;
    J->stats.segment_hits = n_seg_hits;
    J->stats.fell_back = ss.fell_back;
    J->stats.candidates = ss.n_candidates;
    J->stats.bands = ss.n_bands;
    J->seg_hit_hint = std::max<uint64_t>(J->seg_hit_hint, n_seg_hits);
    unsigned long long n_out = 0;
    if (seg->hook_final) {
        n_out = seg->fan_count; // counted by the fan-out that ran behind the scan, read back with the scan's counters
    } else {
        unsigned long long *d_count = J->d_fan_count;
        SPM_HIP_CHECK(ctx, hipMemsetAsync(d_count, 0, 8, ctx->stream));
        SPM_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
        if (n_seg_hits) {
            jst_fan_params G = F;
            G.hits = static_cast<const spm_hit *>(d_rec);
            G.n_hits = n_seg_hits;
            G.n_hits_dev = nullptr;
            G.out_count = d_count;
            hipLaunchKernelGGL(jst_fanout_kernel, dim3((unsigned)((n_seg_hits + 255) / 256)), dim3(256), 0, ctx->stream,
                               J->dev(), G);
            SPM_HIP_CHECK(ctx, hipGetLastError());
        }
        SPM_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
        SPM_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_counters + 8, d_count, 8, hipMemcpyDeviceToHost, ctx->stream));
        SPM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        n_out = ctx->h_counters[8];
    }
    hipEventElapsedTime(&J->stats.ms_fanout, e0, e1);
    if (n_out > out_cap) {
        SPM_SET_ERR(ctx, "journaled-sequence search produced %llu hits but the buffer holds %llu; raise "
                         "spm_scan_opts.max_hits", n_out, (unsigned long long)out_cap);
        return SPM_E_OVERFLOW;
    }
    R->n = n_out;
    if (spm_trace_on())
        fprintf(stderr, "[spm_hip] jst_search: %llu segment hits -> %llu records; scan %.3f ms (main %.3f, verification %.3f), "
                        "fan-out %.3f ms%s\n", (unsigned long long)n_seg_hits, n_out, J->stats.ms_scan, J->stats.ms_main,
                J->stats.ms_verify, J->stats.ms_fanout, J->stats.fell_back ? "; the seed filter fell back" : "");
    *out = R.release();
    return SPM_OK;
}

extern "C" int spm_hip_jst_hits_view(spm_jst_hits *h, const spm_jst_hit **records, uint64_t *n)
{
    if (!h || !records || !n)
        return SPM_E_INVALID;
    if (!h->sorted) {
        h->host.resize(h->n);
        if (h->n) {
            SPM_HIP_CHECK(h->ctx, hipMemcpyAsync(h->host.data(), h->d, h->n * sizeof(spm_jst_hit),
                                                 hipMemcpyDeviceToHost, h->ctx->stream));
            SPM_HIP_CHECK(h->ctx, hipStreamSynchronize(h->ctx->stream));
        }
        std::sort(h->host.begin(), h->host.end(), [](const spm_jst_hit &a, const spm_jst_hit &b) {
            if (a.haplotype != b.haplotype)
                return a.haplotype < b.haplotype;
            if (a.pos != b.pos)
                return a.pos < b.pos;
            if (a.pattern != b.pattern)
                return a.pattern < b.pattern;
            return a.score < b.score;
        });
        h->sorted = true;
    }
    *records = h->host.data();
    *n = h->n;
    return SPM_OK;
}

extern "C" int spm_hip_jst_hits_device(spm_jst_hits *h, const void **device_records, uint64_t *n)
{
    if (!h || !device_records || !n)
        return SPM_E_INVALID;
    *device_records = h->d;
    *n = h->n;
    return SPM_OK;
}

extern "C" int spm_hip_jst_hits_copy_device(spm_jst_hits *h, void *device_dst, uint64_t cap, uint64_t *n)
{
    if (!h || !n || (cap && !device_dst))
        return SPM_E_INVALID;
    *n = h->n;
    const uint64_t c = std::min(h->n, cap);
    if (c)
        SPM_HIP_CHECK(h->ctx, hipMemcpyAsync(device_dst, h->d, c * sizeof(spm_jst_hit), hipMemcpyDeviceToDevice,
                                             h->ctx->stream));
    return SPM_OK;
}

// Synthetic variants of config C5.  Per 1000-base block b of the reference: one SNP at an offset in
// [0, 900); per 10 000 bases one indel of length 1..50 placed in [900, 950) of one of its 1000-blocks, so no two
// alleles overlap.  Everything derives from mix64(seed_var, global block index), i.e. shards agree on the variants.
extern "C" int spm_hip_jst_synth_variants(uint64_t seed_text, uint64_t seed_var, uint64_t ref_begin, uint64_t n_ref,
                                          uint32_t n_hap, spm_jst_allele *alleles, uint64_t *n_alleles,
                                          uint8_t *alt_pool, uint64_t *alt_pool_len, uint64_t *coverage)
{
    using spm_hip::mix64;
    if (!n_alleles || !alt_pool_len || n_hap == 0 || n_hap > 64 || (ref_begin % 10000) != 0)
        return SPM_E_INVALID;
    const uint64_t hap_mask = n_hap == 64 ? ~0ull : ((1ull << n_hap) - 1);
    uint64_t na = 0, np = 0;
    auto cov_of = [&](uint64_t r) {
        uint64_t c = mix64(r ^ 0xC0FEull) & hap_mask;
        if (c == 0)
            c = 1ull << (r % n_hap);
        return c;
    };
    const uint64_t g0 = ref_begin / 1000;
    for (uint64_t b = 0; b * 1000 < n_ref; ++b) {
        const uint64_t gb = g0 + b;
        {
            const uint64_t r = mix64(seed_var + 2 * gb);
            const uint64_t p = b * 1000 + r % 900;
            if (p < n_ref) {
                if (alleles) {
                    const uint8_t rb = spm_hip::synth_base(seed_text, ref_begin + p);
                    alleles[na] = spm_jst_allele{p, 1, 1, np};
                    alt_pool[np] = (uint8_t)((rb + 1 + (r >> 20) % 3) & 3);
                    coverage[na] = cov_of(r);
                }
                ++na;
                ++np;
            }
        }
        const uint64_t r10 = mix64(seed_var + 2 * (gb / 10) + 1);
        if (gb % 10 == r10 % 10) {
            const uint64_t p = b * 1000 + 900 + (r10 >> 8) % 50;
            const uint32_t len = 1 + (uint32_t)((r10 >> 16) % 50);
            const bool del = (r10 >> 24) & 1;
            if (p + len <= n_ref && p < n_ref) {
                if (alleles) {
                    if (del) {
                        alleles[na] = spm_jst_allele{p, len, 0, np};
                    } else {
                        alleles[na] = spm_jst_allele{p, 0, len, np};
                        for (uint32_t x = 0; x < len; ++x)
                            alt_pool[np + x] = (uint8_t)((mix64(r10 + 0x1234567ull * (x / 32 + 1)) >> (2 * (x & 31))) & 3);
                    }
                    coverage[na] = cov_of(r10 ^ 0x5555ull);
                }
                ++na;
                if (!del)
                    np += len;
            }
        }
    }
    *n_alleles = na;
    *alt_pool_len = np;
    return SPM_OK;
}

void spm_warm_jst_kernels()
{
    hipFuncAttributes a;
    (void)hipFuncGetAttributes(&a, (const void *)spm_hip::jst_fanout_kernel);
}
