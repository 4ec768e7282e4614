/*
 * spm_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see spm_oracle.h for scope, citations, pinning).
 *
 * Plain C11, no dependencies.  Built by oracle/Makefile into oracle/libspm_oracle.so.
 */
#include "spm_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Alphabets.  seqan/alphabet.hpp:68-72 makes the implicit integer conversion of a symbol its seqan3
 * rank; that rank indexes every Peq / mask / skip table.  Rank orders are seqan3's:
 *   dna4  A C G T            (unknown char -> A)
 *   dna5  A C G N T          (unknown char -> N)     [confirmed by .jst fixture bytes, SURVEY 8f-2]
 *   dna15 A B C D G H K M N R S T V W Y (unknown -> N)
 * ---------------------------------------------------------------------------------------------- */
static const char DNA4[] = "ACGT";
static const char DNA5[] = "ACGNT";
static const char DNA15[] = "ABCDGHKMNRSTVWY";

static const char *alpha_of(uint32_t sigma) { return sigma == 4 ? DNA4 : sigma == 5 ? DNA5 : DNA15; }

uint8_t spm_oracle_char_to_rank(uint32_t sigma, char c)
{
    const char *a = alpha_of(sigma);
    if (c >= 'a' && c <= 'z')
        c = (char)(c - 'a' + 'A');
    if (c == 'U')
        c = 'T';
    for (uint32_t r = 0; a[r]; ++r)
        if (a[r] == c)
            return (uint8_t)r;
    if (sigma == 4) { /* seqan3::dna4 folds IUPAC codes onto a member base; plain unknowns -> A */
        switch (c) {
        case 'B': case 'S': case 'Y': case 'M': case 'H': case 'V': return 1; /* C */
        case 'K': return 2;                                                     /* G */
        case 'D': case 'N': case 'R': case 'W': default: return 0;              /* A */
        }
    }
    return spm_oracle_char_to_rank(sigma, 'N');
}

char spm_oracle_rank_to_char(uint32_t sigma, uint8_t rank)
{
    const char *a = alpha_of(sigma);
    return rank < sigma ? a[rank] : '?';
}

/* ------------------------------------------------------------------------------------------------
 * Exact matchers
 * ---------------------------------------------------------------------------------------------- */
size_t spm_oracle_naive_exact(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint64_t *out_pos,
                              size_t cap)
{
    size_t cnt = 0;
    if (m == 0 || m > n)
        return 0;
    for (size_t i = 0; i + m <= n; ++i)
        if (memcmp(text + i, pat, m) == 0) {
            if (cnt < cap)
                out_pos[cnt] = i;
            ++cnt;
        }
    return cnt;
}

/* Boyer-Moore-Horspool.  skip[c] = m-1-(last index of c in P[0..m-2]), default m.  After a window is
 * examined (hit or not) shift by skip[text[window_end]] -- this never skips an occurrence, so the hit set
 * is every occurrence, overlapping ones included.  Follows SURVEY 8a-a7 / horspool_matcher.hpp:28,38-40. */
size_t spm_oracle_horspool(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint32_t sigma,
                           uint64_t *out_pos, size_t cap)
{
    size_t cnt = 0;
    if (m == 0 || m > n)
        return 0;
    size_t skip[256];
    for (uint32_t c = 0; c < 256; ++c)
        skip[c] = m;
    for (size_t i = 0; i + 1 < m; ++i)
        skip[pat[i]] = m - 1 - i;
    (void)sigma;
    size_t end = m - 1; /* index of window's last symbol */
    while (end < n) {
        size_t i = 0;
        while (i < m && text[end - i] == pat[m - 1 - i])
            ++i;
        if (i == m) {
            if (cnt < cap)
                out_pos[cnt] = end + 1 - m;
            ++cnt;
        }
        end += skip[text[end]];
    }
    return cnt;
}

/* Shift-Or, 32-bit words (SeqAn's TWord = unsigned), any number of blocks.  mask[c][b] has bit j CLEAR
 * iff P[32b+j]==c.  Per symbol: R = (R<<1, carry across blocks) | mask[c]; occurrence ends here iff bit
 * (m-1) of R is 0.  Reported position = begin = end-m+1 (finder is moved back by m-1 on a hit). */
size_t spm_oracle_shiftor(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint32_t sigma,
                          uint32_t *state, uint64_t text_offset, uint64_t *out_pos, size_t cap)
{
    size_t cnt = 0;
    if (m == 0)
        return 0;
    size_t nb = (m + 31) / 32;
    uint32_t *mask = (uint32_t *)malloc(sizeof(uint32_t) * nb * sigma);
    uint32_t *R = state ? state : (uint32_t *)malloc(sizeof(uint32_t) * nb);
    memset(mask, 0xFF, sizeof(uint32_t) * nb * sigma);
    for (size_t j = 0; j < m; ++j)
        if (pat[j] < sigma)
            mask[(size_t)pat[j] * nb + j / 32] &= ~(1u << (j % 32));
    if (!state)
        memset(R, 0xFF, sizeof(uint32_t) * nb);
    const uint32_t last_bit = 1u << ((m - 1) % 32);
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = text[i];
        uint32_t carry = 0;
        for (size_t b = 0; b < nb; ++b) {
            uint32_t top = R[b] >> 31;
            uint32_t mk = c < sigma ? mask[(size_t)c * nb + b] : 0xFFFFFFFFu;
            R[b] = ((R[b] << 1) | carry) | mk;
            carry = top;
        }
        if ((R[nb - 1] & last_bit) == 0) {
            uint64_t end = text_offset + i; /* inclusive end */
            if (end + 1 >= m) {             /* occurrence may start in an earlier chunk */
                if (cnt < cap)
                    out_pos[cnt] = end + 1 - m;
                ++cnt;
            }
        }
    }
    free(mask);
    if (!state)
        free(R);
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * Myers bit-vector
 * ---------------------------------------------------------------------------------------------- */
static uint32_t rows_in_block(size_t m, uint32_t b)
{
    size_t lo = (size_t)b * 64;
    size_t hi = lo + 64 < m ? lo + 64 : m;
    return (uint32_t)(hi - lo);
}

void spm_oracle_myers_init(spm_oracle_myers_state *st, size_t m, uint32_t k, int cutoff)
{
    memset(st, 0, sizeof(*st));
    uint32_t nb = (uint32_t)((m + 63) / 64);
    if (nb == 0)
        nb = 1;
    st->n_blocks = nb;
    int32_t acc = 0;
    for (uint32_t b = 0; b < nb; ++b) {
        st->vp[b] = ~0ull; /* D[i][0] = i : every vertical delta is +1 */
        st->vn[b] = 0;
        acc += (int32_t)rows_in_block(m, b);
        st->score[b] = acc;
    }
    if (cutoff) {
        /* rows 1..k have D <= k in the empty-prefix column */
        uint32_t a = (uint32_t)(k / 64) + 1;
        st->active = a < nb ? a : nb;
    } else {
        st->active = nb;
    }
}

static void build_peq(uint64_t *peq, const uint8_t *pat, size_t m, uint32_t sigma, uint32_t nb)
{
    memset(peq, 0, sizeof(uint64_t) * (size_t)nb * sigma);
    for (size_t j = 0; j < m; ++j)
        if (pat[j] < sigma)
            peq[(size_t)pat[j] * nb + j / 64] |= 1ull << (j % 64);
}

size_t spm_oracle_myers_scan(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint32_t sigma,
                             uint32_t k, int mode, int variant, spm_oracle_myers_state *st,
                             uint64_t text_offset, spm_oracle_hit *out, size_t cap)
{
    size_t cnt = 0;
    if (m == 0)
        return 0;
    const uint32_t nb = st->n_blocks;
    uint64_t *peq = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)nb * (sigma + 1));
    build_peq(peq, pat, m, sigma, nb);
    const uint64_t hp0 = (mode == SPM_ORACLE_PREFIX) ? 1u : 0u; /* [upstream] MyersUkkonenHP0_ */
    const int32_t kk = (int32_t)k;

    if (variant == 0) {
        /* _findMyersSmallPatterns: one 64-bit word. */
        uint64_t VP = st->vp[0], VN = st->vn[0];
        int32_t errors = st->score[0];
        const uint64_t last = 1ull << (m - 1);
        for (size_t i = 0; i < n; ++i) {
            uint32_t c = text[i];
            uint64_t X = (c < sigma ? peq[c] : 0) | VN;
            uint64_t D0 = ((VP + (X & VP)) ^ VP) | X;
            uint64_t HN = VP & D0;
            uint64_t HP = VN | ~(VP | D0);
            X = (HP << 1) | hp0;
            VN = X & D0;
            VP = (HN << 1) | ~(X | D0);
            if (HP & last)
                ++errors;
            else if (HN & last)
                --errors;
            if (errors <= kk) {
                if (cnt < cap) {
                    out[cnt].pos = text_offset + i + 1;
                    out[cnt].pattern = 0;
                    out[cnt].score = errors;
                }
                ++cnt;
            }
        }
        st->vp[0] = VP;
        st->vn[0] = VN;
        st->score[0] = errors;
        free(peq);
        return cnt;
    }

    const int cutoff = (variant == 2);
    for (size_t i = 0; i < n; ++i) {
        uint32_t c = text[i];
        if (cutoff && st->active < nb && st->score[st->active - 1] <= kk) {
            /* Band grows: the first row of the next block can be <= k in this column only if the bottom
             * row of the last active block was <= k in the previous one (D[i][j] >= D[i-1][j-1]).  The
             * entering block is given vertical deltas +1 -- an over-estimate of cells that are all > k,
             * which leaves every cell whose true value is <= k exact (Myers 1999, sec. 4). */
            uint32_t b = st->active;
            st->vp[b] = ~0ull;
            st->vn[b] = 0;
            st->score[b] = st->score[b - 1] + (int32_t)rows_in_block(m, b);
            st->active = b + 1;
        }
        uint64_t carryD0 = 0, carryHP = hp0, carryHN = 0;
        const uint32_t act = st->active;
        for (uint32_t b = 0; b < act; ++b) {
            uint64_t VP = st->vp[b], VN = st->vn[b];
            uint64_t X = (c < sigma ? peq[(size_t)c * nb + b] : 0) | VN;
            uint64_t t = X & VP;
            uint64_t s1 = VP + t;
            uint64_t c1 = s1 < VP;
            uint64_t s2 = s1 + carryD0;
            uint64_t c2 = s2 < s1;
            carryD0 = c1 | c2;
            uint64_t D0 = (s2 ^ VP) | X;
            uint64_t HN = VP & D0;
            uint64_t HP = VN | ~(VP | D0);
            uint32_t top = rows_in_block(m, b) - 1; /* bit of this block's bottom row */
            st->score[b] += (int32_t)((HP >> top) & 1) - (int32_t)((HN >> top) & 1);
            X = (HP << 1) | carryHP;
            carryHP = HP >> 63;
            st->vn[b] = X & D0;
            uint64_t t2 = (HN << 1) | carryHN;
            carryHN = HN >> 63;
            st->vp[b] = t2 | ~(X | D0);
        }
        if (cutoff) {
            /* Band shrinks while the whole last block is provably > k (vertical deltas are <= 1). */
            while (st->active > 1 &&
                   st->score[st->active - 1] >= kk + (int32_t)rows_in_block(m, st->active - 1))
                --st->active;
        }
        if (st->active == nb && st->score[nb - 1] <= kk) {
            if (cnt < cap) {
                out[cnt].pos = text_offset + i + 1;
                out[cnt].pattern = 0;
                out[cnt].score = st->score[nb - 1];
            }
            ++cnt;
        }
    }
    free(peq);
    return cnt;
}

/* The same scan as variant 2 (blocks with Ukkonen cut-off, infix mode, fresh matcher) for 64 < m <= 128, everything in
 * registers: the loop the CPU baseline times.  [upstream] _findMyersLargePatterns keeps its state in a heap object and
 * loops over lastBlock+1 blocks; for two blocks that is this code.  On text without near-occurrences only block 0 is
 * inside the band, so a symbol costs one 64-bit word step (~1 ns), which is what SURVEY 8(a)-a5 quotes for SeqAn.
 * tests/test_oracle_golden.py checks it hit for hit against variant 2 and against Sellers. */
size_t spm_oracle_myers2_fast(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint32_t sigma, uint32_t k,
                              uint64_t text_offset, spm_oracle_hit *out, size_t cap)
{
    if (m <= 64 || m > 128)
        return (size_t)-1;
    uint64_t peq0[256], peq1[256];
    memset(peq0, 0, sizeof(peq0));
    memset(peq1, 0, sizeof(peq1));
    for (size_t j = 0; j < m; ++j)
        if (pat[j] < sigma) {
            if (j < 64)
                peq0[pat[j]] |= 1ull << j;
            else
                peq1[pat[j]] |= 1ull << (j - 64);
        }
    const int32_t kk = (int32_t)k;
    const int32_t rows1 = (int32_t)(m - 64);
    const uint32_t top1 = (uint32_t)(m - 65); /* bit of the needle's last row inside block 1 */
    uint64_t VP0 = ~0ull, VN0 = 0, VP1 = ~0ull, VN1 = 0;
    int32_t s0 = 64, s1 = (int32_t)m; /* D at the bottom row of block 0 / block 1 */
    int two = kk >= 64;               /* is block 1 inside the band? */
    size_t cnt = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t c = text[i] < sigma ? text[i] : 255; /* row 255 of the tables stays empty for sigma <= 255 */
        if (!two && s0 <= kk) { /* band grows (see variant 2) */
            VP1 = ~0ull;
            VN1 = 0;
            s1 = s0 + rows1;
            two = 1;
        }
        uint64_t X = peq0[c] | VN0;
        const uint64_t sum0 = VP0 + (X & VP0);
        const uint64_t carry = sum0 < VP0;
        uint64_t D0 = (sum0 ^ VP0) | X;
        uint64_t HN = VP0 & D0;
        uint64_t HP = VN0 | ~(VP0 | D0);
        s0 += (int32_t)(HP >> 63) - (int32_t)(HN >> 63);
        const uint64_t cHP = HP >> 63, cHN = HN >> 63;
        X = HP << 1;
        VN0 = X & D0;
        VP0 = (HN << 1) | ~(X | D0);
        if (two) {
            X = peq1[c] | VN1;
            const uint64_t t = X & VP1;
            const uint64_t sum1 = VP1 + t + carry;
            D0 = (sum1 ^ VP1) | X;
            HN = VP1 & D0;
            HP = VN1 | ~(VP1 | D0);
            s1 += (int32_t)((HP >> top1) & 1) - (int32_t)((HN >> top1) & 1);
            X = (HP << 1) | cHP;
            VN1 = X & D0;
            VP1 = ((HN << 1) | cHN) | ~(X | D0);
            if (s1 <= kk) {
                if (cnt < cap) {
                    out[cnt].pos = text_offset + i + 1;
                    out[cnt].pattern = 0;
                    out[cnt].score = s1;
                }
                ++cnt;
            }
            if (s1 >= kk + rows1) /* band shrinks: the whole of block 1 is > k */
                two = 0;
        }
    }
    return cnt;
}

size_t spm_oracle_sellers(const uint8_t *text, size_t n, const uint8_t *pat, size_t m, uint32_t k, int mode,
                          int32_t *col, uint64_t text_offset, spm_oracle_hit *out, size_t cap)
{
    size_t cnt = 0;
    if (m == 0)
        return 0;
    int32_t *D = col ? col : (int32_t *)malloc(sizeof(int32_t) * (m + 1));
    if (!col)
        for (size_t i = 0; i <= m; ++i)
            D[i] = (int32_t)i;
    for (size_t j = 0; j < n; ++j) {
        int32_t diag = D[0];
        if (mode == SPM_ORACLE_PREFIX)
            D[0] += 1;
        for (size_t i = 1; i <= m; ++i) {
            int32_t v = diag + (pat[i - 1] == text[j] ? 0 : 1);
            if (D[i] + 1 < v)
                v = D[i] + 1;
            if (D[i - 1] + 1 < v)
                v = D[i - 1] + 1;
            diag = D[i];
            D[i] = v;
        }
        if (D[m] <= (int32_t)k) {
            if (cnt < cap) {
                out[cnt].pos = text_offset + j + 1;
                out[cnt].pattern = 0;
                out[cnt].score = D[m];
            }
            ++cnt;
        }
    }
    if (!col)
        free(D);
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * Synthetic inputs; the product has its own copy in libspm_amd/csrc/synth.hpp --
 * tests cross-check the two.
 * ---------------------------------------------------------------------------------------------- */
uint64_t spm_oracle_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint8_t text_base(uint64_t seed, uint64_t i)
{
    return (uint8_t)((spm_oracle_mix64(seed + (i >> 5)) >> (2 * (i & 31))) & 3);
}

void spm_oracle_text(uint64_t seed, uint64_t begin, uint64_t n, uint8_t *out)
{
    for (uint64_t i = 0; i < n; ++i)
        out[i] = text_base(seed, begin + i);
}

/* Repeat-rich text of bench workload c3r (product copy: libspm_amd/csrc/synth.hpp, repeat_base).  1024-base blocks;
 * block b holds one stretch with probability ppm * 1024 / 136 / 1e6, decided by the low 32 bits of the block hash:
 * length 16 * (1 + bits 32..35), offset (bits 36..51) mod (1024 - length + 1); bit 52 = kind.
 *   kind 0: tandem repeat, unit length 1 + (hash >> 53) mod 6, unit symbols = bit pairs of a second hash, one base in 64
 *           replaced by a different one;   kind 1: dominant base (hash >> 53) & 3 with probability 7/8, else uniform. */
void spm_oracle_repeat_text(uint64_t seed, uint32_t ppm, uint64_t begin, uint64_t n, uint8_t *out)
{
    const uint64_t thr = ((uint64_t)ppm * 1024ull * 4294967296ull) / (136ull * 1000000ull);
    for (uint64_t t = 0; t < n; ++t) {
        const uint64_t i = begin + t;
        const uint64_t h = spm_oracle_mix64((seed ^ 0x7E9EA7ull) + (i >> 10) * 0x9E3779B97F4A7C15ull);
        const uint32_t len = 16u * (1u + (uint32_t)((h >> 32) & 15));
        const uint32_t off = (uint32_t)((h >> 36) & 0xFFFF) % (1024u - len + 1u);
        const uint32_t in = (uint32_t)(i & 1023);
        if ((h & 0xFFFFFFFFull) >= thr || in < off || in >= off + len) {
            out[t] = text_base(seed, i);
            continue;
        }
        const uint32_t j = in - off;
        const uint64_t h2 = spm_oracle_mix64(h);
        const uint32_t x = (uint32_t)(spm_oracle_mix64(h2 + 1 + (j >> 3)) >> (8 * (j & 7))) & 0xFF;
        if (((h >> 52) & 1) == 0) {
            const uint32_t u = 1u + (uint32_t)((h >> 53) % 6);
            uint32_t sym = (uint32_t)(h2 >> (2 * (j % u))) & 3u;
            if ((x & 63u) == 0)
                sym = (sym + 1u + (x >> 6) % 3u) & 3u;
            out[t] = (uint8_t)sym;
        } else {
            out[t] = (uint8_t)((x & 7u) != 0 ? ((h >> 53) & 3u) : ((x >> 3) & 3u));
        }
    }
}

static inline uint64_t pat_rnd(uint64_t seed_pat, uint32_t p, uint32_t t)
{
    return spm_oracle_mix64(seed_pat + ((uint64_t)p << 16) + t);
}

uint64_t spm_oracle_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                            uint32_t kmax, uint8_t *out)
{
    const uint64_t o = pat_rnd(seed_pat, p, 0) % (n_total - 2ull * L);
    const uint32_t e = p % (kmax + 1);
    /* edit positions: distinct, in source coordinates [0,L) */
    uint32_t epos[64], etype[64], ebase[64];
    const uint32_t ne = e < 64 ? e : 64;
    for (uint32_t i = 0; i < ne; ++i) {
        uint32_t pos = (uint32_t)(pat_rnd(seed_pat, p, 1 + 2 * i) % L);
        for (;;) {
            int clash = 0;
            for (uint32_t j = 0; j < i; ++j)
                if (epos[j] == pos)
                    clash = 1;
            if (!clash)
                break;
            pos = (pos + 1) % L;
        }
        uint64_t r = pat_rnd(seed_pat, p, 2 + 2 * i);
        epos[i] = pos;
        etype[i] = (uint32_t)(r % 3);
        ebase[i] = (uint32_t)((r >> 8) & 3);
    }
    uint32_t produced = 0;
    for (uint64_t x = 0; produced < L; ++x) {
        uint8_t b = text_base(seed_text, o + x);
        int hit = -1;
        if (x < L)
            for (uint32_t i = 0; i < ne; ++i)
                if (epos[i] == x)
                    hit = (int)i;
        if (hit < 0) {
            out[produced++] = b;
        } else if (etype[hit] == 0) { /* substitute with a different base */
            out[produced++] = (uint8_t)((b + 1 + ebase[hit] % 3) & 3);
        } else if (etype[hit] == 1) { /* delete */
        } else {                      /* insert a base in front */
            out[produced++] = (uint8_t)ebase[hit];
            if (produced < L)
                out[produced++] = b;
        }
    }
    return o;
}

uint64_t spm_oracle_checksum(const spm_oracle_hit *hits, size_t n)
{
    uint64_t s = 0;
    for (size_t i = 0; i < n; ++i)
        s += spm_oracle_mix64(hits[i].pos ^ ((uint64_t)hits[i].pattern << 40) ^
                              ((uint64_t)(uint32_t)hits[i].score << 58));
    return s;
}

/* ------------------------------------------------------------------------------------------------
 * Multi-pattern driver (CPU baseline): the reference's usage model is one matcher object per needle and
 * one full sequential pass per matcher (seqan_pattern_base.hpp:40-52); patterns are dealt to threads.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int algo;
    const uint8_t *text;
    size_t n;
    const uint8_t *pats;
    const uint32_t *offsets;
    uint32_t p_begin, p_end;
    uint32_t sigma, k;
    spm_oracle_hit *out;
    size_t cap, cnt;
    int overflow;
} multi_job;

static void *multi_worker(void *arg)
{
    multi_job *J = (multi_job *)arg;
    size_t tmp_cap = 1u << 16;
    spm_oracle_hit *tmp = (spm_oracle_hit *)malloc(sizeof(spm_oracle_hit) * tmp_cap);
    uint64_t *tpos = (uint64_t *)malloc(sizeof(uint64_t) * tmp_cap);
    for (uint32_t p = J->p_begin; p < J->p_end; ++p) {
        const uint8_t *pat = J->pats + J->offsets[p];
        size_t m = J->offsets[p + 1] - J->offsets[p];
        size_t c = 0;
        if (J->algo == 1 && m > 64 && m <= 128) {
            c = spm_oracle_myers2_fast(J->text, J->n, pat, m, J->sigma, J->k, 0, tmp, tmp_cap);
        } else if (J->algo == 1) {
            spm_oracle_myers_state st;
            spm_oracle_myers_init(&st, m, J->k, m > 64);
            c = spm_oracle_myers_scan(J->text, J->n, pat, m, J->sigma, J->k, SPM_ORACLE_INFIX, m > 64 ? 2 : 0,
                                      &st, 0, tmp, tmp_cap);
        } else {
            c = J->algo == 0 ? spm_oracle_shiftor(J->text, J->n, pat, m, J->sigma, NULL, 0, tpos, tmp_cap)
                             : spm_oracle_horspool(J->text, J->n, pat, m, J->sigma, tpos, tmp_cap);
            for (size_t i = 0; i < c && i < tmp_cap; ++i) {
                tmp[i].pos = tpos[i];
                tmp[i].score = 0;
            }
        }
        if (c > tmp_cap) {
            J->overflow = 1;
            c = tmp_cap;
        }
        for (size_t i = 0; i < c; ++i) {
            if (J->cnt < J->cap) {
                J->out[J->cnt] = tmp[i];
                J->out[J->cnt].pattern = p;
                ++J->cnt;
            } else {
                J->overflow = 1;
            }
        }
    }
    free(tmp);
    free(tpos);
    return NULL;
}

size_t spm_oracle_scan_multi(int algo, const uint8_t *text, size_t n, const uint8_t *pats,
                             const uint32_t *offsets, uint32_t n_patterns, uint32_t sigma, uint32_t k,
                             int n_threads, spm_oracle_hit *out, size_t cap)
{
    if (n_threads < 1)
        n_threads = 1;
    if ((uint32_t)n_threads > n_patterns)
        n_threads = (int)(n_patterns ? n_patterns : 1);
    multi_job *jobs = (multi_job *)calloc((size_t)n_threads, sizeof(multi_job));
    pthread_t *th = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
    size_t per_cap = cap; /* each worker gets a private buffer, merged in pattern order afterwards */
    for (int t = 0; t < n_threads; ++t) {
        multi_job *J = &jobs[t];
        J->algo = algo;
        J->text = text;
        J->n = n;
        J->pats = pats;
        J->offsets = offsets;
        J->p_begin = (uint32_t)((uint64_t)n_patterns * t / n_threads);
        J->p_end = (uint32_t)((uint64_t)n_patterns * (t + 1) / n_threads);
        J->sigma = sigma;
        J->k = k;
        J->out = (spm_oracle_hit *)malloc(sizeof(spm_oracle_hit) * (per_cap ? per_cap : 1));
        J->cap = per_cap;
        pthread_create(&th[t], NULL, multi_worker, J);
    }
    size_t total = 0;
    int overflow = 0;
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        multi_job *J = &jobs[t];
        overflow |= J->overflow;
        for (size_t i = 0; i < J->cnt; ++i) {
            if (total < cap)
                out[total++] = J->out[i];
            else
                overflow = 1;
        }
        free(J->out);
    }
    free(jobs);
    free(th);
    return overflow ? (size_t)-1 : total;
}
