/*
 * spm_hip.h -- C ABI of the MI355X-native online pattern-matching engine (libspm_hip.so).
 *
 * This is the drop-in boundary for libspm's matcher hot path.  The reference has no FFI for this path:
 * its boundary is the header-only C++ template API
 *     spm::{horspool,shiftor,myers,restorable_*}_matcher::operator()(haystack, callback)
 *         /root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:40-52
 * so the entry points below are what a binding for that call operator needs: build the pattern tables once
 * (the matcher constructors), hand over a haystack of 1-byte ranks, run one scan, read back the hits in the
 * order the callback would have seen them.  include/libspm/ holds the C++ mirror of the reference API that
 * marshals to these calls; INTEGRATION.md shows the binding from the reference side.
 *
 * Conventions: plain pointers and sizes, no C++ or torch types.  Every function returns 0 on success and a
 * negative spm_status on failure; spm_hip_last_error() gives the message.  Nothing throws across the ABI.
 * Handles are opaque and owned by the library.  Not thread-safe per context; contexts are independent.
 */
#ifndef SPM_HIP_H
#define SPM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct spm_ctx spm_ctx;           /* one device + one HIP stream */
typedef struct spm_text spm_text;         /* haystack resident in HBM, one uint8 rank per symbol */
typedef struct spm_patterns spm_patterns; /* compiled needle set: Peq / mask tables, seed index */
typedef struct spm_hits spm_hits;         /* result of one scan */
typedef struct spm_jst spm_jst;           /* journaled sequence tree: reference + alleles + coverage, resident in HBM */
typedef struct spm_jst_hits spm_jst_hits; /* result of one search over all haplotypes */

enum spm_status {
    SPM_OK = 0,
    SPM_E_INVALID = -1,     /* bad argument */
    SPM_E_HIP = -2,         /* HIP runtime error (message has the hipError string) */
    SPM_E_NOMEM = -3,
    SPM_E_UNSUPPORTED = -4, /* e.g. needle longer than SPM_MAX_NEEDLE */
    SPM_E_OVERFLOW = -5,    /* hit buffer too small; see spm_scan_opts.max_hits */
    SPM_E_PEER = -6         /* multi-GPU exchange: another rank reported an error; nothing was sent or received */
};

/* Which reference matcher a pattern set stands for. */
enum spm_algo {
    SPM_ALGO_SHIFTOR = 0,      /* spm::shiftor_matcher   matcher/shiftor_matcher.hpp:20-41  -> begin positions */
    SPM_ALGO_MYERS = 1,        /* spm::myers_matcher     matcher/myers_matcher.hpp:19-54    -> (end, distance) */
    SPM_ALGO_MYERS_PREFIX = 2, /* spm::restorable_myers_prefix_matcher  matcher/myers_prefix_matcher_restorable.hpp:117-160 */
    SPM_ALGO_HORSPOOL = 3      /* spm::horspool_matcher  matcher/horspool_matcher.hpp:20-41 -> begin positions
                                  (same hit set as Shift-Or: every occurrence) */
};

enum spm_engine {
    SPM_ENGINE_AUTO = 0,  /* seed filter when the pattern set admits one, else brute force */
    SPM_ENGINE_BRUTE = 1, /* one lane per pattern, every text symbol through the recurrence */
    SPM_ENGINE_FILTER = 2 /* lossless pigeonhole seed filter + bit-vector verification */
};

#define SPM_MAX_NEEDLE 2048u

/* One hit record, 16 bytes.  Replaces what the reference's callback reads off the seqan2::Finder:
 *   Myers:  pos = seqan2::endPosition(finder)   (exclusive end; test/api/libspm/matcher/myers_matcher_test.cpp:49-51)
 *           score = edit distance of that end position (the pattern state's `errors`)
 *   exact:  pos = seqan2::beginPosition(finder) (test/api/libspm/matcher/horspool_matcher_test.cpp:48-50), score = 0
 * `pattern` = index into the pattern set.  Positions are relative to text[0] plus spm_scan_opts.pos_offset. */
typedef struct spm_hit {
    uint64_t pos;
    uint32_t pattern;
    int32_t score;
} spm_hit;

typedef struct spm_scan_opts {
    uint32_t engine;       /* spm_engine */
    uint32_t left_context; /* 0: text[begin] is the first symbol of the haystack (cold start, the reference's
                              operator() semantics).  1: the symbols before `begin` belong to the same haystack
                              and may be read as warm-up, so the hits equal those of a scan of the whole text
                              whose last symbol lies in [begin,end) -- the shard rule of SURVEY.md 8(e). */
    uint64_t pos_offset;   /* added to every reported position (global coordinate of text[0]) */
    uint64_t max_hits;     /* capacity of the hit buffer; 0 = library default */
    uint32_t flags;        /* SPM_SCAN_* bits.  (The host view spm_hip_hits_view is always sorted by (pattern, pos);
                              the device view is in arrival order.) */
    uint32_t reserved;
} spm_scan_opts;

#define SPM_SCAN_IGNORE_PACKED 1u /* do not use the text's 2-bit shadow even if it has one */
/* Deferred completion.  spm_hip_scan normally returns when it KNOWS the scan is complete: it reads the device counters
 * back once (one host synchronisation), because a list that proved too small means another attempt.  With this flag a
 * whole (unsegmented), stateless scan through the seed filter returns as soon as its kernels are enqueued; the counters are
 * read at the first accessor that needs the hit count (view, device, copy_device, copy_fused, stats, destroy), and a scan
 * that does need attention -- a list overflowed, a span gave up, or an EARLIER deferred scan left the context's band table
 * in a state this one could not trust -- is repeated there, synchronously, retries and fallbacks included.  The text and
 * the needle set must stay alive until then.  Stateful and segmented scans and the brute-force engine ignore the flag.
 * With spm_hip_hits_copy_fused_device a step of a scan loop has no host synchronisation at all: the GPU never waits for
 * the host between steps (the status word of its header says whether the host has to look). */
#define SPM_SCAN_DEFER 2u

/* Per-scan device timings, HIP events on the context's stream (ms). */
typedef struct spm_scan_stats {
    float ms_total;
    float ms_main;        /* the dominant kernel: brute-force scan, or the seed filter */
    float ms_verify;      /* filter engine: bit-vector verification of the candidates */
    uint32_t engine_used; /* spm_engine actually run */
    uint32_t fell_back;   /* 1 if the filter engine gave up on the WHOLE scan and the brute engine re-ran it (dedupe set or
                             overflow list exhausted, or a segmented scan overflowed); see fallback_spans for the usual,
                             span-local form */
    uint64_t n_candidates;
    uint64_t n_hits;
    uint32_t main_launches;
    uint32_t n_bands;     /* filter engine: what was actually verified -- diagonal bands after candidate merging (sets with
                             k >= 8), else the candidates whose whole seed matches the text */
    uint32_t fallback_spans;   /* filter engine: spans of the text whose seed hits exceeded their budget (repeat-rich
                                  stretches); only those were scanned again by the brute-force kernel */
    uint32_t reserved;
    uint64_t fallback_symbols; /* ... and how many text symbols that re-scan covered */
} spm_scan_stats;

/* ---- context -------------------------------------------------------------------------------------- */
/* stream: a hipStream_t to run on (e.g. torch's current stream), or NULL to create a private one. */
int spm_hip_init(int device, void *stream, spm_ctx **out);
void spm_hip_destroy(spm_ctx *ctx);
const char *spm_hip_last_error(const spm_ctx *ctx); /* ctx may be NULL: error of the failed spm_hip_init */
int spm_hip_synchronize(spm_ctx *ctx);

/* ---- haystack: replaces spm::make_seqan_container(views::all(haystack)), seqan_pattern_base.hpp:44-45 ----
 * sigma = alphabet size (4 dna4, 5 dna5, 15 dna15; seqan/alphabet.hpp:100-105); symbols are ranks < sigma. */
int spm_hip_text_upload(spm_ctx *ctx, const uint8_t *ranks, uint64_t n, uint32_t sigma, spm_text **out);
/* Borrow a device buffer (16-byte aligned) that the caller keeps alive, e.g. a torch uint8 tensor.  The buffer is read
 * once (at HBM speed) to check that every symbol is a rank < sigma; SPM_E_INVALID otherwise -- the same contract as
 * spm_hip_text_upload.  The caller must not change it while the handle lives. */
int spm_hip_text_wrap(spm_ctx *ctx, const void *device_ranks, uint64_t n, uint32_t sigma, spm_text **out);
/* Synthetic uniform dna4 text generated in HBM: base(i) of SURVEY.md 8(d) for i in [global_begin, +n). */
int spm_hip_text_generate(spm_ctx *ctx, uint64_t seed, uint64_t global_begin, uint64_t n, spm_text **out);
/* Synthetic repeat-rich dna4 text (bench workload c3r): the uniform text above with `repeat_ppm` parts per million of
 * its bases inside tandem-repeat / low-complexity stretches of 16..256 bases (libspm_amd/csrc/synth.hpp says exactly
 * how).  spm_hip_synth_repeat_text regenerates any slice on the host; spm_hip_synth_repeat_pattern cuts needle p from
 * it at a uniformly random position, and every `across_every`-th one (0: none) across a stretch on purpose. */
int spm_hip_text_generate_repeats(spm_ctx *ctx, uint64_t seed, uint64_t global_begin, uint64_t n, uint32_t repeat_ppm,
                                  spm_text **out);
/* Optional: build a 2-bit shadow of a dna4 haystack (16 symbols per uint32, +25 % HBM).  Later seed-filter scans of
 * this text stream the shadow instead of the 1-byte ranks -- a quarter of the HBM traffic -- and return identical hits;
 * verification and the brute-force engine keep reading the original ranks.  Meant for a reference that is scanned
 * against many needle batches.  SPM_E_INVALID if the text holds a symbol >= 4 or is not dna4.
 * spm_scan_opts.flags & SPM_SCAN_IGNORE_PACKED makes a scan read the 1-byte text anyway. */
int spm_hip_text_pack(spm_ctx *ctx, spm_text *text);
int spm_hip_text_is_packed(const spm_text *text);
int spm_hip_text_download(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t n, uint8_t *out);
uint64_t spm_hip_text_length(const spm_text *text);
const void *spm_hip_text_device_ptr(const spm_text *text);
void spm_hip_text_destroy(spm_text *text);

/* ---- needles: replaces the matcher constructors (myers_matcher.hpp:40-43, shiftor_matcher.hpp:38-40,
 * horspool_matcher.hpp:38-40, myers_matcher_restorable.hpp:132).  ranks_concat holds the needles back to back,
 * needle p = ranks_concat[offsets[p] .. offsets[p+1]).  k[p] = max_error_count of needle p (NULL = all 0;
 * ignored by the exact matchers). */
int spm_hip_patterns_create(spm_ctx *ctx, int algo, const uint8_t *ranks_concat, const uint32_t *offsets,
                            uint32_t n_patterns, const uint16_t *k, uint32_t sigma, spm_patterns **out);
void spm_hip_patterns_destroy(spm_patterns *p);
/* spm::window_size (seqan_pattern_base.hpp:97-99, myers_matcher.hpp:51-53): |P| exact, |P|+k Myers, 0 if empty. */
uint64_t spm_hip_patterns_window_size(const spm_patterns *p, uint32_t pattern);
/* 1 if the set admits the lossless seed filter (dna4, every needle long enough for its k). */
int spm_hip_patterns_filterable(const spm_patterns *p);

/* What spm_hip_patterns_create spent where (host wall clock, ms) and what it built.  The reference's constructors are
 * O(|P|) per needle (myers_matcher.hpp:40-43); a set of 100 000 needles is built by `threads` host threads here
 * (SPM_HIP_BUILD_THREADS; default: the hardware concurrency, at most 16). */
typedef struct spm_build_stats {
    float ms_total;
    float ms_tables;  /* match-mask tables of the bit-vector engines */
    float ms_index;   /* seed index of the filter engine */
    float ms_upload;  /* the device allocation + the host-to-device stream of every table (pinned chunks) */
    uint32_t threads;
    uint32_t passes;            /* passes of the seed filter over the text per scan (0: brute-force engine only) */
    uint32_t dense;             /* 1: the one dense pass (presence bits in LDS + fingerprint buckets in L2) */
    uint32_t anchor_sixteenths; /* sixteenths of all text windows that are looked up, summed over the passes */
    uint64_t keys;              /* indexed windows */
    uint32_t stride, key_len;
    uint64_t bytes_device;      /* the set's one device allocation (every table, 256-byte aligned) */
} spm_build_stats;
int spm_hip_patterns_build_stats(const spm_patterns *p, spm_build_stats *out);

/* ---- matcher state: replaces capture()/restore() (myers_matcher_restorable.hpp:57-63,136-142;
 * shiftor_matcher_restorable.hpp:44-50).  A state blob holds one record per pattern, each
 * spm_hip_patterns_state_stride() bytes:
 *   Myers:    int32 score; uint32 n_words; uint64 vp[n_words]; uint64 vn[n_words]     (n_words = ceil(|P|/64))
 *   Shift-Or: uint32 n_words; uint32 pad;  uint32 r[n_words]                          (n_words = ceil(|P|/32))
 * spm_hip_patterns_state_init writes the constructor-time state (VP=~0, VN=0, score=|P|; R=~0). */
size_t spm_hip_patterns_state_stride(const spm_patterns *p);
int spm_hip_patterns_state_init(const spm_patterns *p, void *state);

/* ---- scan: replaces seqan_pattern_base::operator()(haystack, callback), seqan_pattern_base.hpp:40-52 ----
 * Scans text[begin,end).  state_in == NULL: fresh matcher (non-restorable semantics).  state_in != NULL:
 * continue from that state (restorable semantics, myers_matcher_restorable.hpp:72-82); state_out (may alias
 * state_in, may be NULL) receives the state after the last symbol. */
int spm_hip_scan(spm_ctx *ctx, const spm_text *text, uint64_t begin, uint64_t end, const spm_patterns *patterns,
                 const spm_scan_opts *opts, const void *state_in, void *state_out, spm_hits **out);

/* Batch of independent haystacks stored back to back in one text: haystack s = text[seg_offsets[s], seg_offsets[s+1])
 * (n_segments + 1 ascending host offsets).  Each one is scanned as seqan_pattern_base::operator() would scan it on its
 * own -- cold start at its first symbol, no hit spans two haystacks -- in ONE launch.  Positions are reported relative to
 * text[0]; the caller maps them to (segment, local position).  Used by the journaled-sequence traversal, whose
 * variant contexts are thousands of short haystacks. */
int spm_hip_scan_segments(spm_ctx *ctx, const spm_text *text, const uint64_t *seg_offsets, uint64_t n_segments,
                          const spm_patterns *patterns, const spm_scan_opts *opts, spm_hits **out);

/* ---- hits ------------------------------------------------------------------------------------------ */
/* Host view, sorted by (pattern, pos): per pattern this is the order the reference's callback fires in. */
int spm_hip_hits_view(spm_hits *hits, const spm_hit **records, uint64_t *n);
/* Device view (arrival order, not sorted): pointer to n spm_hit records in HBM, for an RCCL gather. */
int spm_hip_hits_device(spm_hits *hits, const void **device_records, uint64_t *n);
/* Copy the first min(n, cap) records into a caller-owned device buffer (e.g. a torch tensor that an RCCL
 * send/recv will read), asynchronously on the context's stream.  *n receives the number of hits. */
int spm_hip_hits_copy_device(spm_hits *hits, void *device_dst, uint64_t cap, uint64_t *n);
/* The same with a 16-byte header {n as uint64, 0} in front of the records: device_dst (16-byte aligned) holds cap + 1
 * records' worth of bytes; one kernel writes header and records.
 * This is the fixed-size [count | records] buffer one ncclAllGather per scan exchanges (libspm_amd.dist.gather_hits_fused):
 * the count is written from the host value the library already has, no second call from the caller. */
int spm_hip_hits_copy_fused(spm_hits *hits, void *device_dst, uint64_t cap, uint64_t *n);
/* The same without the host knowing the count: a kernel reads the scan's counters on the device and writes the header
 * {n as uint64, status as uint64} and the first min(n, cap) records.  status 0: the records are the scan's final result;
 * nonzero: the scan needs the host's attention (a list overflowed, spans gave up) -- call an accessor (which completes
 * the scan) and copy again.  Does not synchronise; completes nothing. */
int spm_hip_hits_copy_fused_device(spm_hits *hits, void *device_dst, uint64_t cap);
int spm_hip_hits_stats(const spm_hits *hits, spm_scan_stats *out);
/* order-independent checksum: sum over hits of mix64(pos ^ pattern<<40 ^ score<<58), SURVEY.md 8(d) */
uint64_t spm_hip_hits_checksum(spm_hits *hits);
void spm_hip_hits_destroy(spm_hits *hits);

/* ---- journaled-sequence (pan-genome) search, config C5 ----------------------------------------------------
 * The reference only designs the journaled sequence (specs/journaled_sequence_class_diagram.drawio:7-298) and gives the
 * matcher-side hooks a traverser needs (spm::window_size / capture / restore, matcher/concept.hpp:26-161).  The
 * contract implemented here is SURVEY.md 8(f)-2: the hit set equals the union over haplotypes of a linear scan of each
 * materialised haplotype, reported as (haplotype, position in haplotype coordinates).
 *
 * Device-side scheme: the reference axis is cut into blocks; (block, haplotype) pairs whose haplotype-local sequence
 * plus window-1 symbols of left context are byte-identical are found by an exact allele-set signature, one
 * representative of each is spelled out into a context buffer, the buffer is scanned as independent segments by the
 * same kernels as spm_hip_scan_segments, and every hit is fanned out to the haplotypes sharing its context.
 *
 * Alleles: sorted by `pos` (ties keep the given order); allele i replaces reference[pos, pos + ref_len) by
 * alt_pool[alt_off, alt_off + alt_len).  coverage: n_alleles x ceil(n_haplotypes / 64) words, bit h of row i set iff
 * haplotype h carries allele i.  Two alleles that overlap on the reference (pos_j < pos_i + ref_len_i for i < j) must
 * have disjoint coverage (multi-allelic sites); otherwise SPM_E_UNSUPPORTED.  At most 65 535 haplotypes; contexts are
 * shared among the haplotypes of one group of 1024. */
typedef struct spm_jst_allele {
    uint64_t pos;
    uint32_t ref_len;
    uint32_t alt_len;
    uint64_t alt_off;
} spm_jst_allele;

typedef struct spm_jst_hit {
    uint64_t pos;       /* what the matcher reports (Myers: exclusive end; exact: begin), haplotype coordinates */
    uint32_t haplotype;
    uint32_t pattern;
    int32_t score;
    uint32_t reserved;
} spm_jst_hit;

typedef struct spm_jst_stats {
    uint64_t haplotype_symbols; /* sum of haplotype lengths over the indexed blocks: what per-haplotype scans read */
    uint64_t context_symbols;   /* symbols laid out in the context buffer (what the device streams per search) */
    uint64_t contexts;          /* non-empty (block, haplotype) pairs */
    uint64_t unique_contexts;
    uint64_t n_blocks;
    uint32_t block_len;
    uint32_t window;
    float ms_index;             /* build of the context index (once per window size) */
    float ms_scan;              /* last search: segment scan incl. verification */
    float ms_main;              /* last search: the scan's main kernel(s) (seed filter / brute force) */
    float ms_verify;            /* last search: verification of the filter's candidates */
    float ms_fanout;            /* last search: hit fan-out to haplotypes */
    uint32_t engine_used;
    uint32_t main_launches;
    uint32_t fell_back;         /* last search: 1 if the seed filter overflowed and the brute engine re-ran the scan */
    uint64_t segment_hits;      /* last search: hits in context coordinates, before the fan-out */
    uint64_t candidates;        /* last search: seed-filter candidates */
    uint64_t bands;             /* last search: diagonal bands verified after candidate merging (0: not merged) */
} spm_jst_stats;

/* `reference` must stay alive as long as the tree (it is not copied). */
int spm_hip_jst_create(spm_ctx *ctx, const spm_text *reference, const spm_jst_allele *alleles, uint64_t n_alleles,
                       const uint8_t *alt_pool, uint64_t alt_pool_len, const uint64_t *coverage,
                       uint32_t n_haplotypes, spm_jst **out);
void spm_hip_jst_destroy(spm_jst *jst);
uint64_t spm_hip_jst_haplotype_length(const spm_jst *jst, uint32_t haplotype);
/* Symbols [begin, begin + n) of haplotype h (host walk over the allele table; for needles and tests). */
int spm_hip_jst_extract(spm_jst *jst, uint32_t haplotype, uint64_t begin, uint64_t n, uint8_t *out);
/* Build the context index for needle sets whose spm::window_size is <= window.  block_len = reference positions per
 * block (0 = library default); only blocks [block_begin, block_end) are indexed (block_end = 0: all) -- the shard of
 * one GPU when a tree is searched by several, SURVEY.md 8(e). */
int spm_hip_jst_index(spm_jst *jst, uint32_t window, uint32_t block_len, uint64_t block_begin, uint64_t block_end);
int spm_hip_jst_search(spm_jst *jst, const spm_patterns *patterns, const spm_scan_opts *opts, spm_jst_hits **out);
int spm_hip_jst_stats(const spm_jst *jst, spm_jst_stats *out);
/* Host view sorted by (haplotype, pos, pattern); device view in arrival order (for an RCCL gather). */
int spm_hip_jst_hits_view(spm_jst_hits *hits, const spm_jst_hit **records, uint64_t *n);
int spm_hip_jst_hits_device(spm_jst_hits *hits, const void **device_records, uint64_t *n);
/* Copy the first min(n, cap) records into a caller-owned device buffer, asynchronously on the context's stream. */
int spm_hip_jst_hits_copy_device(spm_jst_hits *hits, void *device_dst, uint64_t cap, uint64_t *n);
void spm_hip_jst_hits_destroy(spm_jst_hits *hits);
/* Synthetic variants of config C5: one SNP per 1000 reference bases, one indel of length 1..50 per
 * 10 000, each carried by a random non-empty subset of n_haplotypes <= 64; the reference is the synthetic text of
 * `seed_text`.  Call with alleles == NULL to get the counts (*n_alleles, *alt_pool_len) first.  Host side. */
int spm_hip_jst_synth_variants(uint64_t seed_text, uint64_t seed_var, uint64_t ref_begin, uint64_t n_ref,
                               uint32_t n_haplotypes, spm_jst_allele *alleles, uint64_t *n_alleles, uint8_t *alt_pool,
                               uint64_t *alt_pool_len, uint64_t *coverage);

/* ---- multi-GPU exchange: the gatherv of hit records to one rank over RCCL (xGMI), SURVEY.md 8(e) -------------------
 * One process per GPU; the path shards by text position (spm_scan_opts.left_context / pos_offset) and needs no
 * data-path collective.  The single exchange step is this gatherv (RCCL has none of its own: one ncclAllGather of the
 * per-rank counts, then grouped ncclSend / ncclRecv).  librccl.so is opened when the first communicator is made.
 *   rank 0:      spm_hip_comm_unique_id(id)  -> hand the 128 bytes to the other ranks (MPI, a file, a socket ...)
 *   every rank:  spm_hip_comm_init(ctx, id, rank, world, &comm)
 *   per scan:    spm_hip_gatherv_hits(comm, hits, 0, &records, &n, counts)
 * Records arrive in rank order (= ascending shard order); on the root *device_records points at n_total records in HBM
 * (owned by the communicator, valid until its next gatherv), elsewhere it is NULL.  counts may be NULL. */
typedef struct spm_comm spm_comm;
int spm_hip_comm_unique_id(void *id128);
int spm_hip_comm_init(spm_ctx *ctx, const void *unique_id128, int rank, int world, spm_comm **out);
void spm_hip_comm_destroy(spm_comm *comm);
int spm_hip_gatherv_hits(spm_comm *comm, spm_hits *local, int root, const void **device_records, uint64_t *n_total,
                         uint64_t *counts);
int spm_hip_gatherv_jst_hits(spm_comm *comm, spm_jst_hits *local, int root, const void **device_records,
                             uint64_t *n_total, uint64_t *counts);
/* host arithmetic of the gatherv: byte offset of every rank's records in the root's buffer, offsets[world] = total */
int spm_hip_gatherv_plan(const uint64_t *counts, uint32_t world, uint32_t record_bytes, uint64_t *offsets);
/* Failure is collective: a rank whose local result is unusable (e.g. SPM_E_OVERFLOW of its scan), a root that cannot hold
 * the records, counts that overflow the offsets -- all ranks learn of it in an exchange every rank takes part in and return
 * an error (the failing rank its own, the others SPM_E_PEER) BEFORE any send or receive is posted; nobody is left waiting.
 * Host-only self-check of that protocol over an in-process loopback of `world` threads (no GPU, no RCCL): scenario 0 clean,
 * 1 rank `victim` has a local error, 2 the root cannot reserve its buffer, 3 the counts overflow, 4 no rank has records.
 * detail[world] (may be NULL) receives the status every rank returned.  SPM_OK iff all ranks behaved as promised and
 * nothing hung. */
int spm_hip_comm_selftest(int world, int root, int scenario, int victim, uint32_t record_bytes, uint64_t seed, int *detail);

/* ---- synthetic needles of the benchmark configs (host side) -------------------------- */
uint64_t spm_hip_synth_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                               uint32_t kmax, uint8_t *out);
uint64_t spm_hip_synth_repeat_pattern(uint64_t seed_text, uint64_t seed_pat, uint64_t n_total, uint32_t p, uint32_t L,
                                      uint32_t kmax, uint32_t repeat_ppm, uint32_t across_every, uint8_t *out);
void spm_hip_synth_repeat_text(uint64_t seed, uint32_t repeat_ppm, uint64_t begin, uint64_t n, uint8_t *out);
uint64_t spm_hip_mix64(uint64_t z);

/* Host-only self-check of the seed index (no device, no context): builds the level-1 / level-2 tables exactly as
 * spm_hip_patterns_create does and verifies what the filter's losslessness rests on -- every indexed 16-symbol window of
 * every seed is found at both levels, every needle sits in exactly one pass (anchored sets: every seed has its one key
 * in one pass, beginning with an anchor dimer of that pass), the stride fits every seed.
 * stats[8] = {passes, stride, keys, windows checked, windows missing, level-1 false positives, trials,
 *            hash variant | anchor dimers per pass << 8 (0: unanchored)}.
 * Returns SPM_OK iff nothing is missing. */
int spm_hip_host_selftest(int algo, const uint8_t *ranks_concat, const uint32_t *offsets, uint32_t n_patterns,
                          const uint16_t *k, uint32_t sigma, uint64_t *stats);

const char *spm_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SPM_HIP_H */
