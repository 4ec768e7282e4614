// libspm/jst/journaled_sequence_tree.hpp -- a reference sequence + variants shared by many haplotypes, and the
// search of a needle set over ALL haplotypes on the MI355X (config C5 of BASELINE.json).
//
// The reference has no traversal code -- only the journaled-sequence design (specs/...class_diagram.drawio) and the
// matcher-side hooks a traverser needs (window_size / capture / restore, matcher/concept.hpp:26-161).  The contract
// taken here is the one SURVEY 8f-2 states: the hit set must equal the union over haplotypes of a linear scan of each
// materialised haplotype, reported as (haplotype, position).
//
// How (GPU-first): the reference axis is cut into blocks; for every haplotype the block's haplotype-local sequence
// plus window_size-1 symbols of left context is a *context*.  Haplotypes that carry the same alleles around a block
// have byte-identical contexts, so contexts are deduplicated and only the UNIQUE ones are laid out back to back and
// scanned -- in one launch -- as independent haystacks (spm_hip_scan_segments).  A hit is owned by the context whose
// block contains its last symbol and is fanned out to every haplotype sharing that context.  By the window property
// (a hit depends only on the window_size symbols ending at it) this is exact.  Work on the device is proportional to
// the distinct sequence content, not to haplotypes x length.
//
// No haplotype is materialised to do this: contexts are cut and deduplicated by *signature* (start point in
// reference/alt space + the alleles the stretch touches) straight from the per-haplotype allele lists, and only one
// representative per signature is spelled out.  Host work is blocks x haplotypes x (alleles per block) plus the unique
// context bytes.
#pragma once

#include <algorithm>
#include <map>
#include <memory>
#include <string_view>
#include <unordered_map>

#include <libspm/hip/context.hpp>
#include <libspm/jst/io.hpp>
#include <libspm/jst/journaled_sequence.hpp>

namespace spm
{
struct jst_hit
{
    std::uint32_t haplotype;
    std::uint64_t position; // what the matcher reports (Myers: exclusive end; exact: begin), haplotype coordinates
    std::uint32_t needle;
    std::int32_t errors;
    bool operator==(jst_hit const &) const noexcept = default;
    auto operator<=>(jst_hit const &) const noexcept = default;
};

struct jst_search_stats
{
    std::uint64_t haplotype_symbols{}; // sum of haplotype lengths (what per-haplotype scans would read)
    std::uint64_t context_symbols{};   // symbols actually laid out for the device after deduplication
    std::uint64_t contexts{}, unique_contexts{};
};

class journaled_sequence_tree
{
    std::vector<std::uint8_t> _reference;
    io::vcf_data _variants;

    // the tree resident on the MI355X (reference text + allele table + context index), built on first use
    struct device_tree
    {
        hip::text_ptr reference{};
        spm_jst * tree{};
        bool tried{}, usable{};
        std::size_t window{}, block{};
        ~device_tree()
        {
            if (tree)
                spm_hip_jst_destroy(tree);
        }
    };
    std::shared_ptr<device_tree> _device{std::make_shared<device_tree>()};

public:
    journaled_sequence_tree(std::vector<std::uint8_t> reference, io::vcf_data variants) :
        _reference{std::move(reference)}, _variants{std::move(variants)}
    {}

    std::size_t haplotype_count() const noexcept { return _variants.n_haplotypes; }
    std::vector<std::uint8_t> const & reference() const noexcept { return _reference; }

    // Haplotype h as a journaled sequence over the reference (alleles applied from the right so that reference
    // positions stay valid).  `ref_to_hap`, if given, receives the breakpoints of the monotone coordinate map:
    // reference position p maps to p + shift of the last breakpoint at or before p.
    journaled_sequence<std::uint8_t> haplotype(std::size_t h,
                                               std::vector<std::pair<std::size_t, std::ptrdiff_t>> * ref_to_hap = nullptr) const
    {
        journaled_sequence<std::uint8_t> js{std::span<std::uint8_t const>{_reference}};
        std::vector<io::vcf_allele const *> mine;
        for (auto const & a : _variants.alleles)
            if (a.coverage[h])
                mine.push_back(&a);
        for (auto it = mine.rbegin(); it != mine.rend(); ++it) {
            io::vcf_allele const & a = **it;
            js.replace(js.begin() + static_cast<std::ptrdiff_t>(a.pos),
                       js.begin() + static_cast<std::ptrdiff_t>(std::min(a.pos + a.ref_len, _reference.size())),
                       std::span<std::uint8_t const>{a.alt});
        }
        if (ref_to_hap) {
            ref_to_hap->clear();
            std::ptrdiff_t shift = 0;
            for (io::vcf_allele const * a : mine) {
                std::ptrdiff_t const d = static_cast<std::ptrdiff_t>(a->alt.size()) - static_cast<std::ptrdiff_t>(a->ref_len);
                if (d != 0) {
                    shift += d;
                    ref_to_hap->emplace_back(a->pos + a->ref_len, shift); // positions behind the allele are shifted
                }
            }
        }
        return js;
    }

    // Per-haplotype event table: the alleles it carries, in reference order, with running coordinate shifts.
    // Haplotype symbols are generated by walking the reference: an allele at p emits its alt when the walk reaches p
    // and skips ref_len reference symbols.  Symbol ownership for blocking: an alt symbol belongs to reference
    // position p, a reference symbol to its own position.
    struct hap_events
    {
        std::vector<std::uint32_t> id;       // index into _variants.alleles
        std::vector<std::uint64_t> p, rend;  // allele reference interval [p, rend)
        std::vector<std::uint64_t> hs;       // haplotype position of the first alt symbol
        std::vector<std::int64_t> cs;        // cumulative shift (alt - ref lengths) after this allele
        std::uint64_t length{};
    };

    hap_events events_of(std::size_t h) const
    {
        hap_events E;
        std::int64_t shift = 0;
        std::uint64_t last_end = 0;
        for (std::size_t i = 0; i < _variants.alleles.size(); ++i) {
            io::vcf_allele const & a = _variants.alleles[i];
            if (!a.coverage[h])
                continue;
            std::uint64_t const p = std::max<std::uint64_t>(a.pos, last_end); // overlapping alleles: clip (none in the fixtures)
            std::uint64_t const rend = std::max<std::uint64_t>(p, std::min<std::uint64_t>(a.pos + a.ref_len, _reference.size()));
            E.id.push_back(static_cast<std::uint32_t>(i));
            E.p.push_back(p);
            E.rend.push_back(rend);
            E.hs.push_back(static_cast<std::uint64_t>(static_cast<std::int64_t>(p) + shift));
            shift += static_cast<std::int64_t>(a.alt.size()) - static_cast<std::int64_t>(rend - p);
            E.cs.push_back(shift);
            last_end = rend;
        }
        E.length = static_cast<std::uint64_t>(static_cast<std::int64_t>(_reference.size()) + shift);
        return E;
    }

    // haplotype position of the first symbol owned by reference positions >= r
    static std::uint64_t owned_from(hap_events const & E, std::uint64_t r)
    {
        std::size_t const n = static_cast<std::size_t>(std::lower_bound(E.p.begin(), E.p.end(), r) - E.p.begin());
        if (n == 0)
            return r;
        return static_cast<std::uint64_t>(static_cast<std::int64_t>(std::max(r, E.rend[n - 1])) + E.cs[n - 1]);
    }

    // Append haplotype symbols [lo, hi) to `out`, and the signature of that stretch to `sig` (start point in
    // reference/alt space + the alleles it touches): equal signatures <=> byte-identical stretches.
    void emit_range(hap_events const & E, std::uint64_t lo, std::uint64_t hi, std::vector<std::uint8_t> * out,
                    std::vector<std::uint64_t> & sig) const
    {
        // first allele whose alt ends after lo
        std::size_t i = static_cast<std::size_t>(std::upper_bound(E.hs.begin(), E.hs.end(), lo) - E.hs.begin());
        if (i > 0 && lo < E.hs[i - 1] + _variants.alleles[E.id[i - 1]].alt.size())
            --i; // lo falls inside the alt of allele i-1
        std::uint64_t pos = lo;
        // reference position that haplotype position `pos` reads when it is in a reference run before allele i
        auto ref_of = [&](std::uint64_t hp, std::size_t idx) {
            return static_cast<std::uint64_t>(static_cast<std::int64_t>(hp) - (idx == 0 ? 0 : E.cs[idx - 1]));
        };
        bool first = true;
        while (pos < hi) {
            std::uint64_t const next_hs = i < E.hs.size() ? E.hs[i] : E.length;
            if (pos < next_hs) { // reference run up to the next allele
                std::uint64_t const r0 = ref_of(pos, i);
                std::uint64_t const n = std::min(hi, next_hs) - pos;
                if (first)
                    sig.push_back(r0 << 1); // start inside a reference run
                if (out)
                    out->insert(out->end(), _reference.begin() + static_cast<std::ptrdiff_t>(r0),
                                _reference.begin() + static_cast<std::ptrdiff_t>(r0 + n));
                pos += n;
            } else { // inside / at the alt of allele i
                std::vector<std::uint8_t> const & alt = _variants.alleles[E.id[i]].alt;
                std::uint64_t const off = pos - E.hs[i];
                if (first)
                    sig.push_back((off << 1) | 1); // start inside an alt, at this offset
                sig.push_back(static_cast<std::uint64_t>(E.id[i]) + (1ull << 40));
                if (off < alt.size()) {
                    std::uint64_t const n = std::min<std::uint64_t>(hi - pos, alt.size() - off);
                    if (out)
                        out->insert(out->end(), alt.begin() + static_cast<std::ptrdiff_t>(off),
                                    alt.begin() + static_cast<std::ptrdiff_t>(off + n));
                    pos += n;
                }
                ++i;
                if (pos < hi && i <= E.hs.size()) {
                    // a deletion (or the ref part of a replacement) was skipped: it is part of the signature above
                }
            }
            first = false;
        }
        // alleles that emit nothing but sit at the very end still matter only to later positions: not part of this range
        sig.push_back(hi - lo);
    }

    struct context_index
    {
        struct member
        {
            std::uint32_t haplotype;
            std::uint64_t ctx_lo; // haplotype coordinate of the context's first symbol
        };
        struct context
        {
            std::size_t offset{}, length{}, owned_from{};
            std::vector<member> members;
        };
        std::vector<context> contexts;
        std::vector<std::uint8_t> buffer;
        jst_search_stats stats{};
    };

    // Cut every haplotype into per-block contexts (block = `L` reference positions, plus window-1 symbols of left
    // context) and deduplicate them by signature -- no haplotype is materialised; work is proportional to
    // blocks x haplotypes x (alleles per block) plus the unique context bytes.
    context_index build_contexts(std::size_t window, std::size_t L) const
    {
        context_index X;
        std::size_t const H = haplotype_count();
        std::size_t const n_blocks = (_reference.size() + L - 1) / L;
        struct sig_hash
        {
            std::size_t operator()(std::vector<std::uint64_t> const & v) const noexcept
            {
                std::uint64_t h = 0x9E3779B97F4A7C15ull;
                for (std::uint64_t x : v) {
                    h ^= x + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
                    h *= 0xBF58476D1CE4E5B9ull;
                }
                return static_cast<std::size_t>(h ^ (h >> 31));
            }
        };
        std::unordered_map<std::vector<std::uint64_t>, std::size_t, sig_hash> index;
        std::vector<std::uint64_t> sig;
        for (std::size_t h = 0; h < H; ++h) {
            hap_events const E = events_of(h);
            X.stats.haplotype_symbols += E.length;
            std::uint64_t prev_b = 0;
            for (std::size_t j = 0; j < n_blocks; ++j) {
                std::uint64_t const a = j == 0 ? 0 : prev_b;
                std::uint64_t const b = j + 1 == n_blocks ? E.length : std::max(a, std::min(E.length, owned_from(E, (j + 1) * L)));
                prev_b = b;
                if (b == a)
                    continue;
                std::uint64_t const lo = a >= window - 1 ? a - (window - 1) : 0;
                sig.clear();
                sig.push_back(a - lo); // ownership is part of the identity
                emit_range(E, lo, b, nullptr, sig);
                ++X.stats.contexts;
                auto [it, fresh] = index.try_emplace(sig, X.contexts.size());
                if (fresh) {
                    typename context_index::context c;
                    c.offset = X.buffer.size();
                    c.length = static_cast<std::size_t>(b - lo);
                    c.owned_from = static_cast<std::size_t>(a - lo);
                    std::vector<std::uint64_t> ignore;
                    emit_range(E, lo, b, &X.buffer, ignore);
                    X.contexts.push_back(std::move(c));
                }
                X.contexts[it->second].members.push_back({static_cast<std::uint32_t>(h), lo});
            }
        }
        X.stats.unique_contexts = X.contexts.size();
        X.stats.context_symbols = X.buffer.size();
        return X;
    }

    // Search a compiled needle set over every haplotype.  `window` = max spm::window_size of the set, `needle_len`
    // = per-needle lengths (exact matchers report the begin position, so the last symbol is begin + |P| - 1).
    // The contexts are cut, deduplicated and spelled out on the device (spm_hip_jst_*, include/spm_hip.h); trees the
    // device path does not take (alleles overlapping on a shared haplotype, > 65 535 haplotypes) go through
    // search_host, which builds the same contexts on the host.  Both return the same hits.
    std::vector<jst_hit> search(spm_patterns * needles, std::size_t window, std::vector<std::uint32_t> const & needle_len,
                                bool reports_begin, std::size_t block = 0, jst_search_stats * stats = nullptr) const
    {
        if (device_ready())
            return search_device(needles, window, block, stats);
        return search_host(needles, window, needle_len, reports_begin, block, stats);
    }

    // true once the tree is resident on the device (first call uploads it)
    bool device_ready() const
    {
        device_tree & D = *_device;
        if (D.tried)
            return D.usable;
        D.tried = true;
        spm_ctx * ctx = hip::default_context();
        std::size_t const H = haplotype_count(), cw = (H + 63) / 64;
        std::vector<spm_jst_allele> al;
        std::vector<std::uint8_t> pool;
        std::vector<std::uint64_t> cov;
        al.reserve(_variants.alleles.size());
        cov.reserve(_variants.alleles.size() * cw);
        for (io::vcf_allele const & a : _variants.alleles) {
            al.push_back({a.pos, static_cast<std::uint32_t>(a.ref_len), static_cast<std::uint32_t>(a.alt.size()), pool.size()});
            pool.insert(pool.end(), a.alt.begin(), a.alt.end());
            for (std::size_t w = 0; w < cw; ++w) {
                std::uint64_t bits = 0;
                for (std::size_t h = w * 64; h < std::min(H, (w + 1) * 64); ++h)
                    bits |= static_cast<std::uint64_t>(a.coverage[h] != 0) << (h & 63);
                cov.push_back(bits);
            }
        }
        std::uint32_t sigma = 4;
        for (std::uint8_t c : _reference)
            sigma = std::max<std::uint32_t>(sigma, c + 1u);
        for (std::uint8_t c : pool)
            sigma = std::max<std::uint32_t>(sigma, c + 1u);
        spm_text * t = nullptr;
        if (H == 0 || spm_hip_text_upload(ctx, _reference.data(), _reference.size(), sigma, &t) != SPM_OK)
            return false;
        D.reference = hip::text_ptr{t};
        int const rc = spm_hip_jst_create(ctx, t, al.data(), al.size(), pool.data(), pool.size(), cov.data(),
                                          static_cast<std::uint32_t>(H), &D.tree);
        if (rc == SPM_E_HIP)
            hip::fatal("spm_hip_jst_create", ctx);
        D.usable = rc == SPM_OK;
        return D.usable;
    }

    std::vector<jst_hit> search_device(spm_patterns * needles, std::size_t window, std::size_t block = 0,
                                       jst_search_stats * stats = nullptr) const
    {
        spm_ctx * ctx = hip::default_context();
        if (!device_ready())
            hip::fatal("journaled_sequence_tree::search_device (tree not representable on the device)", ctx);
        device_tree & D = *_device;
        window = std::max<std::size_t>(window, 1);
        if (D.window != window || D.block != block) { // index once per (window, block)
            if (spm_hip_jst_index(D.tree, static_cast<std::uint32_t>(window), static_cast<std::uint32_t>(block), 0, 0) != SPM_OK)
                hip::fatal("spm_hip_jst_index", ctx);
            D.window = window;
            D.block = block;
        }
        spm_jst_stats st{};
        spm_hip_jst_stats(D.tree, &st);
        spm_scan_opts opts{};
        opts.max_hits = std::max<std::uint64_t>(1u << 22, 8 * st.context_symbols / window);
        spm_jst_hits * hh = nullptr;
        for (int attempt = 0;; ++attempt) { // (a hit buffer that proves too small is doubled, not fatal)
            int const rc = spm_hip_jst_search(D.tree, needles, &opts, &hh);
            if (rc == SPM_OK)
                break;
            if (rc != SPM_E_OVERFLOW || attempt >= 12)
                hip::fatal("spm_hip_jst_search", ctx);
            opts.max_hits *= 2;
        }
        spm_jst_hit const * rec = nullptr;
        std::uint64_t n = 0;
        if (spm_hip_jst_hits_view(hh, &rec, &n) != SPM_OK)
            hip::fatal("spm_hip_jst_hits_view", ctx);
        std::vector<jst_hit> out;
        out.reserve(n);
        for (std::uint64_t i = 0; i < n; ++i)
            out.push_back({rec[i].haplotype, rec[i].pos, rec[i].pattern, rec[i].score});
        spm_hip_jst_hits_destroy(hh);
        if (stats) {
            stats->haplotype_symbols = st.haplotype_symbols;
            stats->context_symbols = st.context_symbols;
            stats->contexts = st.contexts;
            stats->unique_contexts = st.unique_contexts;
        }
        std::sort(out.begin(), out.end());
        return out;
    }

    // The same search with the contexts built on the host (any allele table) and uploaded.
    std::vector<jst_hit> search_host(spm_patterns * needles, std::size_t window, std::vector<std::uint32_t> const & needle_len,
                                     bool reports_begin, std::size_t block = 0, jst_search_stats * stats = nullptr) const
    {
        spm_ctx * ctx = hip::default_context();
        std::size_t const L = block ? block : std::max<std::size_t>(256, 4 * window);
        context_index const X = build_contexts(window, L);
        auto const & contexts = X.contexts;
        auto const & buffer = X.buffer;
        using member = typename context_index::member;
        using context = typename context_index::context;
        if (stats)
            *stats = X.stats;
        std::vector<jst_hit> out;
        if (contexts.empty())
            return out;

        std::vector<std::uint64_t> seg(contexts.size() + 1);
        for (std::size_t i = 0; i < contexts.size(); ++i)
            seg[i] = contexts[i].offset;
        seg.back() = buffer.size();
        spm_text * t = nullptr;
        if (spm_hip_text_upload(ctx, buffer.data(), buffer.size(), 4, &t) != SPM_OK)
            hip::fatal("spm_hip_text_upload", ctx);
        hip::text_ptr text{t};
        spm_scan_opts opts{};
        opts.max_hits = std::max<std::uint64_t>(1u << 20, 4 * buffer.size() / std::max<std::size_t>(window, 1));
        spm_hit const * rec = nullptr;
        std::uint64_t n = 0;
        hip::hits_ptr hits = hip::scan_all_hits(
            ctx, opts,
            [&](spm_scan_opts const & o, spm_hits ** h) {
                return spm_hip_scan_segments(ctx, text.get(), seg.data(), contexts.size(), needles, &o, h);
            },
            rec, n, "spm_hip_scan_segments");
        for (std::uint64_t i = 0; i < n; ++i) {
            // context of this hit
            std::size_t const c =
                static_cast<std::size_t>(std::upper_bound(seg.begin(), seg.end(), rec[i].pos - (reports_begin ? 0 : 1)) -
                                         seg.begin()) - 1;
            context const & cx = contexts[c];
            std::uint64_t const local = rec[i].pos - cx.offset;
            std::uint64_t const last = reports_begin ? local + needle_len[rec[i].pattern] - 1 : local - 1;
            if (last < cx.owned_from)
                continue; // ends in the left context: owned by the previous block's context
            for (member const & m : cx.members)
                out.push_back({m.haplotype, m.ctx_lo + local, rec[i].pattern, rec[i].score});
        }
        std::sort(out.begin(), out.end());
        return out;
    }
};
} // namespace spm
