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
// Limit of this first version: haplotypes are materialised on the host to cut the contexts (fine for the reference's
// 100 x 10 kb fixtures; a pan-genome-scale build would cut contexts from the journals directly).
#pragma once

#include <algorithm>
#include <map>
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

    // Search a compiled needle set over every haplotype.  `window` = max spm::window_size of the set, `needle_len`
    // = per-needle lengths (exact matchers report the begin position, so the last symbol is begin + |P| - 1).
    std::vector<jst_hit> search(spm_patterns * needles, std::size_t window, std::vector<std::uint32_t> const & needle_len,
                                bool reports_begin, std::size_t block = 0, jst_search_stats * stats = nullptr) const
    {
        spm_ctx * ctx = hip::default_context();
        std::size_t const H = haplotype_count();
        std::size_t const L = block ? block : std::max<std::size_t>(256, 4 * window);
        std::size_t const n_blocks = (_reference.size() + L - 1) / L;

        struct member
        {
            std::uint32_t haplotype;
            std::uint64_t ctx_lo; // haplotype coordinate of the context's first symbol
        };
        struct context
        {
            std::size_t offset{}, length{}, owned_from{}; // inside the device buffer / local index of the first owned symbol
            std::vector<member> members;
        };
        std::vector<context> contexts;
        std::vector<std::uint8_t> buffer;
        std::unordered_map<std::string, std::size_t> index; // (owned_from, bytes) -> context id
        jst_search_stats st{};

        std::vector<std::pair<std::size_t, std::ptrdiff_t>> map;
        for (std::size_t h = 0; h < H; ++h) {
            std::vector<std::uint8_t> const hap = haplotype(h, &map).materialize();
            st.haplotype_symbols += hap.size();
            auto to_hap = [&](std::size_t ref_pos) -> std::size_t {
                // monotone map: the haplotype position of the first symbol derived from reference >= ref_pos
                // (inside a deleted stretch the value is only approximate; block borders are made monotone below, and
                // any monotone partition of the haplotype is exact -- the map only steers how well contexts dedupe)
                std::ptrdiff_t shift = 0;
                for (auto const & [p, s] : map) {
                    if (p > ref_pos)
                        break;
                    shift = s;
                }
                std::ptrdiff_t const v = static_cast<std::ptrdiff_t>(ref_pos) + shift;
                return static_cast<std::size_t>(std::clamp<std::ptrdiff_t>(v, 0, static_cast<std::ptrdiff_t>(hap.size())));
            };
            std::size_t prev_b = 0;
            for (std::size_t j = 0; j < n_blocks; ++j) {
                std::size_t a = j == 0 ? 0 : prev_b;
                std::size_t b = j + 1 == n_blocks ? hap.size() : std::max(a, to_hap((j + 1) * L));
                prev_b = b;
                if (b == a)
                    continue;
                std::size_t const lo = a >= window - 1 ? a - (window - 1) : 0;
                std::string key(reinterpret_cast<char const *>(hap.data() + lo), b - lo);
                std::size_t const owned_from = a - lo; // part of the identity: same bytes, same ownership
                key.push_back(static_cast<char>(owned_from & 0xFF));
                key.push_back(static_cast<char>((owned_from >> 8) & 0xFF));
                key.push_back(static_cast<char>((owned_from >> 16) & 0xFF));
                ++st.contexts;
                auto [it, fresh] = index.try_emplace(std::move(key), contexts.size());
                if (fresh) {
                    context c;
                    c.offset = buffer.size();
                    c.length = b - lo;
                    c.owned_from = owned_from;
                    buffer.insert(buffer.end(), hap.begin() + static_cast<std::ptrdiff_t>(lo),
                                  hap.begin() + static_cast<std::ptrdiff_t>(b));
                    contexts.push_back(std::move(c));
                }
                contexts[it->second].members.push_back({static_cast<std::uint32_t>(h), lo});
            }
        }
        st.unique_contexts = contexts.size();
        st.context_symbols = buffer.size();
        if (stats)
            *stats = st;
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
        spm_hits * hh = nullptr;
        if (spm_hip_scan_segments(ctx, text.get(), seg.data(), contexts.size(), needles, &opts, &hh) != SPM_OK)
            hip::fatal("spm_hip_scan_segments", ctx);
        hip::hits_ptr hits{hh};
        spm_hit const * rec = nullptr;
        std::uint64_t n = 0;
        if (spm_hip_hits_view(hits.get(), &rec, &n) != SPM_OK)
            hip::fatal("spm_hip_hits_view", ctx);
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
