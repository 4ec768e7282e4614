// jst_cases.cpp -- journaled sequence + pan-genome search on the reference's own data:
//   tests/golden/jst/sim_ref_10Kb.fasta.gz, sim_ref_10Kb_SNPs.vcf, sim_ref_10Kb_SNP_INDELs.vcf and the fully
//   materialised haplotype FASTAs (copied data files of /root/reference/test/data, datasources.cmake:70-102).
// Checks: (1) journal invariants and edits (design: specs/journaled_sequence_class_diagram.drawio), (2) applying the
// VCF to the reference reproduces all 100 haplotypes of the fixture, (3) the deduplicated-context search on the GPU
// returns exactly the union of per-haplotype linear scans, (4) and the CPU oracle's hits on a sample.
#include <cstdio>
#include <cstring>
#include <string>

#include <libspm/jst/journaled_sequence_tree.hpp>

#include "../../oracle/spm_oracle.h"

static int failures = 0, checks = 0;
#define EXPECT_TRUE(cond)                                                                                              \
    do {                                                                                                               \
        ++checks;                                                                                                      \
        if (!(cond)) {                                                                                                 \
            ++failures;                                                                                                \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);                                              \
        }                                                                                                              \
    } while (0)

static std::string const DATA = std::string(SPM_TEST_DATA) + "/";

static void journal_cases()
{
    std::vector<std::uint8_t> const src{0, 1, 2, 3, 0, 1, 2, 3};
    std::vector<std::uint8_t> const ins{3, 3, 3};
    spm::journaled_sequence<std::uint8_t> js{std::span<std::uint8_t const>{src}};
    EXPECT_TRUE(js.size() == 8 && js.get_journal().check_journal_invariants());
    js.insert(js.begin() + 2, std::span<std::uint8_t const>{ins}); // 0 1 [3 3 3] 2 3 0 1 2 3
    EXPECT_TRUE(js.size() == 11 && js[2] == 3 && js[5] == 2 && js.get_journal().entry_count() == 3);
    js.erase(js.begin() + 4, js.begin() + 7); // 0 1 3 3 | 0 1 2 3
    EXPECT_TRUE(js.size() == 8);
    EXPECT_TRUE((js.materialize() == std::vector<std::uint8_t>{0, 1, 3, 3, 0, 1, 2, 3}));
    js.replace(js.begin(), js.begin() + 1, std::span<std::uint8_t const>{ins}.first(1)); // SNP at 0
    EXPECT_TRUE(js[0] == 3 && js.size() == 8 && js.get_journal().check_journal_invariants());
    js.insert(js.end(), std::span<std::uint8_t const>{ins}); // append
    EXPECT_TRUE(js.size() == 11 && js[10] == 3);
    js.erase(js.begin(), js.end());
    EXPECT_TRUE(js.empty() && js.get_journal().check_journal_invariants());
    // iterator is random access
    spm::journaled_sequence<std::uint8_t> j2{std::span<std::uint8_t const>{src}};
    EXPECT_TRUE(j2.end() - j2.begin() == 8 && *(j2.begin() + 3) == 3 && j2.begin()[5] == 1);
    std::vector<std::uint8_t> copy(j2.begin(), j2.end());
    EXPECT_TRUE(copy == src);
}

struct needle_set
{
    std::vector<std::vector<std::uint8_t>> needles;
    std::vector<std::uint32_t> lens;
    spm::hip::patterns_ptr compiled;
    std::size_t window{};
};

static needle_set make_needles(spm::journaled_sequence_tree const & jst, int algo, std::size_t L, unsigned k, std::size_t count)
{
    needle_set ns;
    std::vector<std::uint8_t> cat;
    std::vector<std::uint32_t> off{0};
    std::vector<std::uint16_t> ks;
    std::uint64_t r = 0x9E3779B97F4A7C15ull;
    for (std::size_t i = 0; i < count; ++i) {
        r = spm_oracle_mix64(r + i);
        std::size_t const h = r % jst.haplotype_count();
        std::vector<std::uint8_t> hap = jst.haplotype(h).materialize();
        std::size_t const at = (r >> 20) % (hap.size() - L - 4);
        std::vector<std::uint8_t> nd(hap.begin() + at, hap.begin() + at + L);
        unsigned const n_edits = k > 8 ? static_cast<unsigned>((i * 9) % (k + 1)) : static_cast<unsigned>(i % (k + 1));
        for (unsigned e = 0; e < n_edits; ++e) // plant up to k substitutions
            nd[spm_oracle_mix64(r + 77 * e) % L] ^= 1;
        cat.insert(cat.end(), nd.begin(), nd.end());
        off.push_back(static_cast<std::uint32_t>(cat.size()));
        ks.push_back(static_cast<std::uint16_t>(k));
        ns.lens.push_back(static_cast<std::uint32_t>(L));
        ns.needles.push_back(std::move(nd));
    }
    spm_patterns * p = nullptr;
    if (spm_hip_patterns_create(spm::hip::default_context(), algo, cat.data(), off.data(), static_cast<std::uint32_t>(count),
                                ks.data(), 4, &p) != SPM_OK)
        spm::hip::fatal("spm_hip_patterns_create", spm::hip::default_context());
    ns.compiled = spm::hip::patterns_ptr{p, spm::hip::patterns_deleter{}};
    ns.window = L + k;
    return ns;
}

static std::vector<spm::jst_hit> linear_scans(spm::journaled_sequence_tree const & jst, needle_set const & ns)
{
    std::vector<spm::jst_hit> out;
    spm_ctx * ctx = spm::hip::default_context();
    for (std::size_t h = 0; h < jst.haplotype_count(); ++h) {
        std::vector<std::uint8_t> const hap = jst.haplotype(h).materialize();
        spm_text * t = nullptr;
        spm_hip_text_upload(ctx, hap.data(), hap.size(), 4, &t);
        spm_hits * hh = nullptr;
        spm_scan_opts opts{};
        if (spm_hip_scan(ctx, t, 0, hap.size(), ns.compiled.get(), &opts, nullptr, nullptr, &hh) != SPM_OK)
            spm::hip::fatal("spm_hip_scan", ctx);
        spm_hit const * rec = nullptr;
        std::uint64_t n = 0;
        spm_hip_hits_view(hh, &rec, &n);
        for (std::uint64_t i = 0; i < n; ++i)
            out.push_back({static_cast<std::uint32_t>(h), rec[i].pos, rec[i].pattern, rec[i].score});
        spm_hip_hits_destroy(hh);
        spm_hip_text_destroy(t);
    }
    std::sort(out.begin(), out.end());
    return out;
}

static bool cpu_only = false;

static void fixture_cases(char const * vcf, char const * haplotypes)
{
    auto ref = spm::io::read_fasta(DATA + "sim_ref_10Kb.fasta.gz");
    EXPECT_TRUE(ref.size() == 1 && ref[0].ranks.size() == 10000);
    auto variants = spm::io::read_vcf(DATA + vcf);
    EXPECT_TRUE(variants.n_haplotypes == 100);
    auto expected = spm::io::read_fasta(DATA + haplotypes);
    EXPECT_TRUE(expected.size() == 100);
    spm::journaled_sequence_tree jst{ref[0].ranks, variants};
    std::size_t equal = 0;
    for (std::size_t h = 0; h < 100; ++h) {
        auto js = jst.haplotype(h);
        EXPECT_TRUE(js.get_journal().check_journal_invariants());
        equal += js.materialize() == expected[h].ranks;
    }
    std::printf("%s: %zu/100 haplotypes reproduce the fixture FASTA\n", vcf, equal);
    EXPECT_TRUE(equal == 100);
    // the context builder works from the allele lists alone (no haplotype is materialised): for every haplotype the
    // owned parts of its contexts, in order, must spell the haplotype, and every context must carry window-1 symbols
    // of left context (or start at the haplotype's first symbol)
    for (std::size_t window : {33u, 103u}) {
        auto X = jst.build_contexts(window, 4 * window);
        std::vector<std::vector<std::pair<std::uint64_t, std::size_t>>> per_h(100);
        for (std::size_t c = 0; c < X.contexts.size(); ++c)
            for (auto const & m : X.contexts[c].members)
                per_h[m.haplotype].emplace_back(m.ctx_lo, c);
        std::size_t good = 0;
        for (std::size_t h = 0; h < 100; ++h) {
            std::sort(per_h[h].begin(), per_h[h].end());
            std::vector<std::uint8_t> spelled;
            bool ok = true;
            for (auto const & [lo, c] : per_h[h]) {
                auto const & cx = X.contexts[c];
                ok = ok && (lo + cx.owned_from == spelled.size()) && (cx.owned_from == window - 1 || lo == 0);
                // the left context must equal what was spelled before
                for (std::size_t i = 0; i < cx.owned_from && ok; ++i)
                    ok = X.buffer[cx.offset + i] == spelled[lo + i];
                spelled.insert(spelled.end(), X.buffer.begin() + cx.offset + cx.owned_from,
                               X.buffer.begin() + cx.offset + cx.length);
            }
            good += ok && spelled == expected[h].ranks;
        }
        EXPECT_TRUE(good == 100);
    }
    if (cpu_only)
        return; // --cpu: journal + ingestion checks only (no device)

    struct cfg
    {
        int algo;
        std::size_t L;
        unsigned k;
        bool begin;
    };
    for (cfg c : {cfg{SPM_ALGO_MYERS, 100, 3, false}, cfg{SPM_ALGO_SHIFTOR, 32, 0, true}, cfg{SPM_ALGO_MYERS, 24, 2, false}}) {
        needle_set ns = make_needles(jst, c.algo, c.L, c.k, 48);
        spm::jst_search_stats st{};
        auto got = jst.search(ns.compiled.get(), ns.window, ns.lens, c.begin, 0, &st);
        auto want = linear_scans(jst, ns);
        std::printf("  algo %d |P|=%zu k=%u: %zu hits; device scanned %llu symbols for %llu haplotype symbols "
                    "(%.1fx less; %llu unique of %llu contexts)\n",
                    c.algo, c.L, c.k, got.size(), (unsigned long long)st.context_symbols,
                    (unsigned long long)st.haplotype_symbols, double(st.haplotype_symbols) / double(st.context_symbols),
                    (unsigned long long)st.unique_contexts, (unsigned long long)st.contexts);
        EXPECT_TRUE(got.size() >= 48);
        EXPECT_TRUE(got == want);
        EXPECT_TRUE(jst.device_ready()); // contexts were cut and deduplicated on the device ...
        spm::jst_search_stats st_host{};
        auto got_host = jst.search_host(ns.compiled.get(), ns.window, ns.lens, c.begin, 0, &st_host);
        EXPECT_TRUE(got_host == want);   // ... and the host-built contexts give the same hits
        EXPECT_TRUE(st_host.haplotype_symbols == st.haplotype_symbols);
        // a different block length must not change anything
        auto got2 = jst.search(ns.compiled.get(), ns.window, ns.lens, c.begin, 777, nullptr);
        EXPECT_TRUE(got2 == want);
        // CPU oracle on three haplotypes x eight needles
        if (c.algo == SPM_ALGO_MYERS) {
            for (std::size_t h : {0u, 37u, 99u}) {
                auto hap = jst.haplotype(h).materialize();
                std::vector<spm::jst_hit> o;
                for (std::uint32_t p = 0; p < 8; ++p) {
                    spm_oracle_myers_state s;
                    spm_oracle_myers_init(&s, c.L, c.k, c.L > 64);
                    std::vector<spm_oracle_hit> buf(hap.size() + 1);
                    std::size_t n = spm_oracle_myers_scan(hap.data(), hap.size(), ns.needles[p].data(), c.L, 4, c.k,
                                                          SPM_ORACLE_INFIX, c.L > 64 ? 2 : 0, &s, 0, buf.data(), buf.size());
                    for (std::size_t i = 0; i < n; ++i)
                        o.push_back({static_cast<std::uint32_t>(h), buf[i].pos, p, buf[i].score});
                }
                std::sort(o.begin(), o.end());
                std::vector<spm::jst_hit> g;
                for (auto const & x : got)
                    if (x.haplotype == h && x.needle < 8)
                        g.push_back(x);
                EXPECT_TRUE(g == o);
            }
        }
    }
}

// C5-shaped synthetic pan-genome at test size: 1 Mbase reference, 64 haplotypes, one SNP per 1 000 bases and one
// indel (1..50) per 10 000 bases, each carried by a pseudo-random subset of the haplotypes.  At this
// (realistic) variant density most contexts are shared, so the device scans far less than haplotypes x length.
static void synthetic_case()
{
    std::size_t const N = 1u << 20, H = 64;
    std::vector<std::uint8_t> ref(N);
    spm_oracle_text(0x5EED0001ull, 0, N, ref.data());
    spm::io::vcf_data v;
    v.n_haplotypes = H;
    std::uint64_t r = 0x5EED0003ull;
    for (std::size_t p = 500; p + 100 < N; p += 1000) {
        r = spm_oracle_mix64(r + p);
        spm::io::vcf_allele a;
        a.coverage.assign(H, 0);
        std::uint64_t cov = spm_oracle_mix64(r ^ 0xABCDEF);
        if ((p / 1000) % 4)
            cov &= spm_oracle_mix64(cov); // most variants are carried by a minority
        if (cov == 0)
            cov = 1;
        for (std::size_t h = 0; h < H; ++h)
            a.coverage[h] = (cov >> h) & 1;
        if ((p / 1000) % 10 == 7) { // indel
            std::size_t const len = 1 + (r >> 8) % 50;
            a.pos = p;
            if ((r >> 20) & 1) {
                a.ref_len = len; // deletion
            } else {
                a.ref_len = 0; // insertion
                for (std::size_t i = 0; i < len; ++i)
                    a.alt.push_back(static_cast<std::uint8_t>((spm_oracle_mix64(r + i) >> 3) & 3));
            }
        } else { // SNP
            a.pos = p;
            a.ref_len = 1;
            a.alt = {static_cast<std::uint8_t>((ref[p] + 1 + (r >> 5) % 3) & 3)};
        }
        v.alleles.push_back(std::move(a));
    }
    spm::journaled_sequence_tree jst{ref, v};
    needle_set ns = make_needles(jst, SPM_ALGO_MYERS, 100, 3, 64);
    spm::jst_search_stats st{};
    auto got = jst.search(ns.compiled.get(), ns.window, ns.lens, false, 0, &st);
    auto want = linear_scans(jst, ns);
    std::printf("synthetic 1 Mbase x 64 haplotypes, %zu variants: %zu hits; device scanned %llu symbols for %llu "
                "haplotype symbols (%.1fx less; %llu unique of %llu contexts)\n",
                v.alleles.size(), got.size(), (unsigned long long)st.context_symbols,
                (unsigned long long)st.haplotype_symbols, double(st.haplotype_symbols) / double(st.context_symbols),
                (unsigned long long)st.unique_contexts, (unsigned long long)st.contexts);
    EXPECT_TRUE(got == want);
    EXPECT_TRUE(got.size() >= 64);
    EXPECT_TRUE(st.context_symbols * 4 < st.haplotype_symbols);

    // C5-shaped needles: |P| = 1024, k <= 64 (seeds of q = 15 symbols, 14-symbol keys; contexts scanned as segmented
    // haystacks).  Needles carry up to 64 planted substitutions.
    needle_set big = make_needles(jst, SPM_ALGO_MYERS, 1024, 64, 8);
    auto got_big = jst.search(big.compiled.get(), big.window, big.lens, false, 0, &st);
    auto want_big = linear_scans(jst, big);
    std::printf("synthetic, 8 needles |P|=1024 k=64: %zu hits; device scanned %llu symbols for %llu (%.1fx less)\n",
                got_big.size(), (unsigned long long)st.context_symbols, (unsigned long long)st.haplotype_symbols,
                double(st.haplotype_symbols) / double(st.context_symbols));
    EXPECT_TRUE(got_big == want_big);
    EXPECT_TRUE(got_big.size() >= 8);
    EXPECT_TRUE(jst.search_host(big.compiled.get(), big.window, big.lens, false, 0, nullptr) == want_big);
}

int main(int argc, char ** argv)
{
    cpu_only = argc > 1 && std::strcmp(argv[1], "--cpu") == 0;
    journal_cases();
    fixture_cases("sim_ref_10Kb_SNPs.vcf", "sim_ref_10Kb_SNPs_haplotypes.fasta.gz");
    fixture_cases("sim_ref_10Kb_SNP_INDELs.vcf", "sim_ref_10Kb_SNP_INDELs_haplotypes.fasta.gz");
    if (!cpu_only)
        synthetic_case();
    std::printf("%d checks, %d failures\n", checks, failures);
    return failures;
}
