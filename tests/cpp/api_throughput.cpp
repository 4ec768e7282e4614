// api_throughput.cpp -- what a caller of the C++ mirror gets: spm::batch_myers_matcher over a haystack that is resident in
// HBM (spm::hip::resident_haystack), callbacks included, against the C-ABI number bench.py reports for the same shape
// (1 024 needles |P| = 100, k <= 3; 2 GiB of synthetic dna4 text, needles cut from it with up to k edits).
//   usage: api_throughput [GiB = 2] [repetitions = 5]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <libspm/matcher/hip_batch.hpp>
#include <libspm/seqan/alphabet.hpp>

int main(int argc, char ** argv)
{
    double const gib = argc > 1 ? std::atof(argv[1]) : 2.0;
    int const reps = argc > 2 ? std::atoi(argv[2]) : 5;
    std::uint64_t const n = static_cast<std::uint64_t>(gib * (1ull << 30)) & ~1023ull;
    std::uint64_t const seed_text = 0x5EED0001, seed_pat = 0x5EED0002;
    std::uint32_t const n_needles = 1024, L = 100, k = 3;

    spm_ctx * ctx = spm::hip::default_context();
    spm_text * text = nullptr;
    if (spm_hip_text_generate(ctx, seed_text, 0, n, &text) != SPM_OK)
        spm::hip::fatal("spm_hip_text_generate", ctx);
    // the text lives in HBM already: borrowed, not copied
    auto const hs = spm::hip::resident_haystack::wrap(spm_hip_text_device_ptr(text), n, 4);

    std::vector<std::vector<spm::dna4>> needles(n_needles);
    for (std::uint32_t p = 0; p < n_needles; ++p) {
        std::vector<std::uint8_t> ranks(L);
        spm_hip_synth_pattern(seed_text, seed_pat, n, p, L, k, ranks.data());
        needles[p].resize(L);
        for (std::uint32_t i = 0; i < L; ++i)
            needles[p][i].assign_rank(ranks[i]);
    }
    auto const t_build = std::chrono::steady_clock::now();
    auto matcher = spm::batch_myers_matcher{needles, k};
    double const ms_build = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();

    std::uint64_t hits = 0, found_mask_count = 0;
    std::vector<char> found(n_needles, 0);
    double best = 1e30;
    for (int r = 0; r < reps + 1; ++r) { // (the first repetition is the warm-up)
        hits = 0;
        auto const t0 = std::chrono::steady_clock::now();
        matcher(hs, [&](std::size_t needle, auto const & finder) {
            ++hits;
            found[needle] = 1;
            (void)seqan2::endPosition(finder);
        });
        double const ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (r > 0)
            best = std::min(best, ms);
    }
    for (char f : found)
        found_mask_count += f;
    std::printf("{\"api\": \"spm::batch_myers_matcher over spm::hip::resident_haystack\", \"needles\": %u, \"text_bytes\": %llu, "
                "\"construct_ms\": %.2f, \"ms_per_call\": %.3f, \"Gbases_per_s\": %.1f, \"hits\": %llu, \"needles_found\": %llu}\n",
                n_needles, static_cast<unsigned long long>(n), ms_build, best, static_cast<double>(n) / best / 1e6,
                static_cast<unsigned long long>(hits), static_cast<unsigned long long>(found_mask_count));
    spm_hip_text_destroy(text);
    return found_mask_count == n_needles ? 0 : 1;
}
