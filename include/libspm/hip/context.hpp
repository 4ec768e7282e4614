// libspm/hip/context.hpp -- RAII glue between the header-only spm:: API and the C ABI of libspm_hip.so.
//
// The reference's call operator is noexcept and has no error channel
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:41).  There is no CPU scan path in this library, so
// a HIP failure cannot be papered over: it is reported on stderr and the process terminates (fail loudly).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include <spm_hip.h>

namespace spm::hip
{
[[noreturn]] inline void fatal(char const * where, spm_ctx const * ctx) noexcept
{
    std::fprintf(stderr, "libspm (MI355X back-end): %s failed: %s\n", where, spm_hip_last_error(ctx));
    std::abort();
}

// One context (device + stream) per process by default; SPM_HIP_DEVICE selects the GPU.
inline spm_ctx * default_context() noexcept
{
    static spm_ctx * ctx = [] {
        spm_ctx * c = nullptr;
        char const * dev = std::getenv("SPM_HIP_DEVICE");
        if (spm_hip_init(dev ? std::atoi(dev) : 0, nullptr, &c) != SPM_OK)
            fatal("spm_hip_init", nullptr);
        return c;
    }();
    return ctx;
}

struct patterns_deleter
{
    void operator()(spm_patterns * p) const noexcept { spm_hip_patterns_destroy(p); }
};
struct text_deleter
{
    void operator()(spm_text * p) const noexcept { spm_hip_text_destroy(p); }
};
struct hits_deleter
{
    void operator()(spm_hits * p) const noexcept { spm_hip_hits_destroy(p); }
};
using patterns_ptr = std::shared_ptr<spm_patterns>; // shared between copies of a matcher (std::copyable)
using text_ptr = std::unique_ptr<spm_text, text_deleter>;
using hits_ptr = std::unique_ptr<spm_hits, hits_deleter>;

// A scan whose hits are read on the host.  The reference's find loop has no hit limit
// (seqan_pattern_base.hpp:49-51), the device hit buffer has one (spm_scan_opts.max_hits): when it was too small the views
// return SPM_E_OVERFLOW with the count so far in the scan's statistics, and the scan is repeated with room for it.
// Only real failures are fatal.  `run(opts, &hits)` performs the scan (spm_hip_scan / spm_hip_scan_segments).
template <typename run_t>
inline hits_ptr scan_all_hits(spm_ctx * ctx, spm_scan_opts opts, run_t && run, spm_hit const *& rec, std::uint64_t & n,
                              char const * what) noexcept
{
    for (int attempt = 0;; ++attempt) {
        spm_hits * h = nullptr;
        if (run(opts, &h) != SPM_OK)
            fatal(what, ctx);
        hits_ptr hits{h};
        int const rc = spm_hip_hits_view(h, &rec, &n);
        if (rc == SPM_OK)
            return hits;
        if (rc != SPM_E_OVERFLOW || attempt >= 12)
            fatal("spm_hip_hits_view", ctx);
        spm_scan_stats st{};
        spm_hip_hits_stats(h, &st);
        std::uint64_t const cap = opts.max_hits ? opts.max_hits : (1ull << 20);
        opts.max_hits = std::max<std::uint64_t>(2 * cap, st.n_hits + st.n_hits / 8 + 1024);
    }
}
} // namespace spm::hip
