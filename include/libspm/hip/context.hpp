// libspm/hip/context.hpp -- RAII glue between the header-only spm:: API and the C ABI of libspm_hip.so.
//
// The reference's call operator is noexcept and has no error channel
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp:41).  There is no CPU scan path in this library, so
// a HIP failure cannot be papered over: it is reported on stderr and the process terminates (fail loudly).
#pragma once

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
} // namespace spm::hip
