// libspm/matcher/seqan_pattern_base.hpp -- include-path compatibility with the reference
// (/root/reference/libspm/libspm/matcher/seqan_pattern_base.hpp); the driver lives in hip_pattern_base.hpp.
#pragma once

#include <libspm/matcher/hip_pattern_base.hpp>

namespace spm
{
template <typename derived_t>
using seqan_pattern_base = hip_pattern_base<derived_t>;
}
