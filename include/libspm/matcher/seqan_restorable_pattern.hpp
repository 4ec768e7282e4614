// libspm/matcher/seqan_restorable_pattern.hpp -- tag kept for source compatibility with
// /root/reference/libspm/libspm/matcher/seqan_restorable_pattern.hpp:17-23 (spm::Restorable<tag_t> selects the
// restorable SeqAn pattern specialisations there; here it only names the variant).
#pragma once

namespace spm
{
template <typename tag_t>
struct Restorable : public tag_t
{};
} // namespace spm
