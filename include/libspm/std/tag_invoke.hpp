// libspm/std/tag_invoke.hpp -- std::tag_invoke polyfill so that the CPO spelling spm::window_size(m) keeps working.
// Mirrors the facility of /root/reference/libspm/libspm/std/tag_invoke.hpp:13-108 (P1895 tag_invoke); written fresh.
#pragma once

#include <type_traits>
#include <utility>

namespace std
{
namespace _spm_tag_invoke
{
    void tag_invoke(); // poison pill: hides outer names so that only ADL finds overloads

    struct _fn
    {
        template <typename cpo_t, typename... args_t>
        constexpr auto operator()(cpo_t cpo, args_t &&... args) const
            noexcept(noexcept(tag_invoke((cpo_t &&) cpo, (args_t &&) args...)))
                -> decltype(tag_invoke((cpo_t &&) cpo, (args_t &&) args...))
        {
            return tag_invoke((cpo_t &&) cpo, (args_t &&) args...);
        }
    };
} // namespace _spm_tag_invoke

inline namespace _spm_tag_invoke_cpo
{
    inline constexpr _spm_tag_invoke::_fn tag_invoke{};
}

template <auto & cpo>
using tag_t = std::remove_cvref_t<decltype(cpo)>;

template <typename cpo_t, typename... args_t>
concept tag_invocable = requires(cpo_t && cpo, args_t &&... args) {
    std::tag_invoke((cpo_t &&) cpo, (args_t &&) args...);
};

template <typename cpo_t, typename... args_t>
concept nothrow_tag_invocable = tag_invocable<cpo_t, args_t...> && requires(cpo_t && cpo, args_t &&... args) {
    { std::tag_invoke((cpo_t &&) cpo, (args_t &&) args...) } noexcept;
};

template <typename cpo_t, typename... args_t>
inline constexpr bool is_nothrow_tag_invocable_v = nothrow_tag_invocable<cpo_t, args_t...>;

template <typename cpo_t, typename... args_t>
using tag_invoke_result_t = decltype(std::tag_invoke(std::declval<cpo_t>(), std::declval<args_t>()...));

// the trait spellings of the same questions (tag_invoke.hpp:79-102 in the reference)
template <typename cpo_t, typename... args_t>
inline constexpr bool is_tag_invocable_v = tag_invocable<cpo_t, args_t...>;
template <typename cpo_t, typename... args_t>
using is_tag_invocable = std::bool_constant<is_tag_invocable_v<cpo_t, args_t...>>;
template <typename cpo_t, typename... args_t>
using is_nothrow_tag_invocable = std::bool_constant<is_nothrow_tag_invocable_v<cpo_t, args_t...>>;
template <typename cpo_t, typename... args_t>
struct tag_invoke_result // ::type exists iff the call is well-formed
{};
template <typename cpo_t, typename... args_t>
    requires tag_invocable<cpo_t, args_t...>
struct tag_invoke_result<cpo_t, args_t...>
{
    using type = tag_invoke_result_t<cpo_t, args_t...>;
};
} // namespace std
