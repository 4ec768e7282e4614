// libspm/matcher/concept.hpp -- customisation points and concepts of the matcher API.
//
// Public surface (names, call syntax, constraints) as in /root/reference/libspm/libspm/matcher/concept.hpp:26-161:
//   spm::window_size(m)          size of the text window a hit depends on         (:27-53)
//   spm::capture(m)              the matcher's state                              (:56-78)
//   spm::restore(m, state)       continue from a captured state                   (:81-102)
//   spm::aggregate(s1, s2)       reduce two states (declared, no default)         (:108-120)
//   window_matcher / restorable_matcher / reducable_with / reducable_state / online_matcher_for   (:126-161)
// Every point dispatches through std::tag_invoke and falls back to the member function of the same name.  Here all
// four are instances of ONE small template (member_cpo) instead of four hand-written function objects.
#pragma once

#include <concepts>
#include <type_traits>
#include <utility>

#include <libspm/std/tag_invoke.hpp>

namespace spm
{
namespace detail
{
    // A customisation point object: tag_invoke(cpo, args...) if some overload exists, else the member function
    // selected by `member_t` (a stateless callable that spells obj.member(rest...)).
    template <typename member_t>
    struct member_cpo
    {
        template <typename... args_t>
            requires std::tag_invocable<member_cpo, args_t...>
        constexpr decltype(auto) operator()(args_t &&... args) const
            noexcept(std::is_nothrow_tag_invocable_v<member_cpo, args_t...>)
        {
            return std::tag_invoke(*this, std::forward<args_t>(args)...);
        }

        template <typename... args_t>
            requires(!std::tag_invocable<member_cpo, args_t...>) && std::invocable<member_t, args_t...>
        constexpr decltype(auto) operator()(args_t &&... args) const
            noexcept(std::is_nothrow_invocable_v<member_t, args_t...>)
        {
            return member_t{}(std::forward<args_t>(args)...);
        }
    };

    struct call_window_size
    {
        template <typename m_t>
            requires requires(m_t && m) { { std::forward<m_t>(m).window_size() } -> std::integral; }
        constexpr auto operator()(m_t && m) const noexcept(noexcept(std::forward<m_t>(m).window_size()))
        {
            return std::forward<m_t>(m).window_size();
        }
    };
    struct call_capture
    {
        template <typename m_t>
            requires requires(m_t const & m) { m.capture(); }
        constexpr decltype(auto) operator()(m_t const & m) const noexcept(noexcept(m.capture()))
        {
            return m.capture();
        }
    };
    struct call_restore
    {
        template <typename m_t, typename s_t>
            requires requires(m_t & m, s_t && s) { m.restore(std::forward<s_t>(s)); }
        constexpr void operator()(m_t & m, s_t && s) const noexcept(noexcept(m.restore(std::forward<s_t>(s))))
        {
            m.restore(std::forward<s_t>(s));
        }
    };
    struct no_member // aggregate has no member fallback: it exists only where someone tag_invokes it
    {};
} // namespace detail

inline constexpr detail::member_cpo<detail::call_window_size> window_size{};
inline constexpr detail::member_cpo<detail::call_capture> capture{};
inline constexpr detail::member_cpo<detail::call_restore> restore{};
inline constexpr detail::member_cpo<detail::no_member> aggregate{};

template <typename matcher_t>
using window_size_t = std::invoke_result_t<decltype(window_size), matcher_t>;

template <typename matcher_t>
using matcher_state_t = std::remove_cvref_t<std::invoke_result_t<decltype(capture), matcher_t const &>>;

// a matcher that can be copied freely and tells how much text a hit depends on
template <typename matcher_t>
concept window_matcher = std::copyable<std::remove_cvref_t<matcher_t>> &&
                         std::integral<std::remove_cvref_t<window_size_t<matcher_t>>>;

// ... whose progress can be captured into a regular value and resumed from it
template <typename matcher_t>
concept restorable_matcher =
    window_matcher<matcher_t> && std::semiregular<matcher_state_t<std::remove_cvref_t<matcher_t>>> &&
    std::invocable<decltype(restore), std::remove_cvref_t<matcher_t> &, matcher_state_t<std::remove_cvref_t<matcher_t>>>;

template <typename t1, typename t2>
concept reducable_with =
    std::common_with<std::remove_cvref_t<t1>, std::remove_cvref_t<t2>> &&
    requires(std::remove_cvref_t<t1> const & a, std::remove_cvref_t<t2> const & b) {
        { spm::aggregate(a, b) } -> std::convertible_to<std::common_type_t<std::remove_cvref_t<t1>, std::remove_cvref_t<t2>>>;
    };

template <typename state_t>
concept reducable_state = reducable_with<state_t, state_t>;

template <typename matcher_t, typename... args_t>
concept online_matcher_for = window_matcher<matcher_t> && std::invocable<matcher_t, args_t...>;
} // namespace spm
