// libspm/matcher/concept.hpp -- matcher CPOs and concepts.
// Same contract as /root/reference/libspm/libspm/matcher/concept.hpp:26-161:
//   spm::window_size(m)  member .window_size() or tag_invoke              (:27-53)
//   spm::capture(m) / spm::restore(m, s) / spm::aggregate(s1, s2)         (:56-120; aggregate is declared only)
//   window_matcher (std::copyable + integral window), restorable_matcher, online_matcher_for  (:126-161)
#pragma once

#include <concepts>
#include <type_traits>

#include <libspm/std/tag_invoke.hpp>

namespace spm
{
namespace _window_size
{
    inline constexpr struct _cpo
    {
        template <typename matcher_t>
            requires std::tag_invocable<_cpo, matcher_t>
        constexpr auto operator()(matcher_t && m) const noexcept(std::is_nothrow_tag_invocable_v<_cpo, matcher_t>)
            -> std::tag_invoke_result_t<_cpo, matcher_t>
        {
            return std::tag_invoke(_cpo{}, (matcher_t &&) m);
        }

    private:
        template <typename matcher_t>
            requires requires(matcher_t && m) { { ((matcher_t &&) m).window_size() } -> std::integral; }
        constexpr friend auto tag_invoke(_cpo, matcher_t && m) noexcept(noexcept(((matcher_t &&) m).window_size()))
        {
            return ((matcher_t &&) m).window_size();
        }
    } window_size;
} // namespace _window_size
using _window_size::window_size;

template <typename matcher_t>
using window_size_t = std::invoke_result_t<_window_size::_cpo, matcher_t>;

namespace _capture
{
    inline constexpr struct _cpo
    {
        template <typename matcher_t>
            requires std::tag_invocable<_cpo, matcher_t const &>
        constexpr auto operator()(matcher_t const & m) const
            noexcept(std::is_nothrow_tag_invocable_v<_cpo, matcher_t const &>)
                -> std::tag_invoke_result_t<_cpo, matcher_t const &>
        {
            return std::tag_invoke(_cpo{}, m);
        }

    private:
        template <typename matcher_t>
            requires requires(matcher_t && m) { ((matcher_t &&) m).capture(); }
        constexpr friend auto tag_invoke(_cpo, matcher_t && m) noexcept(noexcept(((matcher_t &&) m).capture()))
            -> decltype(((matcher_t &&) m).capture())
        {
            return ((matcher_t &&) m).capture();
        }
    } capture;
} // namespace _capture
using _capture::capture;

namespace _restore
{
    inline constexpr struct _cpo
    {
        template <typename matcher_t, typename state_t>
            requires std::tag_invocable<_cpo, matcher_t &, state_t>
        constexpr auto operator()(matcher_t & m, state_t && s) const
            noexcept(std::is_nothrow_tag_invocable_v<_cpo, matcher_t &, state_t>)
                -> std::tag_invoke_result_t<_cpo, matcher_t &, state_t>
        {
            return std::tag_invoke(_cpo{}, m, (state_t &&) s);
        }

    private:
        template <typename matcher_t, typename state_t>
            requires requires(matcher_t && m, state_t && s) { ((matcher_t &&) m).restore((state_t &&) s); }
        constexpr friend void tag_invoke(_cpo, matcher_t && m, state_t && s)
            noexcept(noexcept(((matcher_t &&) m).restore((state_t &&) s)))
        {
            ((matcher_t &&) m).restore((state_t &&) s);
        }
    } restore;
} // namespace _restore
using _restore::restore;

template <typename matcher_t>
using matcher_state_t = std::remove_cvref_t<std::invoke_result_t<_capture::_cpo, matcher_t>>;

namespace _aggregate
{
    inline constexpr struct _cpo
    {
        template <typename s1_t, typename s2_t>
            requires std::tag_invocable<_cpo, s1_t, s2_t>
        constexpr auto operator()(s1_t && a, s2_t && b) const noexcept(std::is_nothrow_tag_invocable_v<_cpo, s1_t, s2_t>)
            -> std::tag_invoke_result_t<_cpo, s1_t, s2_t>
        {
            return std::tag_invoke(_cpo{}, (s1_t &&) a, (s2_t &&) b);
        }
    } aggregate;
} // namespace _aggregate
using _aggregate::aggregate;

template <typename matcher_t>
concept window_matcher = std::copyable<std::remove_cvref_t<matcher_t>> && requires(matcher_t && m) {
    { spm::window_size((matcher_t &&) m) } -> std::integral;
};

namespace detail
{
    template <typename matcher_t>
    concept stateful_matcher = requires {
        typename spm::matcher_state_t<matcher_t>;
        requires std::semiregular<spm::matcher_state_t<matcher_t>>;
    };
} // namespace detail

template <typename matcher_t>
concept restorable_matcher = window_matcher<matcher_t> && detail::stateful_matcher<matcher_t> &&
                             requires(matcher_t && m) {
    { spm::capture((matcher_t &&) m) } -> std::convertible_to<spm::matcher_state_t<matcher_t>>;
    { spm::restore(m, std::declval<spm::matcher_state_t<matcher_t>>()) };
};

template <typename t1, typename t2>
concept reducable_with = std::common_with<std::remove_cvref_t<t1>, std::remove_cvref_t<t2>> &&
                         requires(std::remove_cvref_t<t1> const & a, std::remove_cvref_t<t2> const & b) {
    { spm::aggregate(a, b) } -> std::convertible_to<std::common_type_t<std::remove_cvref_t<t1>, std::remove_cvref_t<t2>>>;
};

template <typename state_t>
concept reducable_state = reducable_with<state_t, state_t>;

template <typename matcher_t, typename... args_t>
concept online_matcher_for = window_matcher<matcher_t> && std::invocable<matcher_t, args_t...>;
} // namespace spm
