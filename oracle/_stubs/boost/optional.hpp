// TEST INFRASTRUCTURE (oracle/Makefile `ref`): Boost is not in this image; boost::optional as the std one.
#pragma once
#include <optional>
namespace boost { template <class T> using optional = std::optional<T>; inline constexpr std::nullopt_t none = std::nullopt; template <class T> std::optional<std::decay_t<T>> make_optional(T&& v) { return std::optional<std::decay_t<T>>(std::forward<T>(v)); } }
