#pragma once  // stand-in
#include <functional>
#include <optional>
namespace mrpt { template <class T> using optional_ref = std::optional<std::reference_wrapper<T>>; }
