#pragma once  // stand-in
#include <string_view>
namespace mrpt::system {
class CTimeLogger { public: bool isEnabled() const { return true; } void registerUserMeasure(const std::string_view&, double, bool = false) {} };
struct CTimeLoggerEntry { CTimeLoggerEntry(const CTimeLogger&, const std::string_view&) {} };
}
