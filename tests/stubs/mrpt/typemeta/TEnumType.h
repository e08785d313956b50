#pragma once  // stand-in
#include <string>
namespace mrpt::typemeta { template <class E> struct TEnumType { static std::string value2name(E) { return ""; } }; }
