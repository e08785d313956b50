#pragma once  // stand-in
#include <stdexcept>
#define THROW_EXCEPTION(msg) throw std::runtime_error(msg)
