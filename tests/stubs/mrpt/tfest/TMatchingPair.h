#pragma once  // stand-in
#include <mrpt/math/types.h>
#include <cstdint>
#include <vector>
namespace mrpt::tfest {
struct TMatchingPair { uint32_t globalIdx = 0, localIdx = 0; mrpt::math::TPoint3Df global, local; float errorSquareAfterTransformation = 0; };
using TMatchingPairList = std::vector<TMatchingPair>;
}
