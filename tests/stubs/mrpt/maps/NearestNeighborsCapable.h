#pragma once  // stand-in
#include <mrpt/math/types.h>
#include <cstdint>
#include <vector>
namespace mrpt::maps {
class NearestNeighborsCapable { public: virtual ~NearestNeighborsCapable() = default;
  virtual bool nn_has_indices_or_ids() const = 0; virtual size_t nn_index_count() const = 0;
  virtual bool nn_single_search(const mrpt::math::TPoint3Df&, mrpt::math::TPoint3Df&, float&, uint64_t&) const = 0;
  virtual bool nn_single_search(const mrpt::math::TPoint2Df&, mrpt::math::TPoint2Df&, float&, uint64_t&) const = 0;
  virtual void nn_multiple_search(const mrpt::math::TPoint3Df&, size_t, std::vector<mrpt::math::TPoint3Df>&, std::vector<float>&, std::vector<uint64_t>&) const = 0;
  virtual void nn_radius_search(const mrpt::math::TPoint3Df&, float, std::vector<mrpt::math::TPoint3Df>&, std::vector<float>&, std::vector<uint64_t>&, size_t) const = 0; };
}
