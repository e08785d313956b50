#pragma once  // stand-in
#include <mrpt/maps/CMetricMap.h>
#include <mrpt/maps/NearestNeighborsCapable.h>
#include <vector>
namespace mrpt::maps {
class CPointsMap : public CMetricMap { public:
  using Ptr = std::shared_ptr<CPointsMap>;
  size_t size() const { return x_.size(); }
  const std::vector<float>& getPointsBufferRef_x() const { return x_; }
  const std::vector<float>& getPointsBufferRef_y() const { return y_; }
  const std::vector<float>& getPointsBufferRef_z() const { return z_; }
 protected: std::vector<float> x_, y_, z_; };
}
