#pragma once  // stand-in
#include <mrpt/maps/CMetricMap.h>
#include <mrpt/maps/NearestNeighborsCapable.h>
#include <vector>
namespace mola {
struct index3d_t { int32_t cx = 0, cy = 0, cz = 0; };
class HashedVoxelPointCloud : public mrpt::maps::CMetricMap, public mrpt::maps::NearestNeighborsCapable { public:
  struct VoxelData { const std::vector<mrpt::math::TPoint3Df>& points() const { return p_; } std::vector<mrpt::math::TPoint3Df> p_; };
  struct TInsertionOptions { uint32_t max_points_per_voxel = 0; float min_distance_between_points = 0, remove_voxels_farther_than = 0; } insertionOptions;
  float voxel_size() const { return 1.0f; }
  template <class F> void visitAllVoxels(const F& f) const { f(index3d_t{}, VoxelData{}); }
  template <class F> void visitAllPoints(const F& f) const { f(mrpt::math::TPoint3Df{}); } };
}
