#pragma once  // stand-in
#include <mola_metric_maps/HashedVoxelPointCloud.h>
namespace mola {
class NDT : public mrpt::maps::CMetricMap, public mrpt::maps::NearestNeighborsCapable { public:
  struct VoxelData { const std::vector<mrpt::math::TPoint3Df>& points() const { return p_; } std::vector<mrpt::math::TPoint3Df> p_; };
  struct TInsertionOptions { uint32_t max_points_per_voxel = 0; float min_distance_between_points = 0, remove_voxels_farther_than = 0; double max_eigen_ratio_for_planes = 0.01; } insertionOptions;
  float voxel_size() const { return 1.0f; }
  template <class F> void visitAllVoxels(const F& f) const { f(index3d_t{}, VoxelData{}); }
  template <class F> void visitAllPoints(const F& f) const { f(mrpt::math::TPoint3Df{}); } };
}
