// hashed_voxel_pointcloud_hip.h -- a device-owned local map for a real MOLA stack.
//
// NOT compiled in this repository's image (MRPT / mola_metric_maps are absent): written against the upstream API as
// recalled, every such line is marked [U] -- re-check against the installed headers.
//
// mola::HashedVoxelPointCloudHIP is selected in the pipeline file instead of mola::HashedVoxelPointCloud
// (lidar3d-default.yaml:228-242, `class:` + `plugin:`): same creationOpts / insertOpts, but the points live in an
// mh_map on the GPU.  insertObservation / insertAnotherMap (what FilterMerge calls, yaml:362-368) forward to
// mh_map_insert, so the key-frame update needs no host copy of the map, and ICP_HIP (mp2p_icp_plugin.cpp) takes the
// handle directly instead of mirroring a host map.  The NearestNeighborsCapable methods are implemented for
// completeness (single queries through mh_nn_search_dense: slow, meant for tools, not for the ICP loop).
#pragma once
#include <mrpt/maps/CMetricMap.h>                // [U]
#include <mrpt/maps/NearestNeighborsCapable.h>   // [U]
#include <mrpt/maps/CPointsMap.h>                // [U]

#include "molahip.h"

namespace mola
{
class HashedVoxelPointCloudHIP : public mrpt::maps::CMetricMap, public mrpt::maps::NearestNeighborsCapable
{
    DEFINE_SERIALIZABLE(HashedVoxelPointCloudHIP, mola)  // [U]
   public:
    HashedVoxelPointCloudHIP(float voxel_size = 1.0f);
    ~HashedVoxelPointCloudHIP() override;

    struct TInsertionOptions  // insertOpts of yaml:234-238
    {
        uint32_t max_points_per_voxel        = 20;
        float    min_distance_between_points = 0;
        float    remove_voxels_farther_than  = 0;
    } insertionOptions;

    mh_map* deviceHandle() const { return map_; }  // what ICP_HIP::align() consumes
    mh_ctx* deviceContext() const { return ctx_; }

    // ---- CMetricMap [U]
    bool isEmpty() const override;
    void internal_clear() override;
    bool internal_insertObservation(const mrpt::obs::CObservation& obs,
                                    const std::optional<const mrpt::poses::CPose3D>& robotPose) override;
    double internal_computeObservationLikelihood(const mrpt::obs::CObservation&, const mrpt::poses::CPose3D&) const override { return 0; }
    std::string asString() const override;
    void getVisualizationInto(mrpt::opengl::CSetOfObjects& o) const override;
    void saveMetricMapRepresentationToFile(const std::string& prefix) const override;
    mrpt::math::TBoundingBoxf boundingBox() const override;
    /** FilterMerge path: points of `pc` (frame of `pc_in_map`) into the map, then far-voxel removal */
    void insertPointCloud(const mrpt::maps::CPointsMap& pc, const mrpt::poses::CPose3D& pc_in_map);

    // ---- NearestNeighborsCapable [U]
    bool   nn_has_indices_or_ids() const override { return true; }
    size_t nn_index_count() const override;
    bool   nn_single_search(const mrpt::math::TPoint3Df& q, mrpt::math::TPoint3Df& result, float& out_dist_sqr,
                            uint64_t& resultIndexOrID) const override;
    bool   nn_single_search(const mrpt::math::TPoint2Df&, mrpt::math::TPoint2Df&, float&, uint64_t&) const override { return false; }
    void   nn_multiple_search(const mrpt::math::TPoint3Df&, size_t, std::vector<mrpt::math::TPoint3Df>&, std::vector<float>&,
                              std::vector<uint64_t>&) const override;
    void   nn_radius_search(const mrpt::math::TPoint3Df&, float, std::vector<mrpt::math::TPoint3Df>&, std::vector<float>&,
                            std::vector<uint64_t>&, size_t) const override;

   private:
    void ensure_device() const;
    float            voxel_size_;
    mutable mh_ctx*  ctx_     = nullptr;
    mutable mh_map*  map_     = nullptr;
    mutable mh_scan* staging_ = nullptr;  // the layer being inserted / the single query point
};
}  // namespace mola
