// hashed_voxel_pointcloud_hip.cpp -- see the header.  NOT compiled here; [U] = verify against the installed MRPT / MOLA.
#include "hashed_voxel_pointcloud_hip.h"

#include "molahip_host/plugin_switches.h"

#include <mrpt/core/initializer.h>
#include <mrpt/obs/CObservationPointCloud.h>  // [U]
#include <mrpt/opengl/CPointCloud.h>          // [U]

#include <iostream>
#include <stdexcept>

namespace mola
{
namespace
{
inline void mh_check(mh_status s, const char* where)
{
    if (s != MH_OK) throw std::runtime_error(std::string(where) + ": " + mh_status_string(s) + ": " + mh_last_error_string());
}
}  // namespace

IMPLEMENTS_SERIALIZABLE(HashedVoxelPointCloudHIP, CMetricMap, mola)  // [U]

HashedVoxelPointCloudHIP::HashedVoxelPointCloudHIP(float voxel_size) : voxel_size_(voxel_size) {}
HashedVoxelPointCloudHIP::~HashedVoxelPointCloudHIP()
{
    if (staging_) mh_scan_destroy(staging_);
    if (map_) mh_map_destroy(map_);
    if (ctx_) mh_ctx_destroy(ctx_);
}

void HashedVoxelPointCloudHIP::ensure_device() const
{
    if (map_) return;
    mh_check(mh_ctx_create(molahip_host::device_index(), nullptr, &ctx_), "mh_ctx_create");  // MOLA_HIP_DEVICE, like ICP_HIP
    mh_map_params p{};
    p.voxel_size                  = voxel_size_;                                   // creationOpts.voxel_size (yaml:233)
    p.max_points_per_voxel        = insertionOptions.max_points_per_voxel;        // yaml:235
    p.index_mode                  = molahip_host::plugin_switches().index_mode;       // MOLA_HIP_INDEX_MODE (default floor)
    p.far_voxel_metric            = molahip_host::plugin_switches().far_voxel_metric; // MOLA_HIP_FAR_VOXEL_METRIC (yaml:237-238)
    p.min_distance_between_points = insertionOptions.min_distance_between_points; // yaml:236
    mh_check(mh_map_create(ctx_, &p, &map_), "mh_map_create");
    mh_check(mh_scan_create(ctx_, nullptr, nullptr, nullptr, 0, MH_MEM_HOST, &staging_), "mh_scan_create");
}

bool HashedVoxelPointCloudHIP::isEmpty() const
{
    if (!map_) return true;
    mh_map_info i;
    mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    return i.n_points == 0;
}
void HashedVoxelPointCloudHIP::internal_clear()
{
    if (map_) mh_check(mh_map_build(map_, nullptr, nullptr, nullptr, 0, MH_MEM_HOST), "mh_map_build");
}

void HashedVoxelPointCloudHIP::insertPointCloud(const mrpt::maps::CPointsMap& pc, const mrpt::poses::CPose3D& pc_in_map)
{
    ensure_device();
    const auto& xs = pc.getPointsBufferRef_x();  // SoA already [U]
    const auto& ys = pc.getPointsBufferRef_y();
    const auto& zs = pc.getPointsBufferRef_z();
    mh_check(mh_scan_update(staging_, xs.data(), ys.data(), zs.data(), xs.size(), MH_MEM_HOST), "mh_scan_update");
    double T[12];
    const auto& R = pc_in_map.getRotationMatrix();
    for (int r = 0; r < 3; r++)
    {
        for (int c = 0; c < 3; c++) T[r * 4 + c] = R(r, c);
        T[r * 4 + 3] = pc_in_map.m_coords[r];
    }
    // insertPoint for every point after the stored content, per-voxel cap, then far-voxel removal (yaml:238)
    const mh_status st = mh_map_insert(map_, staging_, T, insertionOptions.remove_voxels_farther_than);
    if (st == MH_WARN_PREVIOUS_OUT_OF_RANGE)  // this cloud IS in the map; the previous one lost its out-of-range points
        std::cerr << "[HashedVoxelPointCloudHIP] warning: " << mh_last_error_string() << std::endl;
    else
        mh_check(st, "mh_map_insert");
}

bool HashedVoxelPointCloudHIP::internal_insertObservation(const mrpt::obs::CObservation& obs,
                                                          const std::optional<const mrpt::poses::CPose3D>& robotPose)
{
    const auto* o = dynamic_cast<const mrpt::obs::CObservationPointCloud*>(&obs);  // [U] other classes: via a CSimplePointsMap
    if (!o || !o->pointcloud) return false;
    mrpt::poses::CPose3D p = robotPose ? *robotPose : mrpt::poses::CPose3D::Identity();
    insertPointCloud(*o->pointcloud, p + o->sensorPose);
    return true;
}

mrpt::math::TBoundingBoxf HashedVoxelPointCloudHIP::boundingBox() const
{
    mrpt::math::TBoundingBoxf bb;
    if (!map_) return bb;
    mh_map_info i;
    mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    bb.min = {i.bbox_min[0], i.bbox_min[1], i.bbox_min[2]};
    bb.max = {i.bbox_max[0], i.bbox_max[1], i.bbox_max[2]};
    return bb;
}

size_t HashedVoxelPointCloudHIP::nn_index_count() const
{
    if (!map_) return 0;
    mh_map_info i;
    mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    return i.n_offered;  // indices are source indices: positions in everything ever offered to the map
}

bool HashedVoxelPointCloudHIP::nn_single_search(const mrpt::math::TPoint3Df& q, mrpt::math::TPoint3Df& result, float& out_dist_sqr,
                                                uint64_t& resultIndexOrID) const
{
    if (!map_) return false;
    mh_check(mh_scan_update(staging_, &q.x, &q.y, &q.z, 1, MH_MEM_HOST), "mh_scan_update");
    const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    uint32_t gi = 0;
    float    gx, gy, gz, d2;
    mh_check(mh_nn_search_dense(map_, staging_, I, &gi, &gx, &gy, &gz, &d2, MH_MEM_HOST), "mh_nn_search_dense");
    if (gi == 0xFFFFFFFFu) return false;
    result          = {gx, gy, gz};
    out_dist_sqr    = d2;
    resultIndexOrID = gi;
    return true;
}
void HashedVoxelPointCloudHIP::nn_multiple_search(const mrpt::math::TPoint3Df& q, size_t N, std::vector<mrpt::math::TPoint3Df>& results,
                                                  std::vector<float>& out_dists_sqr, std::vector<uint64_t>& resultIndicesOrIDs) const
{
    // one query through the k-best search (mh_nn_search_k with an infinite threshold): the N nearest of the 3x3x3 block in
    // ascending (distance, scan position).  A per-point call is a device round trip: the matchers batch a whole layer instead.
    results.clear();
    out_dists_sqr.clear();
    resultIndicesOrIDs.clear();
    if (!map_ || N == 0) return;
    if (N > MH_MAX_PAIRINGS_PER_POINT) THROW_EXCEPTION("nn_multiple_search: more neighbours than MH_MAX_PAIRINGS_PER_POINT");
    mh_check(mh_scan_update(staging_, &q.x, &q.y, &q.z, 1, MH_MEM_HOST), "mh_scan_update");
    const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    uint32_t     li[MH_MAX_PAIRINGS_PER_POINT], gi[MH_MAX_PAIRINGS_PER_POINT];
    float        gx[MH_MAX_PAIRINGS_PER_POINT], gy[MH_MAX_PAIRINGS_PER_POINT], gz[MH_MAX_PAIRINGS_PER_POINT], d2[MH_MAX_PAIRINGS_PER_POINT];
    mh_pairs_out po{li, gi, gx, gy, gz, d2};
    mh_match_info info{};
    if (N == 1)
    {
        mrpt::math::TPoint3Df r;
        float                 d;
        uint64_t              id;
        if (!nn_single_search(q, r, d, id)) return;
        results.push_back(r);
        out_dists_sqr.push_back(d);
        resultIndicesOrIDs.push_back(id);
        return;
    }
    mh_check(mh_nn_search_k(map_, staging_, I, 1.0e18 /* no threshold: fp32 d^2 stays below its square */, 0.0, (uint32_t)N, &po,
                            MH_MEM_HOST, &info),
             "mh_nn_search_k");
    for (uint64_t k = 0; k < info.n_pairs; k++)
    {
        results.emplace_back(gx[k], gy[k], gz[k]);
        out_dists_sqr.push_back(d2[k]);
        resultIndicesOrIDs.push_back(gi[k]);
    }
}
void HashedVoxelPointCloudHIP::nn_radius_search(const mrpt::math::TPoint3Df&, float, std::vector<mrpt::math::TPoint3Df>&,
                                                std::vector<float>&, std::vector<uint64_t>&, size_t) const
{
    THROW_EXCEPTION("nn_radius_search: not provided by the device map");
}

std::string HashedVoxelPointCloudHIP::asString() const { return "HashedVoxelPointCloudHIP (device resident, libmolahip)"; }

void HashedVoxelPointCloudHIP::getVisualizationInto(mrpt::opengl::CSetOfObjects& o) const
{
    if (!map_) return;
    mh_map_info i;
    mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    std::vector<float> x(i.n_points), y(i.n_points), z(i.n_points);
    mh_check(mh_map_download(map_, x.data(), y.data(), z.data(), nullptr, nullptr, nullptr, nullptr), "mh_map_download");
    auto pc = mrpt::opengl::CPointCloud::Create();
    pc->setAllPoints(x, y, z);  // [U]
    o.insert(pc);
}
void HashedVoxelPointCloudHIP::saveMetricMapRepresentationToFile(const std::string& prefix) const
{
    if (!map_) return;
    mh_map_info i;
    mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    std::vector<float> x(i.n_points), y(i.n_points), z(i.n_points);
    mh_check(mh_map_download(map_, x.data(), y.data(), z.data(), nullptr, nullptr, nullptr, nullptr), "mh_map_download");
    FILE* f = fopen((prefix + "_points.txt").c_str(), "wt");
    if (!f) return;
    for (size_t k = 0; k < x.size(); k++) fprintf(f, "%f %f %f\n", x[k], y[k], z[k]);
    fclose(f);
}

// serialization: the stored points and the options are all the state there is [U]
uint8_t HashedVoxelPointCloudHIP::serializeGetVersion() const { return 0; }
void    HashedVoxelPointCloudHIP::serializeTo(mrpt::serialization::CArchive& out) const
{
    mh_map_info i{};
    if (map_) mh_check(mh_map_get_info(map_, &i), "mh_map_get_info");
    std::vector<float> x(i.n_points), y(i.n_points), z(i.n_points);
    if (map_) mh_check(mh_map_download(map_, x.data(), y.data(), z.data(), nullptr, nullptr, nullptr, nullptr), "mh_map_download");
    out << voxel_size_ << insertionOptions.max_points_per_voxel << insertionOptions.min_distance_between_points
        << insertionOptions.remove_voxels_farther_than << x << y << z;
}
void HashedVoxelPointCloudHIP::serializeFrom(mrpt::serialization::CArchive& in, uint8_t)
{
    std::vector<float> x, y, z;
    in >> voxel_size_ >> insertionOptions.max_points_per_voxel >> insertionOptions.min_distance_between_points >>
        insertionOptions.remove_voxels_farther_than >> x >> y >> z;
    ensure_device();
    mh_check(mh_map_build(map_, x.data(), y.data(), z.data(), x.size(), MH_MEM_HOST), "mh_map_build");
}
}  // namespace mola

// registered like the reference's own classes (module/src/register.cpp:40-46); the pipeline file then says
//   class: mola::HashedVoxelPointCloudHIP      plugin: 'libmolahip_mp2p_icp.so'
MRPT_INITIALIZER(do_register_molahip_metric_map) { mrpt::rtti::registerClass(CLASS_ID(mola::HashedVoxelPointCloudHIP)); }
