// mp2p_icp_granular.cpp -- the Matcher / Solver plugin classes of BASELINE.json's north_star ("keeping the mp2p_icp::ICP /
// Matcher / Solver plugin API"), for a box where mp2p_icp + MRPT + mola_metric_maps ARE installed:
//
//     mp2p_icp::Matcher_Points_DistanceThreshold_HIP : Matcher_Points_DistanceThreshold   implMatchOneLayer -> mh_nn_search
//     mp2p_icp::Matcher_Point2Plane_HIP              : Matcher_Point2Plane                implMatchOneLayer -> mh_nn_search_pt2pl
//     mp2p_icp::Solver_GaussNewton_HIP               : Solver_GaussNewton                 impl_optimal_pose -> mh_gn_solve
//
// They derive from the upstream classes, so the YAML parameters of lidar3d-default.yaml:184-204 / lidar3d-ndt.yaml:184-210
// (threshold formulas, pairingsPerPoint, pointLayerMatches, robustKernel ...) are parsed by upstream's own
// initialize() [U] and every iteration gate (runFromIteration / runUpToIteration, `enabled`) stays in the upstream base
// class: only the O(N) inner loops move to the GPU.  This is the GRANULAR path: the upstream mp2p_icp::ICP loop keeps
// running on the host and every matcher call returns its Pairings to it (an O(N) download per iteration) -- slower than
// ICP_HIP's fused loop, but it takes ANY pipeline built from these classes: two point matchers on different layers
// (extras/lidar3d-dual-map.yaml:121-130), gated matchers (extras/lidar3d-near-far.yaml:183-195), other quality
// evaluators.  `pipelines/make_mola_hip.py --granular` derives the pipeline files that name them.
//
// NOT compiled in this repository's image; upstream signatures are marked [U].  The numeric work behind
// every call is the C ABI that tests/test_gpu_parity.py pins against the oracle (mh_nn_search: test_nn_search_*,
// mh_nn_search_pt2pl: test_pt2pl_*, mh_gn_solve: test_gn_solve_*); the mirror classes of host/src/icp.cpp run the same
// three entry points from the same call shape (tests/test_host_layer.py).
#include <mp2p_icp/Matcher_Point2Plane.h>               // [U]
#include <mp2p_icp/Matcher_Points_DistanceThreshold.h>  // [U]
#include <mp2p_icp/Solver_GaussNewton.h>                // [U]
#include <mrpt/core/initializer.h>
#include <mrpt/rtti/CObject.h>

#include <mutex>

#include "molahip_mrpt_common.h"

namespace mp2p_icp
{
using molahip_mrpt::DeviceSession;
using molahip_mrpt::mh_check;
using molahip_mrpt::pose_to_T12;

/** Drop-in for mp2p_icp::Matcher_Points_DistanceThreshold (lidar3d-default.yaml:196-204). */
class Matcher_Points_DistanceThreshold_HIP : public Matcher_Points_DistanceThreshold
{
    DEFINE_MRPT_OBJECT(Matcher_Points_DistanceThreshold_HIP, mp2p_icp)
   public:
    Matcher_Points_DistanceThreshold_HIP() = default;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, MatchState& ms, const layer_name_t& globalName,
                           const layer_name_t& localName, Pairings& out) const override  // [U]
    {
        const auto& sw = molahip_host::plugin_switches();
        // what the device search implements: up to MH_MAX_PAIRINGS_PER_POINT pairings per point (nn_multiple_search, rgbd.yaml:138),
        // no exclusivity bookkeeping between points, the whole layer (no random subsample), no earlier matcher's pairings to respect
        const bool device_shape = pairingsPerPoint >= 1 && pairingsPerPoint <= MH_MAX_PAIRINGS_PER_POINT &&
                                  allowMatchAlreadyMatchedGlobalPoints_ && maxLocalPointsPerLayer_ == 0 &&
                                  (allowMatchAlreadyMatchedPoints_ || ms.localPairedBitField.point_layers.count(localName) == 0 ||
                                   ms.localPairedBitField.point_layers.at(localName).none());  // [U] members of Matcher_Points_Base / MatchState
        mh_map* dmap = nullptr;
        DeviceSession* dev = nullptr;
        if (!sw.force_cpu && device_shape)
        {
            dev  = &DeviceSession::process_wide();
            dmap = dev->device_map_of(pcGlobal, false);
        }
        if (!dmap) return Matcher_Points_DistanceThreshold::implMatchOneLayer(pcGlobal, pcLocal, localPose, ms, globalName, localName, out);

        std::lock_guard<std::mutex> lk(dev->mutex());  // matchers are const objects that several threads may share [U]
        mh_scan* scan = dev->upload(pcLocal);
        const size_t n = pcLocal.size();
        double T[12];
        pose_to_T12(localPose, T);
        mh_pairs_out po = dev->pairs.out(n * pairingsPerPoint);
        mh_match_info info{};
        mh_check(mh_nn_search_k(dmap, scan, T, threshold, thresholdAngularDeg, (uint32_t)pairingsPerPoint, &po, MH_MEM_HOST, &info),
                 "mh_nn_search_k");
        // Pairings::potential_pairings [U]: every local point of the layer times pairingsPerPoint (U6)
        out.potential_pairings += info.potential_pairings;
        const auto& lx = pcLocal.getPointsBufferRef_x();
        const auto& ly = pcLocal.getPointsBufferRef_y();
        const auto& lz = pcLocal.getPointsBufferRef_z();
        out.paired_pt2pt.reserve(out.paired_pt2pt.size() + info.n_pairs);
        auto& localBits = ms.localPairedBitField.point_layers[localName];  // [U] marks what later matchers must skip
        if (localBits.size() < n) localBits.resize(n);
        for (uint64_t k = 0; k < info.n_pairs; k++)
        {
            mrpt::tfest::TMatchingPair mp;  // [U]
            mp.globalIdx = dev->pairs.gi[k];
            mp.localIdx  = dev->pairs.li[k];
            mp.global    = {dev->pairs.gx[k], dev->pairs.gy[k], dev->pairs.gz[k]};
            mp.local     = {lx[mp.localIdx], ly[mp.localIdx], lz[mp.localIdx]};  // untransformed
            mp.errorSquareAfterTransformation = dev->pairs.d2[k];
            out.paired_pt2pt.push_back(mp);
            localBits.mark_as_set(mp.localIdx);
        }
        // per-layer weight != 1 (pointLayerMatches.weight, yaml:204): recorded like upstream does [U]
        const double w = weight_pt2pt_layers.at(globalName).at(localName);
        if (w != 1.0 && info.n_pairs) out.point_weights.emplace_back(info.n_pairs, w);
    }
};
IMPLEMENTS_MRPT_OBJECT(Matcher_Points_DistanceThreshold_HIP, mp2p_icp::Matcher_Points_DistanceThreshold, mp2p_icp)

/** Drop-in for mp2p_icp::Matcher_Point2Plane on a mola::NDT global layer (lidar3d-ndt.yaml:195-200, 236-254) and, since round 5,
 *  on a plain mola::HashedVoxelPointCloud layer (KNN + PCA, rgbd.yaml:143-151; knn <= MH_MAX_PLANE_KNN). */
class Matcher_Point2Plane_HIP : public Matcher_Point2Plane
{
    DEFINE_MRPT_OBJECT(Matcher_Point2Plane_HIP, mp2p_icp)
   public:
    Matcher_Point2Plane_HIP() = default;

   private:
    void implMatchOneLayer(const mrpt::maps::CMetricMap& pcGlobal, const mrpt::maps::CPointsMap& pcLocal,
                           const mrpt::poses::CPose3D& localPose, MatchState& ms, const layer_name_t& globalName,
                           const layer_name_t& localName, Pairings& out) const override  // [U]
    {
        const auto& sw = molahip_host::plugin_switches();
        const bool device_shape = maxLocalPointsPerLayer_ == 0 &&
                                  (allowMatchAlreadyMatchedPoints_ || ms.localPairedBitField.point_layers.count(localName) == 0 ||
                                   ms.localPairedBitField.point_layers.at(localName).none());
        mh_map* dmap = nullptr;
        bool knn_path = false;
        DeviceSession* dev = nullptr;
        if (!sw.force_cpu && device_shape)
        {
            dev  = &DeviceSession::process_wide();
            dmap = dev->device_map_of(pcGlobal, true);  // an NDT map: per-voxel planes
            if (!dmap && knn >= 3 && knn <= MH_MAX_PLANE_KNN)
            {   // a plain point layer (mola::HashedVoxelPointCloud, rgbd.yaml:143-151): k nearest neighbours + PCA on the device
                dmap     = dev->device_map_of(pcGlobal, false);
                knn_path = dmap != nullptr;
            }
        }
        if (!dmap) return Matcher_Point2Plane::implMatchOneLayer(pcGlobal, pcLocal, localPose, ms, globalName, localName, out);

        std::lock_guard<std::mutex> lk(dev->mutex());
        mh_scan* scan = dev->upload(pcLocal);
        const size_t n = pcLocal.size();
        double T[12];
        pose_to_T12(localPose, T);
        mh_pairs_pl_out po = dev->planes.out(n);
        mh_match_info info{};
        if (knn_path)
        {
            mh_pt2pl_knn_params kp{};  // [U] the upstream members DECLARE_PARAMETER'd from the yaml keys of rgbd.yaml:145-149
            kp.distance_threshold    = distanceThreshold;
            kp.plane_eigen_threshold = planeEigenThreshold;
            kp.search_radius         = searchRadius;
            kp.knn                   = knn;
            kp.minimum_plane_points  = minimumPlanePoints;
            mh_check(mh_nn_search_pt2pl_knn(dmap, scan, T, &kp, &po, MH_MEM_HOST, &info), "mh_nn_search_pt2pl_knn");
        }
        else
            mh_check(mh_nn_search_pt2pl(dmap, scan, T, distanceThreshold, sw.pt2pl_mode, &po, MH_MEM_HOST, &info), "mh_nn_search_pt2pl");
        out.potential_pairings += info.potential_pairings;
        const auto& lx = pcLocal.getPointsBufferRef_x();
        const auto& ly = pcLocal.getPointsBufferRef_y();
        const auto& lz = pcLocal.getPointsBufferRef_z();
        out.paired_pt2pl.reserve(out.paired_pt2pl.size() + info.n_pairs);
        auto& localBits = ms.localPairedBitField.point_layers[localName];
        if (localBits.size() < n) localBits.resize(n);
        for (uint64_t k = 0; k < info.n_pairs; k++)
        {
            const uint32_t i = dev->planes.li[k];
            point_plane_pair_t pp;  // [U] {pl_global{plane, centroid}, pt_local}
            pp.pl_global.centroid = {dev->planes.cx[k], dev->planes.cy[k], dev->planes.cz[k]};
            pp.pl_global.plane    = mrpt::math::TPlane(mrpt::math::TPoint3D(dev->planes.cx[k], dev->planes.cy[k], dev->planes.cz[k]),
                                                       mrpt::math::TVector3D(dev->planes.nx[k], dev->planes.ny[k], dev->planes.nz[k]));
            pp.pt_local           = {lx[i], ly[i], lz[i]};
            out.paired_pt2pl.push_back(pp);
            localBits.mark_as_set(i);
        }
    }
};
IMPLEMENTS_MRPT_OBJECT(Matcher_Point2Plane_HIP, mp2p_icp::Matcher_Point2Plane, mp2p_icp)

/** Drop-in for mp2p_icp::Solver_GaussNewton (lidar3d-default.yaml:184-190): point-to-point and point-to-plane terms, the
 *  prior factor, robust kernel, inner iterations.  Pairings with other geometric entities (lines, plane-to-plane) go to
 *  the upstream solver. */
class Solver_GaussNewton_HIP : public Solver_GaussNewton
{
    DEFINE_MRPT_OBJECT(Solver_GaussNewton_HIP, mp2p_icp)
   public:
    Solver_GaussNewton_HIP() = default;

   protected:
    bool impl_optimal_pose(const Pairings& pairings, OptimalTF_Result& out, const SolverContext& sc) const override  // [U]
    {
        const auto& sw = molahip_host::plugin_switches();
        const bool only_points_and_planes = pairings.paired_pt2ln.empty() && pairings.paired_ln2ln.empty() && pairings.paired_pl2pl.empty();  // [U]
        // Pairings::point_weights [U] = {(pairs, weight)} runs over paired_pt2pt: ONE run (every point pair from layers of the same
        // weight) is a device input; several runs are not
        if (sw.force_cpu || !only_points_and_planes || pairings.point_weights.size() > 1 || !sc.guessRelativePose.has_value())
            return Solver_GaussNewton::impl_optimal_pose(pairings, out, sc);

        DeviceSession& dev = DeviceSession::process_wide();
        std::lock_guard<std::mutex> lk(dev.mutex());
        const size_t np = pairings.paired_pt2pt.size(), nl = pairings.paired_pt2pl.size();
        float* a[6];
        float* b[9];
        buf_.resize(6 * np + 9 * nl);
        for (int c = 0; c < 6; c++) a[c] = buf_.data() + c * np;
        for (int c = 0; c < 9; c++) b[c] = buf_.data() + 6 * np + c * nl;
        for (size_t k = 0; k < np; k++)
        {
            const auto& p = pairings.paired_pt2pt[k];  // [U] mrpt::tfest::TMatchingPair {global, local}
            a[0][k] = p.local.x, a[1][k] = p.local.y, a[2][k] = p.local.z;
            a[3][k] = p.global.x, a[4][k] = p.global.y, a[5][k] = p.global.z;
        }
        for (size_t k = 0; k < nl; k++)
        {
            const auto& p = pairings.paired_pt2pl[k];
            const auto  nrm = p.pl_global.plane.getNormalVector();  // [U] unit normal of the TPlane
            b[0][k] = p.pt_local.x, b[1][k] = p.pt_local.y, b[2][k] = p.pt_local.z;
            b[3][k] = p.pl_global.centroid.x, b[4][k] = p.pl_global.centroid.y, b[5][k] = p.pl_global.centroid.z;
            b[6][k] = static_cast<float>(nrm.x), b[7][k] = static_cast<float>(nrm.y), b[8][k] = static_cast<float>(nrm.z);
        }
        mh_pairs_pt2pt pp{a[0], a[1], a[2], a[3], a[4], a[5], np};
        mh_pairs_pt2pl pl{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], nl};
        mh_gn_params gp{};
        gp.max_inner_iterations = maxIterations;  // yaml:187
        const std::string kname = mrpt::typemeta::TEnumType<RobustKernel>::value2name(robustKernel);  // [U] by NAME, not by value
        gp.robust_kernel        = molahip_host::kernel_from_upstream_name(kname.c_str(), sw);
        gp.robust_kernel_param  = robustKernelParam;  // the formula of yaml:190, realised for this ICP_ITERATION by the ParameterSource [U]
        gp.min_delta            = sw.min_delta;
        gp.max_cost             = sw.max_cost;
        // the solver's own per-kind weights (yaml `pairWeights` [U]: OptimalTF_GN_Parameters::pairWeights upstream; 1.0 in every shipped
        // pipeline) are a device input of mh_gn_solve (they were hard-coded to 1)
        gp.weight_pt2pt = pairWeights.pt2pt * (pairings.point_weights.empty() ? 1.0 : pairings.point_weights[0].second);  // [U] member names
        gp.weight_pt2pl = pairWeights.pt2pl;  // [U]
        mh_prior pr;
        if (sc.prior.has_value())  // [U] SolverContext::prior (the motion model's, LidarOdometry.cpp:859-861)
        {
            pose_to_T12(sc.prior->mean, pr.mean);
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) pr.info[i * 6 + j] = sc.prior->cov_inv(i, j);
        }
        double T[12];
        pose_to_T12(*sc.guessRelativePose, T);  // linearisation point [U]
        int32_t n_steps = 0, ok = 0;
        mh_check(mh_gn_solve(dev.ctx(), np ? &pp : nullptr, nl ? &pl : nullptr, MH_MEM_HOST, &gp, sc.prior.has_value() ? &pr : nullptr, T, &n_steps, &ok,
                             nullptr), "mh_gn_solve");
        if (!ok) return false;  // -> IterTermReason::SolverError in the caller [U]
        mrpt::math::CMatrixDouble44 M = mrpt::math::CMatrixDouble44::Identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) M(i, j) = T[i * 4 + j];
        out.optimalPose  = mrpt::poses::CPose3D(M);
        out.optimalScale = 1.0;
        return true;
    }

   private:
    mutable std::vector<float> buf_;  // SoA staging of the pairings (guarded by the session's mutex)
};
IMPLEMENTS_MRPT_OBJECT(Solver_GaussNewton_HIP, mp2p_icp::Solver_GaussNewton, mp2p_icp)

}  // namespace mp2p_icp

// same registration pattern as module/src/register.cpp:40-46
MRPT_INITIALIZER(do_register_molahip_mp2p_icp_granular)
{
    mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::Matcher_Points_DistanceThreshold_HIP));
    mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::Matcher_Point2Plane_HIP));
    mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::Solver_GaussNewton_HIP));
}
