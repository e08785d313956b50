// mp2p_icp_plugin.cpp -- the adapter a MOLA maintainer builds where mp2p_icp + MRPT ARE installed.
//
// NOT compiled in this repository's image (mp2p_icp, mrpt-*, mola_* are absent: SURVEY.md 0.2); the upstream
// signatures below are written from the upstream API as recalled in SURVEY.md 8(b) and are marked [U]: re-check
// them against the installed headers.  It derives from the REAL mp2p_icp classes so that YAML parameter parsing
// (DECLARE_PARAMETER_*), the iteration hook, the profiler and the ParameterSource attachment keep working, and
// forwards the numeric work to the C ABI of libmolahip (include/molahip.h).  Registration uses the same RTTI
// mechanism as the reference's own module (module/src/register.cpp:40-46), so that
//
//   mola-lidar-odometry-cli -l libmolahip_mp2p_icp.so -c pipelines/lidar3d-default-hip.yaml ...
//
// (apps/mola-lidar-odometry-cli.cpp:93-95,553-562) resolves "class_name: mp2p_icp::ICP_HIP".
//
// Build: see CMakeLists.txt next to this file.
#include <mp2p_icp/ICP.h>                               // [U]
#include <mp2p_icp/Matcher_Points_DistanceThreshold.h>  // [U]
#include <mp2p_icp/Solver_GaussNewton.h>                // [U]
#include <mrpt/core/initializer.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/rtti/CObject.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

#include "hashed_voxel_pointcloud_hip.h"
#include "molahip.h"
#include "molahip_host/hook_replay.h"  // the opaque iteration hook on the fused loop (compiled + tested via host/src/icp.cpp)

namespace mp2p_icp
{
namespace
{
inline void mh_check(mh_status s, const char* where)
{
    // the reference catches std::exception around the whole scan (LidarOdometry.cpp:614-619)
    if (s != MH_OK) throw std::runtime_error(std::string(where) + ": " + mh_status_string(s) + ": " + mh_last_error_string());
}
inline void pose_to_T12(const mrpt::poses::CPose3D& p, double T[12])
{
    const auto& R = p.getRotationMatrix();
    for (int i = 0; i < 3; i++)
    {
        for (int j = 0; j < 3; j++) T[i * 4 + j] = R(i, j);
        T[i * 4 + 3] = p.m_coords[i];
    }
}

/** Device mirror of one host map layer.  The plugin does not own the host map (no change notification), so the
 *  mirror is rebuilt when its CONTENT changes: point count, bounding box and a fingerprint of the coordinates (every
 *  point when there are few, a strided sample of 4096 otherwise: a key-frame insertion that replaces points without
 *  changing their number or their box still moves the fingerprint) -- SURVEY.md 7.3 "map mirror coherence".  The
 *  proper fix is the device-owned CMetricMap class next to this file (hashed_voxel_pointcloud_hip.h, row f2). */
struct MapMirror
{
    mh_map* map   = nullptr;
    size_t  nPts  = 0;
    mrpt::math::TBoundingBoxf bbox;
    uint64_t fingerprint = 0;
    float voxel_size = 0;
    uint32_t max_points_per_voxel = 0;
};

inline uint64_t fingerprint_of(const mrpt::maps::CPointsMap& pm)
{
    const auto& x = pm.getPointsBufferRef_x();
    const auto& y = pm.getPointsBufferRef_y();
    const auto& z = pm.getPointsBufferRef_z();
    const size_t n = x.size(), step = n > 4096 ? n / 4096 : 1;
    uint64_t h = 1469598103934665603ull;  // FNV-1a over the raw coordinate bits
    auto mix = [&h](float v) { uint32_t b; memcpy(&b, &v, 4); h = (h ^ b) * 1099511628211ull; };
    for (size_t i = 0; i < n; i += step) { mix(x[i]); mix(y[i]); mix(z[i]); }
    if (n) { mix(x[n - 1]); mix(y[n - 1]); mix(z[n - 1]); }
    return h;
}

/** creationOpts.voxel_size / insertOpts.max_points_per_voxel of the host map (lidar3d-default.yaml:233,235; the yaml
 *  evaluates voxel_size to 0.5-1.0 m).  [U]: mola::HashedVoxelPointCloud keeps the voxel size private behind
 *  setVoxelProperties() and exposes insertionOptions; adapt the two accessors below to the installed header.  The
 *  environment overrides exist for the day the accessors are wrong. */
inline void voxel_params_of(const mrpt::maps::CMetricMap& g, float& voxel_size, uint32_t& max_points_per_voxel)
{
    voxel_size = 1.0f;
    max_points_per_voxel = 20;
    if (const auto* hv = dynamic_cast<const mola::HashedVoxelPointCloud*>(&g))
    {
        voxel_size           = hv->voxel_size();                            // [U]
        max_points_per_voxel = hv->insertionOptions.max_points_per_voxel;   // [U]
    }
    if (const char* e = getenv("MOLAHIP_VOXEL_SIZE")) voxel_size = static_cast<float>(atof(e));
    if (const char* e = getenv("MOLAHIP_MAX_POINTS_PER_VOXEL")) max_points_per_voxel = static_cast<uint32_t>(atoi(e));
}
}  // namespace

/** Drop-in for mp2p_icp::ICP: same align() signature as the call at LidarOdometry.cpp:961-962. */
class ICP_HIP : public ICP
{
    DEFINE_MRPT_OBJECT(ICP_HIP, mp2p_icp)
   public:
    ICP_HIP() { mh_check(mh_ctx_create(0, nullptr, &ctx_), "mh_ctx_create"); }
    ~ICP_HIP() override
    {
        for (auto& kv : mirrors_) mh_map_destroy(kv.second.map);
        if (scan_) mh_scan_destroy(scan_);
        mh_ctx_destroy(ctx_);
    }

    void align(
        const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const mrpt::math::TPose3D& initialGuessLocalWrtGlobal,
        const Parameters& p, Results& result, const std::optional<mrpt::poses::CPose3DPDFGaussianInf>& prior = std::nullopt,
        const mrpt::optional_ref<LogRecord>& outputDebugInfo = std::nullopt) override  // [U]
    {
        // Fused path only for the pipeline shape of lidar3d-default.yaml:162-209; anything else -> upstream CPU code.
        // generateDebugFiles too: the upstream loop then fills the LogRecord and applies
        // Parameters::functor_before_logging_local (set at LidarOdometry.cpp:360-364) itself before it writes the .icplog.
        auto* m = matchers().size() == 1 ? dynamic_cast<Matcher_Points_DistanceThreshold*>(matchers()[0].get()) : nullptr;
        auto* s = solvers().size() == 1 ? dynamic_cast<Solver_GaussNewton*>(solvers()[0].get()) : nullptr;
        if (!m || !s || m->pairingsPerPoint != 1 || m->weight_pt2pt_layers.size() != 1 /*[U]*/ || p.generateDebugFiles)
            return ICP::align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);

        mrpt::system::CTimeLoggerEntry tle(profiler(), "align_hip");  // keeps profiler() populated (LidarOdometry.cpp:351-352)
        const auto& [globalName, localMap] = *m->weight_pt2pt_layers.begin();  // [U] {global -> {local -> weight}}
        const auto& localName              = localMap.begin()->first;
        const auto* local  = dynamic_cast<const mrpt::maps::CPointsMap*>(pcLocal.layers.at(localName).get());
        const auto& global = pcGlobal.layers.at(globalName);
        ASSERT_(local);

        // thresholds: functions of ICP_ITERATION (lidar3d-default.yaml:190,198): evaluate per iteration up front
        std::vector<double> thr(p.maxIterations), kp(p.maxIterations);
        for (uint32_t k = 0; k < p.maxIterations; k++)
        {
            for (auto* src : attachedSources()) { src->updateVariable("ICP_ITERATION", k); src->realize(); }  // [U]
            thr[k] = m->threshold;
            kp[k]  = s->robustKernelParam;
        }

        mh_map* dmap = mirror_of(*global);
        const auto& lx = local->getPointsBufferRef_x();  // already SoA
        const auto& ly = local->getPointsBufferRef_y();
        const auto& lz = local->getPointsBufferRef_z();
        if (!scan_) mh_check(mh_scan_create(ctx_, lx.data(), ly.data(), lz.data(), lx.size(), MH_MEM_HOST, &scan_), "mh_scan_create");
        else        mh_check(mh_scan_update(scan_, lx.data(), ly.data(), lz.data(), lx.size(), MH_MEM_HOST), "mh_scan_update");

        mh_icp_params ip{};
        ip.max_iterations        = p.maxIterations;
        ip.min_abs_step_trans    = p.minAbsStep_trans;
        ip.min_abs_step_rot      = p.minAbsStep_rot;
        ip.threshold             = thr.data();
        ip.kernel_param          = kp.data();
        ip.threshold_angular_deg = m->thresholdAngularDeg;
        ip.gn.max_inner_iterations = s->maxIterations;
        ip.gn.robust_kernel      = static_cast<uint32_t>(s->robustKernel);  // map the enum explicitly once verified [U]
        ip.gn.min_delta          = 1e-7;
        ip.gn.weight_pt2pt = ip.gn.weight_pt2pl = 1.0;
        ip.compute_covariance    = 1;
        ip.cov_findif_xyz = ip.cov_findif_ang = 1e-7;
        ip.poll_every = 0;

        double T0[12];
        pose_to_T12(mrpt::poses::CPose3D(initialGuessLocalWrtGlobal), T0);
        mh_prior pr;
        if (prior)
        {
            pose_to_T12(prior->mean, pr.mean);
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) pr.info[i * 6 + j] = prior->cov_inv(i, j);
        }
        mh_icp_result r{};
        std::vector<uint32_t> li(lx.size()), gi(lx.size());
        std::vector<float> gx(lx.size()), gy(lx.size()), gz(lx.size()), d2(lx.size());
        mh_pairs_out po{li.data(), gi.data(), gx.data(), gy.data(), gz.data(), d2.data()};
        auto run = [&](uint32_t budget, mh_icp_iter* trace) {
            mh_icp_params q = ip;
            q.max_iterations = budget;
            mh_icp_result rr{};
            mh_check(mh_icp_align(dmap, scan_, &q, T0, prior ? &pr : nullptr, &rr, trace, &po, MH_MEM_HOST), "mh_icp_align");
            return rr;
        };
        if (iteration_hook_)  // [U] ICP::iteration_hook_: what setIterationHook() stored (LidarOdometry.cpp:923)
        {
            // The hook LidarOdometry installs (LidarOdometry.cpp:923-952) is opaque here: replay it on the traced poses
            // and reproduce a requested stop with a second run of that budget (molahip_host/hook_replay.h).  Without
            // this the twist re-estimation loop of :958-1007 would silently never trigger.
            auto hook = [&](uint32_t k, const double* T) {
                mrpt::math::CMatrixDouble44 Mk = mrpt::math::CMatrixDouble44::Identity();
                for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) Mk(i, j) = T[i * 4 + j];
                OptimalTF_Result cur;                                 // [U]
                cur.optimalPose = mrpt::poses::CPose3D(Mk);
                IterationHook_Input in;                               // [U] fields as used at LidarOdometry.cpp:932-936
                in.currentIteration = k;
                in.currentSolution  = &cur;
                in.pcGlobal = &pcGlobal;
                in.pcLocal  = &pcLocal;
                return iteration_hook_(in).request_stop;              // [U] IterationHook_Output::request_stop (:949)
            };
            r = molahip_host::align_with_replayed_hook(p.maxIterations, run, hook);
        }
        else
            r = run(p.maxIterations, nullptr);

        // results back into the upstream structures (Results::finalPairings is read by LidarOdometry and the log writer)
        mrpt::math::CMatrixDouble44 M = mrpt::math::CMatrixDouble44::Identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) M(i, j) = r.T[i * 4 + j];
        result.optimal_tf.mean = mrpt::poses::CPose3D(M);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) result.optimal_tf.cov(i, j) = r.cov[i * 6 + j];
        result.quality           = r.quality;
        result.nIterations       = r.n_iterations;
        result.terminationReason = static_cast<IterTermReason>(r.termination_reason);  // same order as MH_TERM_* [U]
        result.finalPairings     = Pairings();
        result.finalPairings.potential_pairings = r.potential_pairings;
        for (uint32_t k = 0; k < r.n_final_pairs; k++)
        {
            mrpt::tfest::TMatchingPair mp;
            mp.globalIdx = gi[k];
            mp.localIdx  = li[k];
            mp.global    = {gx[k], gy[k], gz[k]};
            mp.local     = {lx[li[k]], ly[li[k]], lz[li[k]]};
            mp.errorSquareAfterTransformation = d2[k];
            result.finalPairings.paired_pt2pt.push_back(mp);
        }
    }

   private:
    mh_map* mirror_of(const mrpt::maps::CMetricMap& g)
    {
        // a device-owned local map (hashed_voxel_pointcloud_hip.h): nothing to mirror, the handle is the map.
        // (Its context must be the one this ICP runs on: both use device 0's default stream here.)
        if (const auto* dm = dynamic_cast<const mola::HashedVoxelPointCloudHIP*>(&g)) return dm->deviceHandle();
        const auto* pm = dynamic_cast<const mrpt::maps::CPointsMap*>(&g);  // HashedVoxelPointCloud exposes its points through
        ASSERT_(pm);                                                        // a visitor [U]; adapt here once verified
        auto& mir = mirrors_[&g];
        float vs; uint32_t cap;
        voxel_params_of(g, vs, cap);  // what the map object says, not constants (yaml:233 evaluates to 0.5-1.0 m)
        if (mir.map && (mir.voxel_size != vs || mir.max_points_per_voxel != cap))
        {
            mh_map_destroy(mir.map);
            mir = MapMirror();
        }
        if (!mir.map)
        {
            mh_map_params mp{};
            mp.voxel_size = vs;
            mp.max_points_per_voxel = cap;
            mp.index_mode = MH_INDEX_FLOOR;
            mh_check(mh_map_create(ctx_, &mp, &mir.map), "mh_map_create");
            mir.voxel_size = vs;
            mir.max_points_per_voxel = cap;
            mir.nPts = ~size_t(0);  // force the first build
        }
        const auto bb = g.boundingBox();
        const uint64_t fp = fingerprint_of(*pm);
        if (mir.nPts != pm->size() || !(bb == mir.bbox) || fp != mir.fingerprint)
        {
            mh_check(mh_map_build(mir.map, pm->getPointsBufferRef_x().data(), pm->getPointsBufferRef_y().data(),
                                  pm->getPointsBufferRef_z().data(), pm->size(), MH_MEM_HOST), "mh_map_build");
            mir.nPts = pm->size();
            mir.bbox = bb;
            mir.fingerprint = fp;
        }
        return mir.map;
    }

    mh_ctx*  ctx_  = nullptr;
    mh_scan* scan_ = nullptr;
    std::unordered_map<const mrpt::maps::CMetricMap*, MapMirror> mirrors_;
};
IMPLEMENTS_MRPT_OBJECT(ICP_HIP, mp2p_icp::ICP, mp2p_icp)

}  // namespace mp2p_icp

// same registration pattern as module/src/register.cpp:40-46
MRPT_INITIALIZER(do_register_molahip_mp2p_icp) { mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::ICP_HIP)); }
