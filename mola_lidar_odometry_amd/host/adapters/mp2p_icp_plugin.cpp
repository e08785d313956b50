// mp2p_icp_plugin.cpp -- the adapter a MOLA maintainer builds where mp2p_icp + MRPT + mola_metric_maps ARE installed.
//
// NOT compiled in this repository's image (mp2p_icp, mrpt-*, mola_* are absent: SURVEY.md 0.2); the upstream
// signatures below are written from the upstream API as recalled in SURVEY.md 8(b) and are marked [U]: re-check
// them against the installed headers.  It derives from the REAL mp2p_icp::ICP so that YAML parameter parsing
// (DECLARE_PARAMETER_*), the iteration hook, the profiler and the ParameterSource attachment keep working, reads the
// parsed parameters of the UPSTREAM matcher / solver objects the pipeline file names, and forwards the numeric work to the
// C ABI of libmolahip (include/molahip.h).  Registration uses the same RTTI mechanism as the reference's own module
// (module/src/register.cpp:40-46).  The names this library adds to MRPT's class factory for the ICP side are
//
//     mp2p_icp::ICP_HIP                                   (this file: the fused device loop)
//     mp2p_icp::Matcher_Points_DistanceThreshold_HIP      (mp2p_icp_granular.cpp: one matcher call = one device search)
//     mp2p_icp::Matcher_Point2Plane_HIP
//     mp2p_icp::Solver_GaussNewton_HIP                    (one solver call = one device accumulate + solve)
//
// and the pipeline files that select ICP_HIP are the reference's own files with that one class_name changed
// (pipelines/make_mola_hip.py -> pipelines/generated/lidar3d-{default,ndt}-mola-hip.yaml; tests/test_mola_hip_pipelines.py
// checks statically that every class name in them is either upstream's or registered here, and that every key
// LidarOdometry.cpp:246-483 requires is present):
//
//   mola-lidar-odometry-cli -l libmolahip_mp2p_icp.so -c pipelines/generated/lidar3d-default-mola-hip.yaml ...
//
// (apps/mola-lidar-odometry-cli.cpp:93-95,553-562).
//
// Pipeline shapes taken by the fused device loop (anything else is delegated to the upstream CPU ICP::align):
//   lidar3d-default.yaml:184-204   one Solver_GaussNewton, matchers = [Matcher_Points_DistanceThreshold]
//   lidar3d-ndt.yaml:184-210       one Solver_GaussNewton, matchers = [Matcher_Point2Plane, Matcher_Points_DistanceThreshold]
// both with one {global, local, weight} entry in pointLayerMatches (yaml :203-204; the point matcher's weight may differ from 1
// since round 5), pairingsPerPoint 1.
// Global layers read: mola::HashedVoxelPointCloud (yaml:230), mola::NDT (ndt yaml:236) -- through their point / voxel
// visitors, they are NOT mrpt::maps::CPointsMap --, mola::HashedVoxelPointCloudHIP (device owned, no mirror), and any
// CPointsMap (pipelines/extras/localmap_definition_pointmap.ini).
//
// Unverified upstream behaviours are environment switches here, read once per process, so that
// tools/parity_pin.py can sweep them against the reference's own run:
//   MOLA_HIP_ROBUST_KERNEL   GemanMcClure (c^4/(c^2+e^2)^2, default) | GemanMcClure_KISS | GemanMcClure_Barron |
//                            GemanMcClure_C2 | Cauchy     -- what `RobustKernel::GemanMcClure` (yaml:188) means   (U1)
//   MOLA_HIP_INDEX_MODE      floor (default) | trunc      -- coordinate -> voxel index of the MIRROR's own table (U2/U3)
//   MOLA_HIP_COV_STEP_XYZ / MOLA_HIP_COV_STEP_ANG   finite-difference steps of mp2p_icp::covariance, 1e-7        (U7)
//   MOLA_HIP_MIN_DELTA / MOLA_HIP_MAX_COST          Gauss-Newton early exits, 1e-7 / 0                           (U8)
//   MOLA_HIP_PT2PL_MODE      plane (default: |n.(p-c)| < distanceThreshold) | centroid (|p-c| < distanceThreshold) (U10)
//   MOLA_HIP_MATCHED_POINTS  again (default: both matchers of the NDT pipeline pair every point) | skip (points paired by
//                            Matcher_Point2Plane are left out of the point matcher: allowMatchAlreadyMatchedPoints = false) (U12)
//   MOLA_HIP_FORCE_CPU=1     every call goes to the upstream loop (sanity A/A through the same plugin)
//   MOLA_HIP_DEVICE=n        the GPU this process uses (default 0): eval/cli_kitti.sh:23-36 runs one process per sequence,
//                            `parallel -j8 MOLA_HIP_DEVICE='{= $_ = slot() - 1 =}' ...` spreads them over a node's GPUs
//   MOLA_HIP_ALIGN_TRACE=f   one CSV row per align() -- which loop ran, nIterations, terminationReason, quality, pairing
//                            counts, pose -- from BOTH loops (with MOLA_HIP_FORCE_CPU=1 it records the reference's own
//                            numbers): what tools/parity_pin.py diffs per scan besides the TUM poses
//
// Build: see CMakeLists.txt next to this file.
#include <mola_metric_maps/HashedVoxelPointCloud.h>     // [U] mola::HashedVoxelPointCloud
#include <mola_metric_maps/NDT.h>                       // [U] mola::NDT
#include <mp2p_icp/ICP.h>                               // [U]
#include <mp2p_icp/Matcher_Point2Plane.h>               // [U]
#include <mp2p_icp/Matcher_Points_DistanceThreshold.h>  // [U]
#include <mp2p_icp/Solver_GaussNewton.h>                // [U]
#include <mrpt/core/initializer.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/rtti/CObject.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "molahip_mrpt_common.h"        // device session, map mirror, MOLA_HIP_* switches
#include "molahip_host/hook_replay.h"  // the opaque iteration hook on the fused loop (compiled + tested via host/src/icp.cpp)

namespace mp2p_icp
{
namespace
{
using molahip_mrpt::mh_check;
using molahip_mrpt::pose_to_T12;
/** MOLA_HIP_ALIGN_TRACE: one row per align(), same columns whichever loop ran. */
struct AlignTrace
{
    FILE*  f = nullptr;
    size_t call = 0;
    AlignTrace()
    {
        if (const char* e = getenv("MOLA_HIP_ALIGN_TRACE")) f = fopen(e, "w");
        if (f) fprintf(f, "call,loop,n_local,nIterations,terminationReason,quality,n_pt2pt,n_pt2pl,x,y,z,yaw,pitch,roll\n");
    }
    ~AlignTrace() { if (f) fclose(f); }
    void row(const char* loop, size_t n_local, const Results& r)
    {
        if (!f) return;
        const auto& m = r.optimal_tf.mean;
        fprintf(f, "%zu,%s,%zu,%u,%d,%.17g,%zu,%zu,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", call++, loop, n_local,
                (unsigned)r.nIterations, (int)r.terminationReason, r.quality, r.finalPairings.paired_pt2pt.size(),
                r.finalPairings.paired_pt2pl.size(), m.x(), m.y(), m.z(), m.yaw(), m.pitch(), m.roll());
        fflush(f);
    }
};

/** The pipeline shapes the fused loop takes (see the file header). */
struct Shape
{
    const Matcher_Points_DistanceThreshold* pt = nullptr;
    const Matcher_Point2Plane*              pl = nullptr;  // nullptr: lidar3d-default shape
    const Solver_GaussNewton*               gn = nullptr;
    std::string globalLayer, localLayer;
    double ptLayerWeight = 1.0;  // pointLayerMatches {..., weight} of the point matcher (yaml:203-204)
};

template <class M> bool single_unit_layer(const M& m, std::string& g, std::string& l, double* weight_out = nullptr)
{
    if (m.weight_pt2pt_layers.size() != 1) return false;                     // [U] {global -> {local -> weight}}
    const auto& [gname, locals] = *m.weight_pt2pt_layers.begin();
    if (locals.size() != 1) return false;
    // a weight != 1 of the point matcher's layer pair is a device input (round 5); the plane matcher's stays at 1 (what upstream
    // does with it is unverified)
    if (weight_out) *weight_out = locals.begin()->second;
    else if (locals.begin()->second != 1.0) return false;
    if (!g.empty() && (g != gname || l != locals.begin()->first)) return false;  // both matchers on the same layers (ndt yaml:199-200,209-210)
    g = gname;
    l = locals.begin()->first;
    return true;
}
}  // namespace

/** Drop-in for mp2p_icp::ICP: same align() signature as the call at LidarOdometry.cpp:961-962. */
class ICP_HIP : public ICP
{
    DEFINE_MRPT_OBJECT(ICP_HIP, mp2p_icp)
   public:
    ICP_HIP() = default;   // the device session (context on GPU MOLA_HIP_DEVICE) is created by the first align()
    ~ICP_HIP() override = default;

    void align(
        const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const mrpt::math::TPose3D& initialGuessLocalWrtGlobal,
        const Parameters& p, Results& result, const std::optional<mrpt::poses::CPose3DPDFGaussianInf>& prior = std::nullopt,
        const mrpt::optional_ref<LogRecord>& outputDebugInfo = std::nullopt) override  // [U]
    {
        const auto& sw = molahip_host::plugin_switches();
        // generateDebugFiles: the upstream loop fills the LogRecord and applies Parameters::functor_before_logging_local
        // (set at LidarOdometry.cpp:360-364) itself before it writes the .icplog.
        Shape sh;
        mh_map* dmap = nullptr;
        if (sw.force_cpu || p.generateDebugFiles || !recognise(sh) || !pcLocal.layers.count(sh.localLayer) || !pcGlobal.layers.count(sh.globalLayer))
            return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);
        if (!dev_) dev_ = std::make_unique<molahip_mrpt::DeviceSession>();
        if (!(dmap = dev_->device_map_of(*pcGlobal.layers.at(sh.globalLayer), sh.pl != nullptr)))
            return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);
        const auto* local = dynamic_cast<const mrpt::maps::CPointsMap*>(pcLocal.layers.at(sh.localLayer).get());
        if (!local) return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);

        // keeps profiler() populated (enabled at LidarOdometry.cpp:351-352): the call as a whole here, the device's own
        // sections below once the result is back
        const bool profiling = profiler().isEnabled();  // [U] CTimeLogger::isEnabled
        mrpt::system::CTimeLoggerEntry tle(profiler(), "align_hip");

        // thresholds: functions of ICP_ITERATION (lidar3d-default.yaml:190,198; ndt yaml:197), evaluated per iteration through the
        // attached ParameterSources -- LAZILY: an alignment of the shipped pipelines ends after ~21 of its 300 iterations, and every
        // evaluation is an updateVariable + realize() of every attached parameter.  The first kScheduleFirstStage iterations are
        // evaluated before the first run; a run that exhausts them without terminating is repeated with the whole budget (same
        // inputs, same result as a run that had the whole schedule from the start).
        std::vector<double> thr, kp, thr_pl;
        auto ensure_schedule = [&](uint32_t upto) {
            for (uint32_t k = static_cast<uint32_t>(thr.size()); k < upto; k++)
            {
                for (auto* src : attachedSources()) { src->updateVariable("ICP_ITERATION", k); src->realize(); }  // [U]
                thr.push_back(sh.pt->threshold);
                kp.push_back(sh.gn->robustKernelParam);
                if (sh.pl) thr_pl.push_back(sh.pl->distanceThreshold);
            }
        };
        constexpr uint32_t kScheduleFirstStage = 48;

        const auto& lx = local->getPointsBufferRef_x();  // already SoA
        const auto& ly = local->getPointsBufferRef_y();
        const auto& lz = local->getPointsBufferRef_z();
        mh_scan* scan_ = dev_->upload(*local);

        mh_icp_params ip{};
        ip.max_iterations        = p.maxIterations;
        ip.min_abs_step_trans    = p.minAbsStep_trans;
        ip.min_abs_step_rot      = p.minAbsStep_rot;
        // (threshold / kernel_param / pt2pl_threshold: set by run() below, once the schedule covers the run's budget)
        ip.pt2pl_mode            = sw.pt2pl_mode;
        // U12: upstream's matchers skip local points an earlier matcher paired unless allowMatchAlreadyMatchedPoints [U] is set
        ip.matched_points        = (sh.pl && !sh.pt->allowMatchAlreadyMatchedPoints_) ? sw.matched_points : static_cast<uint32_t>(MH_MATCHED_POINTS_PAIR_AGAIN);
        ip.threshold_angular_deg = sh.pt->thresholdAngularDeg;
        ip.gn.max_inner_iterations = sh.gn->maxIterations;
        // RobustKernel [U]: None / GemanMcClure / Cauchy by NAME (the numeric values of the upstream enum are not relied on)
        const std::string kname = mrpt::typemeta::TEnumType<RobustKernel>::value2name(sh.gn->robustKernel);  // [U]
        ip.gn.robust_kernel      = molahip_host::kernel_from_upstream_name(kname.c_str(), sw);
        ip.gn.min_delta          = sw.min_delta;
        ip.gn.max_cost           = sw.max_cost;
        ip.gn.weight_pt2pt = sh.ptLayerWeight;  // Pairings::point_weights [U]: one layer pair, one weight
        ip.gn.weight_pt2pl = 1.0;
        ip.compute_covariance    = 1;
        ip.cov_findif_xyz        = sw.cov_step_xyz;
        ip.cov_findif_ang        = sw.cov_step_ang;
        ip.poll_every = 0;
        ip.profile    = profiling ? 1 : 0;  // per-launch device timing (the chunk is then enqueued kernel by kernel)

        double T0[12];
        pose_to_T12(mrpt::poses::CPose3D(initialGuessLocalWrtGlobal), T0);
        mh_prior pr;
        if (prior)
        {
            pose_to_T12(prior->mean, pr.mean);
            for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) pr.info[i * 6 + j] = prior->cov_inv(i, j);
        }
        mh_icp_result r{};
        mh_pairs_out po = dev_->pairs.out(lx.size());  // (buffers of the session: no per-call allocation once warm)
        const auto &li = dev_->pairs.li, &gi = dev_->pairs.gi;
        const auto &gx = dev_->pairs.gx, &gy = dev_->pairs.gy, &gz = dev_->pairs.gz, &d2 = dev_->pairs.d2;
        auto run_with = [&](uint32_t budget, mh_icp_iter* trace) {
            ensure_schedule(budget);
            mh_icp_params q = ip;
            q.max_iterations  = budget;
            q.threshold       = thr.data();
            q.kernel_param    = kp.data();
            q.pt2pl_threshold = sh.pl ? thr_pl.data() : nullptr;
            mh_icp_result rr{};
            mh_check(mh_icp_align(dmap, scan_, &q, T0, prior ? &pr : nullptr, &rr, trace, &po, MH_MEM_HOST), "mh_icp_align");
            return rr;
        };
        auto run = [&](uint32_t budget, mh_icp_iter* trace) {
            // (the first stage covers what the PREVIOUS alignment of this object needed, with a margin: consecutive scans converge
            //  in about as many iterations, so the second run from scratch stays the exception also for slow sequences -- ADVICE r5)
            uint32_t stage = kScheduleFirstStage;
            if (last_iterations_ + 16u > stage) stage = last_iterations_ + 16u;
            const uint32_t first = budget < stage ? budget : stage;
            mh_icp_result rr = run_with(first, trace);
            if (first < budget && rr.termination_reason == MH_TERM_MAX_ITERATIONS) rr = run_with(budget, trace);  // rare
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
        last_iterations_ = r.n_iterations;
        {
            // the attached sources are left where upstream's loop leaves them: at the last iteration it EXECUTED [U] -- not at the
            // last index the lazy schedule evaluated (a formula read after align() must not see another iteration count)
            const uint32_t last_k = r.n_iterations ? r.n_iterations - 1u : 0u;
            for (auto* src : attachedSources()) { src->updateVariable("ICP_ITERATION", last_k); src->realize(); }  // [U]
        }

        if (profiling)
        {
            // one section per kernel family of the device loop (the reference's profiler prints them beside its own
            // "onLidar.*" sections): the correspondence search, everything else the stream ran for this call (Gauss-Newton
            // accumulation, 6x6 solve, covariance, pairing compaction), and the host's polls of the loop
            profiler().registerUserMeasure("align_hip.device.match_kernels", 1e-3 * r.match_kernel_ms, true);  // [U] (name, value, is_time)
            profiler().registerUserMeasure("align_hip.device.accumulate_solve_covariance", 1e-3 * (r.total_ms - r.match_kernel_ms), true);
            profiler().registerUserMeasure("align_hip.device.stream_total", 1e-3 * r.total_ms, true);
            profiler().registerUserMeasure("align_hip.match_launches", r.n_match_launches);
            profiler().registerUserMeasure("align_hip.host_polls", r.n_host_polls);
            profiler().registerUserMeasure("align_hip.iterations_enqueued", r.n_enqueued_iterations);
        }
        // results back into the upstream structures (Results::finalPairings is read by LidarOdometry and the log writer)
        mrpt::math::CMatrixDouble44 M = mrpt::math::CMatrixDouble44::Identity();
        for (int i = 0; i < 3; i++) for (int j = 0; j < 4; j++) M(i, j) = r.T[i * 4 + j];
        result.optimal_tf.mean = mrpt::poses::CPose3D(M);
        for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) result.optimal_tf.cov(i, j) = r.cov[i * 6 + j];
        result.quality           = r.quality;
        result.nIterations       = r.n_iterations;
        result.terminationReason = molahip_host::term_reason_to<IterTermReason>(r.termination_reason);  // by name, not by value
        result.finalPairings     = Pairings();
        result.finalPairings.potential_pairings = r.potential_pairings;
        const uint32_t n_pt = r.n_final_pairs - r.n_final_pairs_pt2pl;
        result.finalPairings.paired_pt2pt.reserve(n_pt);
        for (uint32_t k = 0; k < n_pt; k++)
        {
            mrpt::tfest::TMatchingPair mp;
            mp.globalIdx = gi[k];
            mp.localIdx  = li[k];
            mp.global    = {gx[k], gy[k], gz[k]};
            mp.local     = {lx[li[k]], ly[li[k]], lz[li[k]]};
            mp.errorSquareAfterTransformation = d2[k];
            result.finalPairings.paired_pt2pt.push_back(mp);
        }
        if (r.n_final_pairs_pt2pl)
        {
            // Pairings::paired_pt2pl [U]: {pl_global{plane, centroid}, pt_local}
            const size_t n = r.n_final_pairs_pt2pl;
            mh_pairs_pl_out plo = dev_->planes.out(lx.size());
            const auto &pli = dev_->planes.li;
            const auto &cx = dev_->planes.cx, &cy = dev_->planes.cy, &cz = dev_->planes.cz;
            const auto &nx = dev_->planes.nx, &ny = dev_->planes.ny, &nz = dev_->planes.nz;
            uint64_t got = 0;
            mh_check(mh_icp_get_pt2pl_pairs(scan_, &plo, MH_MEM_HOST, &got), "mh_icp_get_pt2pl_pairs");
            result.finalPairings.paired_pt2pl.reserve(n);
            for (size_t k = 0; k < got; k++)
            {
                point_plane_pair_t pp;  // [U]
                pp.pl_global.centroid = {cx[k], cy[k], cz[k]};
                pp.pl_global.plane    = mrpt::math::TPlane(mrpt::math::TPoint3D(cx[k], cy[k], cz[k]), mrpt::math::TVector3D(nx[k], ny[k], nz[k]));
                pp.pt_local           = {lx[pli[k]], ly[pli[k]], lz[pli[k]]};
                result.finalPairings.paired_pt2pl.push_back(pp);
            }
        }
        trace_.row("hip", lx.size(), result);
    }

   private:
    /** The upstream CPU loop, as it is (unknown pipeline shape / map class, debug files, MOLA_HIP_FORCE_CPU). */
    void upstream_align(const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const mrpt::math::TPose3D& guess,
                        const Parameters& p, Results& result, const std::optional<mrpt::poses::CPose3DPDFGaussianInf>& prior,
                        const mrpt::optional_ref<LogRecord>& outputDebugInfo)
    {
        ICP::align(pcLocal, pcGlobal, guess, p, result, prior, outputDebugInfo);
        trace_.row("cpu", 0, result);
    }

    /** Does the configured pipeline have one of the two shapes of the file header? */
    bool recognise(Shape& sh) const
    {
        if (solvers().size() != 1 || !(sh.gn = dynamic_cast<const Solver_GaussNewton*>(solvers()[0].get()))) return false;
        const auto& ms = matchers();
        if (ms.size() == 1)
            sh.pt = dynamic_cast<const Matcher_Points_DistanceThreshold*>(ms[0].get());
        else if (ms.size() == 2)  // Matcher_Point2Plane runs BEFORE the point matcher (lidar3d-ndt.yaml:195-210)
        {
            sh.pl = dynamic_cast<const Matcher_Point2Plane*>(ms[0].get());
            sh.pt = dynamic_cast<const Matcher_Points_DistanceThreshold*>(ms[1].get());
            if (!sh.pl || !single_unit_layer(*sh.pl, sh.globalLayer, sh.localLayer)) return false;
        }
        if (!sh.pt || sh.pt->pairingsPerPoint != 1 || !sh.pt->allowMatchAlreadyMatchedGlobalPoints) return false;
        if (sh.pt->runFromIteration != 0 || sh.pt->runUpToIteration != 0) return false;  // [U] iteration gates: unused by both files
        return single_unit_layer(*sh.pt, sh.globalLayer, sh.localLayer, &sh.ptLayerWeight);
    }

    std::unique_ptr<molahip_mrpt::DeviceSession> dev_;  // context, map mirrors, staging scan, result buffers
    uint32_t last_iterations_ = 0;  // nIterations of the previous align(): sizes the first stage of the lazy threshold schedule
    AlignTrace trace_;
};
IMPLEMENTS_MRPT_OBJECT(ICP_HIP, mp2p_icp::ICP, mp2p_icp)

}  // namespace mp2p_icp

// same registration pattern as module/src/register.cpp:40-46 (the granular classes register themselves in
// mp2p_icp_granular.cpp, which is part of the same library)
MRPT_INITIALIZER(do_register_molahip_mp2p_icp) { mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::ICP_HIP)); }
