// mp2p_icp_plugin.cpp -- the adapter a MOLA maintainer builds where mp2p_icp + MRPT + mola_metric_maps ARE installed.
//
// NOT compiled in this repository's image (mp2p_icp, mrpt-*, mola_* are absent: SURVEY.md 0.2); the upstream
// signatures below are written from the upstream API as recalled in SURVEY.md 8(b) and are marked [U]: re-check
// them against the installed headers.  It derives from the REAL mp2p_icp::ICP so that YAML parameter parsing
// (DECLARE_PARAMETER_*), the iteration hook, the profiler and the ParameterSource attachment keep working, reads the
// parsed parameters of the UPSTREAM matcher / solver objects the pipeline file names, and forwards the numeric work to the
// C ABI of libmolahip (include/molahip.h).  Registration uses the same RTTI mechanism as the reference's own module
// (module/src/register.cpp:40-46).  The ONLY name this library adds to MRPT's class factory for the ICP side is
//
//     mp2p_icp::ICP_HIP
//
// and the pipeline files that select it are the reference's own files with that one class_name changed
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
// both with one {global, local, weight: 1} entry in pointLayerMatches (yaml :203-204), pairingsPerPoint 1.
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
//   MOLA_HIP_FORCE_CPU=1     every call goes to the upstream loop (sanity A/A through the same plugin)
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
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "hashed_voxel_pointcloud_hip.h"
#include "molahip.h"
#include "molahip_host/hook_replay.h"  // the opaque iteration hook on the fused loop (compiled + tested via host/src/icp.cpp)
#include "molahip_host/plugin_switches.h"  // MOLA_HIP_* environment switches (compiled + tested via host/src/icp.cpp)

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

// `voxel_size()` getters differ between mola_metric_maps versions [U]: use the getter when the class has one, else the
// value MOLAHIP_VOXEL_SIZE gives (the plugin refuses to guess: a wrong voxel size silently changes every pairing).
template <class M, class = void> struct has_voxel_size : std::false_type {};
template <class M> struct has_voxel_size<M, std::void_t<decltype(std::declval<const M&>().voxel_size())>> : std::true_type {};
template <class M> float voxel_size_of(const M& m)
{
    if (const char* e = getenv("MOLAHIP_VOXEL_SIZE")) return static_cast<float>(atof(e));
    if constexpr (has_voxel_size<M>::value) return m.voxel_size();
    else throw std::runtime_error("libmolahip plugin: this mola_metric_maps version has no voxel_size() getter; set MOLAHIP_VOXEL_SIZE");
}

inline uint64_t fnv(uint64_t h, uint32_t b) { return (h ^ b) * 1099511628211ull; }
inline uint32_t fbits(float v) { uint32_t b; memcpy(&b, &v, 4); return b; }

/** What the mirror needs to know about one host map layer, whatever its class. */
struct HostMapView
{
    mh_map_params params{};
    uint64_t fingerprint = 0;  // changes whenever the stored content does
    std::function<void(std::vector<float>&, std::vector<float>&, std::vector<float>&)> gather;  // all stored points, voxel by voxel
};

/** Voxel-hashed upstream maps (HashedVoxelPointCloud, NDT).  Their content only changes by insertPoint (append to a
 *  voxel below its cap) and by far-voxel removal, so {voxel index, point count} over all voxels identifies the content:
 *  O(occupied voxels) per align() instead of O(points); the points themselves are read on a change only (key-frames). */
template <class VoxelMap> void view_voxel_map(const VoxelMap& m, HostMapView& v)
{
    uint64_t h = 1469598103934665603ull, n = 0;
    m.visitAllVoxels([&](const auto& idx, const auto& vox) {  // [U] visitAllVoxels(f(index3d_t, VoxelData))
        const uint32_t cnt = static_cast<uint32_t>(vox.points().size());  // [U] VoxelData::points()
        // order-independent combination: the hash container's iteration order may change when it rehashes
        uint64_t e = fnv(fnv(fnv(fnv(1469598103934665603ull, (uint32_t)idx.cx), (uint32_t)idx.cy), (uint32_t)idx.cz), cnt);
        h += e * 0x9E3779B97F4A7C15ull;
        n += cnt;
    });
    v.fingerprint = h ^ (n << 1);
    v.gather = [&m](std::vector<float>& x, std::vector<float>& y, std::vector<float>& z) {
        m.visitAllPoints([&](const mrpt::math::TPoint3Df& p) { x.push_back(p.x); y.push_back(p.y); z.push_back(p.z); });  // [U]
    };
}

inline bool view_of(const mrpt::maps::CMetricMap& g, HostMapView& v)
{
    const auto& sw = molahip_host::plugin_switches();
    v.params = mh_map_params{};
    v.params.index_mode = sw.index_mode;
    if (const auto* hv = dynamic_cast<const mola::HashedVoxelPointCloud*>(&g))
    {
        v.params.voxel_size                  = voxel_size_of(*hv);
        v.params.max_points_per_voxel        = hv->insertionOptions.max_points_per_voxel;        // [U] yaml:235
        v.params.min_distance_between_points = hv->insertionOptions.min_distance_between_points; // [U] yaml:236
        view_voxel_map(*hv, v);
    }
    else if (const auto* nd = dynamic_cast<const mola::NDT*>(&g))
    {
        v.params.voxel_size                  = voxel_size_of(*nd);
        v.params.max_points_per_voxel        = nd->insertionOptions.max_points_per_voxel;         // [U] ndt yaml:241
        v.params.min_distance_between_points = nd->insertionOptions.min_distance_between_points;  // [U] ndt yaml:242
        v.params.ndt_max_eigen_ratio         = nd->insertionOptions.max_eigen_ratio_for_planes;   // [U] ndt yaml:246
        view_voxel_map(*nd, v);
    }
    else if (const auto* pm = dynamic_cast<const mrpt::maps::CPointsMap*>(&g))
    {
        // a flat point map has no voxel structure of its own: the device table uses 1 m voxels without a cap, so the
        // 27-voxel search reaches >= 1 m (upstream's KD-tree search is unbounded: pairs farther than that are lost)
        v.params.voxel_size = getenv("MOLAHIP_VOXEL_SIZE") ? static_cast<float>(atof(getenv("MOLAHIP_VOXEL_SIZE"))) : 1.0f;
        const auto& x = pm->getPointsBufferRef_x();
        const auto& y = pm->getPointsBufferRef_y();
        const auto& z = pm->getPointsBufferRef_z();
        uint64_t h = fnv(1469598103934665603ull, (uint32_t)x.size());
        const size_t n = x.size(), step = n > 4096 ? n / 4096 : 1;
        for (size_t i = 0; i < n; i += step) h = fnv(fnv(fnv(h, fbits(x[i])), fbits(y[i])), fbits(z[i]));
        if (n) h = fnv(fnv(fnv(h, fbits(x[n - 1])), fbits(y[n - 1])), fbits(z[n - 1]));
        v.fingerprint = h;
        v.gather = [pm](std::vector<float>& ox, std::vector<float>& oy, std::vector<float>& oz) {
            ox = pm->getPointsBufferRef_x(); oy = pm->getPointsBufferRef_y(); oz = pm->getPointsBufferRef_z();
        };
    }
    else
        return false;
    return true;
}

/** Device mirror of one host map layer.  The plugin does not own the host map (no change notification), so the mirror
 *  is rebuilt when the view's parameters or fingerprint change -- SURVEY.md 7.3 "map mirror coherence".  The stored
 *  points arrive voxel by voxel, already capped, so mh_map_build (clear + insertPoint in order) reproduces every voxel's
 *  content and in-voxel order.  The proper fix is the device-owned CMetricMap class next to this file (row f2). */
struct MapMirror
{
    mh_map*       map = nullptr;
    mh_map_params params{};
    uint64_t      fingerprint = 0;
    bool          built = false;
};

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
};

template <class M> bool single_unit_layer(const M& m, std::string& g, std::string& l)
{
    if (m.weight_pt2pt_layers.size() != 1) return false;                     // [U] {global -> {local -> weight}}
    const auto& [gname, locals] = *m.weight_pt2pt_layers.begin();
    if (locals.size() != 1 || locals.begin()->second != 1.0) return false;  // per-layer weights != 1: not in the fused path
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
    ICP_HIP() { mh_check(mh_ctx_create(0, nullptr, &ctx_), "mh_ctx_create"); }
    ~ICP_HIP() override
    {
        for (auto& kv : mirrors_) if (kv.second.map) mh_map_destroy(kv.second.map);
        if (scan_) mh_scan_destroy(scan_);
        mh_ctx_destroy(ctx_);
    }

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
        if (sw.force_cpu || p.generateDebugFiles || !recognise(sh) || !pcLocal.layers.count(sh.localLayer) ||
            !pcGlobal.layers.count(sh.globalLayer) || !(dmap = device_map_of(*pcGlobal.layers.at(sh.globalLayer), sh.pl != nullptr)))
            return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);
        const auto* local = dynamic_cast<const mrpt::maps::CPointsMap*>(pcLocal.layers.at(sh.localLayer).get());
        if (!local) return upstream_align(pcLocal, pcGlobal, initialGuessLocalWrtGlobal, p, result, prior, outputDebugInfo);

        mrpt::system::CTimeLoggerEntry tle(profiler(), "align_hip");  // keeps profiler() populated (LidarOdometry.cpp:351-352)

        // thresholds: functions of ICP_ITERATION (lidar3d-default.yaml:190,198; ndt yaml:197): evaluate per iteration up front
        std::vector<double> thr(p.maxIterations), kp(p.maxIterations), thr_pl(sh.pl ? p.maxIterations : 0);
        for (uint32_t k = 0; k < p.maxIterations; k++)
        {
            for (auto* src : attachedSources()) { src->updateVariable("ICP_ITERATION", k); src->realize(); }  // [U]
            thr[k] = sh.pt->threshold;
            kp[k]  = sh.gn->robustKernelParam;
            if (sh.pl) thr_pl[k] = sh.pl->distanceThreshold;
        }

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
        ip.pt2pl_threshold       = sh.pl ? thr_pl.data() : nullptr;
        ip.pt2pl_mode            = sw.pt2pl_mode;
        ip.threshold_angular_deg = sh.pt->thresholdAngularDeg;
        ip.gn.max_inner_iterations = sh.gn->maxIterations;
        // RobustKernel [U]: None / GemanMcClure / Cauchy by NAME (the numeric values of the upstream enum are not relied on)
        const std::string kname = mrpt::typemeta::TEnumType<RobustKernel>::value2name(sh.gn->robustKernel);  // [U]
        ip.gn.robust_kernel      = molahip_host::kernel_from_upstream_name(kname.c_str(), sw);
        ip.gn.min_delta          = sw.min_delta;
        ip.gn.max_cost           = sw.max_cost;
        ip.gn.weight_pt2pt = ip.gn.weight_pt2pl = 1.0;
        ip.compute_covariance    = 1;
        ip.cov_findif_xyz        = sw.cov_step_xyz;
        ip.cov_findif_ang        = sw.cov_step_ang;
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
            std::vector<uint32_t> pli(lx.size());
            std::vector<float> cx(lx.size()), cy(lx.size()), cz(lx.size()), nx(lx.size()), ny(lx.size()), nz(lx.size());
            mh_pairs_pl_out plo{pli.data(), cx.data(), cy.data(), cz.data(), nx.data(), ny.data(), nz.data()};
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
        return single_unit_layer(*sh.pt, sh.globalLayer, sh.localLayer);
    }

    /** The mh_map to align against: the handle of a device-owned map, or the (re)built mirror of a host map; nullptr
     *  when the layer's class is not one this plugin reads (-> upstream CPU loop). */
    mh_map* device_map_of(const mrpt::maps::CMetricMap& g, bool need_ndt)
    {
        // a device-owned local map (hashed_voxel_pointcloud_hip.h): nothing to mirror, the handle is the map.
        // (Its context must be the one this ICP runs on: both use device 0's default stream here.)
        if (const auto* dm = dynamic_cast<const mola::HashedVoxelPointCloudHIP*>(&g)) return need_ndt ? nullptr : dm->deviceHandle();
        HostMapView v;
        if (!view_of(g, v)) return nullptr;
        if (need_ndt && !(v.params.ndt_max_eigen_ratio > 0)) return nullptr;  // Matcher_Point2Plane on a non-NDT map: KNN+PCA upstream
        auto& mir = mirrors_[&g];
        if (mir.map && memcmp(&mir.params, &v.params, sizeof(v.params)) != 0)
        {
            mh_map_destroy(mir.map);
            mir = MapMirror();
        }
        if (!mir.map)
        {
            mh_check(mh_map_create(ctx_, &v.params, &mir.map), "mh_map_create");
            mir.params = v.params;
        }
        if (!mir.built || mir.fingerprint != v.fingerprint)
        {
            std::vector<float> x, y, z;
            v.gather(x, y, z);
            mh_check(mh_map_build(mir.map, x.data(), y.data(), z.data(), x.size(), MH_MEM_HOST), "mh_map_build");
            mir.fingerprint = v.fingerprint;
            mir.built       = true;
        }
        return mir.map;
    }

    mh_ctx*  ctx_  = nullptr;
    mh_scan* scan_ = nullptr;
    std::unordered_map<const mrpt::maps::CMetricMap*, MapMirror> mirrors_;
    AlignTrace trace_;
};
IMPLEMENTS_MRPT_OBJECT(ICP_HIP, mp2p_icp::ICP, mp2p_icp)

}  // namespace mp2p_icp

// same registration pattern as module/src/register.cpp:40-46
MRPT_INITIALIZER(do_register_molahip_mp2p_icp) { mrpt::rtti::registerClass(CLASS_ID(mp2p_icp::ICP_HIP)); }
