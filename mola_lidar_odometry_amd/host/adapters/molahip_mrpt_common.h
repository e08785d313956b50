// molahip_mrpt_common.h -- what the MRPT-side adapters share: status -> exception, pose conversion, the device session
// (one context per process on the GPU MOLA_HIP_DEVICE names), and the device MIRROR of a host map layer.
//
// NOT compiled in this repository's image (mp2p_icp, mrpt-*, mola_* are absent: SURVEY.md 0.2); lines written from the
// upstream API as recalled are marked [U].  tests/test_adapter_syntax.py compiles every adapter source with
// -fsyntax-only against the minimal stand-in headers under tests/stubs/ (scaffolding: they pin nothing about upstream,
// they only catch plain C++ errors here).
#pragma once
#include <mola_metric_maps/HashedVoxelPointCloud.h>  // [U] mola::HashedVoxelPointCloud
#include <mola_metric_maps/NDT.h>                    // [U] mola::NDT
#include <mrpt/maps/CMetricMap.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/poses/CPose3D.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "hashed_voxel_pointcloud_hip.h"
#include "molahip.h"
#include "molahip_host/plugin_switches.h"  // MOLA_HIP_* environment switches (compiled + tested via host/src/icp.cpp)

namespace molahip_mrpt
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


using molahip_host::device_index;  // MOLA_HIP_DEVICE (default 0)

/** Device mirrors of host map layers + a staging scan per host layer, owned by ONE context (an ICP_HIP instance, or the
 *  process-wide session of the granular matcher / solver classes, which are const objects shared between threads [U]). */
class DeviceSession
{
   public:
    explicit DeviceSession(int device = device_index()) { mh_check(mh_ctx_create(device, nullptr, &ctx_), "mh_ctx_create"); }
    ~DeviceSession()
    {
        for (auto& kv : mirrors_) if (kv.second.map) mh_map_destroy(kv.second.map);
        for (auto& kv : scans_) if (kv.second) mh_scan_destroy(kv.second);
        mh_ctx_destroy(ctx_);
    }
    DeviceSession(const DeviceSession&) = delete;
    DeviceSession& operator=(const DeviceSession&) = delete;

    mh_ctx* ctx() const { return ctx_; }
    std::mutex& mutex() { return mtx_; }

    /** The mh_map to search: the handle of a device-owned map, or the (re)built mirror of a host map; nullptr when the
     *  layer's class is not one the adapters read (-> the caller delegates to the upstream CPU code). */
    mh_map* device_map_of(const mrpt::maps::CMetricMap& g, bool need_ndt)
    {
        // a device-owned local map (hashed_voxel_pointcloud_hip.h): nothing to mirror, the handle is the map
        // (its context must be on the same device as this session's).
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
            x_.clear(), y_.clear(), z_.clear();
            v.gather(x_, y_, z_);
            mh_check(mh_map_build(mir.map, x_.data(), y_.data(), z_.data(), x_.size(), MH_MEM_HOST), "mh_map_build");
            mir.fingerprint = v.fingerprint;
            mir.built       = true;
        }
        return mir.map;
    }

    /** The local layer on the device (created on first use, refilled every call: the layer changes every scan). */
    mh_scan* upload(const mrpt::maps::CPointsMap& local)
    {
        const auto& lx = local.getPointsBufferRef_x();  // already SoA [U]
        const auto& ly = local.getPointsBufferRef_y();
        const auto& lz = local.getPointsBufferRef_z();
        mh_scan*& sc = scans_[&local];
        if (!sc) mh_check(mh_scan_create(ctx_, lx.data(), ly.data(), lz.data(), lx.size(), MH_MEM_HOST, &sc), "mh_scan_create");
        else     mh_check(mh_scan_update(sc, lx.data(), ly.data(), lz.data(), lx.size(), MH_MEM_HOST), "mh_scan_update");
        if (scans_.size() > 16)  // host layers come and go (one per observation): keep the table from growing
        {
            for (auto it = scans_.begin(); it != scans_.end();)
                if (it->first != &local) { mh_scan_destroy(it->second); it = scans_.erase(it); } else ++it;
        }
        return scans_[&local];
    }

    /** Reusable host result arrays of n entries each (no per-call heap traffic once warm). */
    struct PairBuffers
    {
        std::vector<uint32_t> li, gi;
        std::vector<float> gx, gy, gz, d2;
        mh_pairs_out out(size_t n)
        {
            if (li.size() < n) { li.resize(n); gi.resize(n); gx.resize(n); gy.resize(n); gz.resize(n); d2.resize(n); }
            return mh_pairs_out{li.data(), gi.data(), gx.data(), gy.data(), gz.data(), d2.data()};
        }
    };
    struct PlaneBuffers
    {
        std::vector<uint32_t> li;
        std::vector<float> cx, cy, cz, nx, ny, nz;
        mh_pairs_pl_out out(size_t n)
        {
            if (li.size() < n) { li.resize(n); cx.resize(n); cy.resize(n); cz.resize(n); nx.resize(n); ny.resize(n); nz.resize(n); }
            return mh_pairs_pl_out{li.data(), cx.data(), cy.data(), cz.data(), nx.data(), ny.data(), nz.data()};
        }
    };
    PairBuffers  pairs;
    PlaneBuffers planes;

    /** The session the granular matcher / solver classes share (created on first use). */
    static DeviceSession& process_wide()
    {
        static DeviceSession s;
        return s;
    }

   private:
    mh_ctx* ctx_ = nullptr;
    std::mutex mtx_;
    std::unordered_map<const mrpt::maps::CMetricMap*, MapMirror> mirrors_;
    std::unordered_map<const mrpt::maps::CPointsMap*, mh_scan*> scans_;
    std::vector<float> x_, y_, z_;
};

}  // namespace molahip_mrpt
