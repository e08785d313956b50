// icp.cpp -- mp2p_icp_hip: the reference's plugin classes re-stated as thin drivers of the C ABI.
// No arithmetic of the hot path happens here; see include/molahip.h for what each call replaces.
#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

#include "mp2p_icp_hip/mp2p_icp_hip.h"
#include "molahip_host/hook_replay.h"
#include "molahip_host/plugin_switches.h"

namespace mp2p_icp_hip {

// ================================================================== poses (host-side glue only)
CPose3D::CPose3D(const TPose3D& p) {
  const double cy = cos(p.yaw), sy = sin(p.yaw), cp = cos(p.pitch), sp = sin(p.pitch), cr = cos(p.roll), sr = sin(p.roll);
  const double m[12] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, p.x,
                        sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr, p.y,
                        -sp,     cp * sr,                cp * cr,                p.z};
  memcpy(T, m, sizeof(m));
}

TPose3D CPose3D::asTPose() const {
  TPose3D p;
  p.x = T[3]; p.y = T[7]; p.z = T[11];
  const double c = std::hypot(T[0], T[4]);
  p.pitch = atan2(-T[8], c);
  if (c > 1e-12) {
    p.yaw = atan2(T[4], T[0]);
    p.roll = atan2(T[9], T[10]);
  } else {
    p.yaw = atan2(-T[1], T[5]);
    p.roll = 0;
  }
  return p;
}

CPose3D CPose3D::operator+(const CPose3D& b) const {
  CPose3D c;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) c.T[i * 4 + j] = T[i * 4] * b.T[j] + T[i * 4 + 1] * b.T[4 + j] + T[i * 4 + 2] * b.T[8 + j];
    c.T[i * 4 + 3] = T[i * 4] * b.T[3] + T[i * 4 + 1] * b.T[7] + T[i * 4 + 2] * b.T[11] + T[i * 4 + 3];
  }
  return c;
}

CPose3D CPose3D::operator-(const CPose3D& b) const {  // b^-1 (+) a  (mrpt: a - b)
  CPose3D bi;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) bi.T[i * 4 + j] = b.T[j * 4 + i];
    bi.T[i * 4 + 3] = -(b.T[i] * b.T[3] + b.T[4 + i] * b.T[7] + b.T[8 + i] * b.T[11]);
  }
  return bi + *this;
}

double CPose3D::translationNorm() const { return std::sqrt(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]); }

double CPose3D::rotationAngle() const {
  const double vx = T[9] - T[6], vy = T[2] - T[8], vz = T[4] - T[1];
  const double s2 = std::sqrt(vx * vx + vy * vy + vz * vz);
  double cth = 0.5 * (T[0] + T[5] + T[10] - 1.0);
  cth = std::min(1.0, std::max(-1.0, cth));
  return atan2(0.5 * s2, cth);
}

void CPose3D::so3Log(double w[3]) const {
  const double vx = T[9] - T[6], vy = T[2] - T[8], vz = T[4] - T[1];
  const double s2 = std::sqrt(vx * vx + vy * vy + vz * vz);  // 2 sin(th)
  const double th = rotationAngle();
  if (th < 1e-7) {
    const double k = 0.5 * (1.0 + th * th / 6.0);
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  if (M_PI - th > 1e-6) {
    const double k = th / s2;
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
    return;
  }
  // th ~ pi: R + I ~ 2 n n^T
  const double d[3] = {T[0], T[5], T[10]};
  const int k = (d[0] >= d[1] && d[0] >= d[2]) ? 0 : (d[1] >= d[2] ? 1 : 2);
  double n[3];
  const double nk = std::sqrt(std::max(0.0, 0.5 * (d[k] + 1.0)));
  for (int j = 0; j < 3; j++) n[j] = (j == k) ? nk : 0.25 * (T[k * 4 + j] + T[j * 4 + k]) / nk;
  const double dot = n[0] * vx + n[1] * vy + n[2] * vz;
  const double nn = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  const double sc = (dot < 0.0 ? -th : th) / nn;
  for (int j = 0; j < 3; j++) w[j] = sc * n[j];
}

CPose3D CPose3D::FromRotVecAndTranslation(const double w[3], const double t[3]) {
  const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = std::sqrt(th2);
  double a, b;  // sin(th)/th, (1-cos th)/th^2
  if (th < 1e-2) {
    a = 1.0 - th2 / 6.0 * (1.0 - th2 / 20.0 * (1.0 - th2 / 42.0));
    b = 0.5 - th2 / 24.0 * (1.0 - th2 / 30.0 * (1.0 - th2 / 56.0));
  } else {
    const double sh = std::sin(0.5 * th);
    a = std::sin(th) / th;
    b = 2.0 * sh * sh / th2;
  }
  const double W[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
  const double xx = w[0] * w[0], yy = w[1] * w[1], zz = w[2] * w[2], xy = w[0] * w[1], xz = w[0] * w[2], yz = w[1] * w[2];
  const double W2[9] = {-(yy + zz), xy, xz, xy, -(xx + zz), yz, xz, yz, -(xx + yy)};
  CPose3D p;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) p.T[i * 4 + j] = ((i == j) ? 1.0 : 0.0) + a * W[i * 3 + j] + b * W2[i * 3 + j];
    p.T[i * 4 + 3] = t[i];
  }
  return p;
}

// |v| and |w| of log_SE3(d) = [V^-1 t; w]  
static void se3_log_norms(const CPose3D& d, double& nt, double& nr) {
  const double* T = d.T;
  const double vx = T[9] - T[6], vy = T[2] - T[8], vz = T[4] - T[1];
  const double s2 = std::sqrt(vx * vx + vy * vy + vz * vz);
  const double th = d.rotationAngle();
  double w[3] = {0, 0, 0};
  if (s2 > 1e-300) {
    const double k = (th < 1e-7) ? 0.5 * (1.0 + th * th / 6.0) : th / s2;
    w[0] = k * vx; w[1] = k * vy; w[2] = k * vz;
  }
  const double th2 = th * th;
  const double kk = th < 1e-2 ? 1.0 / 12.0 + th2 / 720.0 + th2 * th2 / 30240.0 : (1.0 - 0.5 * th / std::tan(0.5 * th)) / th2;
  const double t[3] = {T[3], T[7], T[11]};
  // V^-1 t = t - 0.5 w x t + kk w x (w x t)
  const double c1[3] = {w[1] * t[2] - w[2] * t[1], w[2] * t[0] - w[0] * t[2], w[0] * t[1] - w[1] * t[0]};
  const double c2[3] = {w[1] * c1[2] - w[2] * c1[1], w[2] * c1[0] - w[0] * c1[2], w[0] * c1[1] - w[1] * c1[0]};
  double v[3];
  for (int i = 0; i < 3; i++) v[i] = t[i] - 0.5 * c1[i] + kk * c2[i];
  nt = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  nr = th;
}

const char* enum2str(IterTermReason r) {
  static const char* n[] = {"Undefined", "NoPairings", "SolverError", "MaxIterations", "Stalled", "QualityCheckpointFailed",
                            "HookRequest"};
  return n[(int)r];
}

void throw_status(mh_status s, const char* where) {
  throw std::runtime_error(std::string(where) + ": " + mh_status_string(s) + ": " + mh_last_error_string());
}

// ================================================================== device handles
DeviceContext::DeviceContext(int device) : device_(device) { check(mh_ctx_create(device, nullptr, &ctx_), "mh_ctx_create"); }
void DeviceContext::synchronize() const { check(mh_ctx_synchronize(ctx_), "mh_ctx_synchronize"); }
DeviceContext::~DeviceContext() { mh_ctx_destroy(ctx_); }

std::shared_ptr<DeviceContext> DeviceContext::Default() {
  static std::mutex mtx;
  static std::weak_ptr<DeviceContext> weak;
  std::lock_guard<std::mutex> lk(mtx);
  auto sp = weak.lock();
  if (!sp) {
    sp = std::make_shared<DeviceContext>(0);
    weak = sp;
  }
  return sp;
}

HashedVoxelPointCloud::HashedVoxelPointCloud(float voxel_size, uint32_t max_points_per_voxel, std::shared_ptr<DeviceContext> ctx)
    : ctx_(std::move(ctx)) {
  mh_map_params p{};
  p.voxel_size = voxel_size;
  p.max_points_per_voxel = max_points_per_voxel;
  p.index_mode = molahip_host::plugin_switches().index_mode;          // MOLA_HIP_INDEX_MODE (default floor)
  p.far_voxel_metric = molahip_host::plugin_switches().far_voxel_metric;  // MOLA_HIP_FAR_VOXEL_METRIC (default Chebyshev)
  check(mh_map_create(ctx_->get(), &p, &map_), "mh_map_create");
}
HashedVoxelPointCloud::HashedVoxelPointCloud(const mh_map_params& p, std::shared_ptr<DeviceContext> ctx) : ctx_(std::move(ctx)) {
  check(mh_map_create(ctx_->get(), &p, &map_), "mh_map_create");
}

static mh_map_params ndt_params(float vs, uint32_t cap, float min_dist, float ratio) {
  mh_map_params p{};
  p.voxel_size = vs;
  p.max_points_per_voxel = cap;
  p.index_mode = molahip_host::plugin_switches().index_mode;
  p.far_voxel_metric = molahip_host::plugin_switches().far_voxel_metric;
  p.min_distance_between_points = min_dist;
  p.ndt_max_eigen_ratio = ratio;
  p.ndt_min_points = 4;
  return p;
}
NDT::NDT(float voxel_size, uint32_t max_points_per_voxel, float min_distance_between_points, float max_eigen_ratio_for_planes,
         std::shared_ptr<DeviceContext> ctx)
    : HashedVoxelPointCloud(ndt_params(voxel_size, max_points_per_voxel, min_distance_between_points, max_eigen_ratio_for_planes),
                            std::move(ctx)) {}
size_t NDT::planeCount() const {
  mh_map_info i;
  check(mh_map_get_info(handle(), &i), "mh_map_get_info");
  return i.n_planes;
}
HashedVoxelPointCloud::~HashedVoxelPointCloud() { mh_map_destroy(map_); }

void HashedVoxelPointCloud::setPoints(const float* x, const float* y, const float* z, size_t n) {
  check(mh_map_build(map_, x, y, z, n, MH_MEM_HOST), "mh_map_build");
}

void HashedVoxelPointCloud::insertPoints(const float* x, const float* y, const float* z, size_t n) {
  // insertPoint for each point after everything already stored: the device-resident incremental update
  DevicePointCloud pc(ctx_);
  pc.setPoints(x, y, z, n);
  insertPointCloud(pc, CPose3D(), 0.f);
}

void HashedVoxelPointCloud::insertPointCloud(const DevicePointCloud& pc, const CPose3D& robot_pose,
                                             float remove_voxels_farther_than) {
  const mh_status st = mh_map_insert(map_, pc.handle(), robot_pose.T, remove_voxels_farther_than);
  if (st == MH_WARN_PREVIOUS_OUT_OF_RANGE) {  // this key-frame IS in the map; the one before it lost its wild points: log, go on
    fprintf(stderr, "[molahip] warning: %s\n", mh_last_error_string());
    return;
  }
  check(st, "mh_map_insert");
}

void HashedVoxelPointCloud::clear() { check(mh_map_build(map_, nullptr, nullptr, nullptr, 0, MH_MEM_HOST), "mh_map_build"); }

DevicePointCloud::DevicePointCloud(std::shared_ptr<DeviceContext> ctx) : ctx_(std::move(ctx)) {
  check(mh_scan_create(ctx_->get(), nullptr, nullptr, nullptr, 0, MH_MEM_HOST, &scan_), "mh_scan_create");
}
DevicePointCloud::~DevicePointCloud() { mh_scan_destroy(scan_); }
size_t DevicePointCloud::size() const {
  uint64_t n = 0;
  check(mh_scan_size(scan_, &n), "mh_scan_size");
  return n;
}
void DevicePointCloud::setPoints(const float* x, const float* y, const float* z, size_t n) {
  check(mh_scan_update(scan_, x, y, z, n, MH_MEM_HOST), "mh_scan_update");
}
void DevicePointCloud::setPointsInterleaved(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                                            size_t off_z, long long off_t, bool pinned) {
  check(mh_scan_update_aos(scan_, data, n, point_step, off_x, off_y, off_z, (int64_t)off_t, pinned ? MH_MEM_HOST_PINNED : MH_MEM_HOST),
        "mh_scan_update_aos");
}
void DevicePointCloud::setTimestamps(const float* t, size_t n) {
  check(mh_scan_set_timestamps(scan_, t, n, MH_MEM_HOST), "mh_scan_set_timestamps");
}
void DevicePointCloud::boundingBox(float mn[3], float mx[3]) const {
  check(mh_scan_bbox(scan_, mn, mx, nullptr), "mh_scan_bbox");
}
void DevicePointCloud::download(std::vector<float>& x, std::vector<float>& y, std::vector<float>& z) const {
  const size_t n = size();
  x.resize(n); y.resize(n); z.resize(n);
  check(mh_scan_download(scan_, x.data(), y.data(), z.data(), nullptr, nullptr), "mh_scan_download");
}

size_t HashedVoxelPointCloud::size() const {
  mh_map_info i;
  check(mh_map_get_info(map_, &i), "mh_map_get_info");
  return i.n_points;
}
size_t HashedVoxelPointCloud::voxelCount() const {
  mh_map_info i;
  check(mh_map_get_info(map_, &i), "mh_map_get_info");
  return i.n_voxels;
}

// ================================================================== parameters
static uint32_t to_u32(const std::string& s) { return (uint32_t)strtoul(s.c_str(), nullptr, 10); }
static bool to_bool(const std::string& s) { return s == "true" || s == "True" || s == "1" || s == "yes"; }

void Parameters::load_from(const Config& c) {
  if (c.has("maxIterations")) maxIterations = to_u32(c["maxIterations"].asString());
  if (c.has("minAbsStep_trans")) minAbsStep_trans = strtod(c["minAbsStep_trans"].asString().c_str(), nullptr);
  if (c.has("minAbsStep_rot")) minAbsStep_rot = strtod(c["minAbsStep_rot"].asString().c_str(), nullptr);
  if (c.has("generateDebugFiles")) generateDebugFiles = to_bool(c["generateDebugFiles"].asString());
  if (c.has("debugFileNameFormat")) {
    debugFileNameFormat = c["debugFileNameFormat"].asString();
    const std::string ext = ".icplog";  // the JSON trace must not masquerade as an MRPT-serialised .icplog
    if (debugFileNameFormat.size() >= ext.size() && debugFileNameFormat.compare(debugFileNameFormat.size() - ext.size(), ext.size(), ext) == 0)
      debugFileNameFormat += ".json";
  }
}

// ================================================================== matcher
bool Matcher::match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose,
                    const MatchContext& mc, Pairings& out) const {
  if (!enabled) return false;
  if (runFromIteration != 0 && mc.icpIteration < runFromIteration) return false;
  if (runUpToIteration != 0 && mc.icpIteration > runUpToIteration) return false;
  impl_match(pcGlobal, pcLocal, localPose, mc, out);
  return true;
}

void Matcher_Points_DistanceThreshold::initialize(const Config& c) {
  parameterFromConfig(c, "threshold", &threshold, true);
  parameterFromConfig(c, "thresholdAngularDeg", &thresholdAngularDeg, false);
  if (c.has("pairingsPerPoint")) pairingsPerPoint = to_u32(c["pairingsPerPoint"].asString());
  if (c.has("allowMatchAlreadyMatchedGlobalPoints"))
    allowMatchAlreadyMatchedGlobalPoints = to_bool(c["allowMatchAlreadyMatchedGlobalPoints"].asString());
  if (c.has("runFromIteration")) runFromIteration = to_u32(c["runFromIteration"].asString());
  if (c.has("runUpToIteration")) runUpToIteration = to_u32(c["runUpToIteration"].asString());
  pointLayerMatches.clear();
  if (c.has("pointLayerMatches")) {
    const Config& s = c["pointLayerMatches"];
    for (size_t i = 0; i < s.size(); i++) {
      LayerMatch lm;
      lm.global = s.at(i)["global"].asString();
      lm.local = s.at(i)["local"].asString();
      lm.weight = strtod(s.at(i).getOr("weight", "1.0").c_str(), nullptr);
      pointLayerMatches.push_back(lm);
    }
  }
  if (pairingsPerPoint < 1 || pairingsPerPoint > MH_MAX_PAIRINGS_PER_POINT)
    throw std::runtime_error("Matcher_Points_DistanceThreshold: pairingsPerPoint must be 1.." +
                             std::to_string(MH_MAX_PAIRINGS_PER_POINT) + " (rgbd.yaml:138 uses 2)");
}

static const PointCloud& local_layer(const metric_map_t& m, const std::string& name) {
  auto it = m.layers.find(name);
  if (it == m.layers.end()) throw std::runtime_error("local layer '" + name + "' not found");
  auto p = std::dynamic_pointer_cast<PointCloud>(it->second);
  if (!p) throw std::runtime_error("local layer '" + name + "' is not a point cloud");
  return *p;
}
static const HashedVoxelPointCloud& global_layer(const metric_map_t& m, const std::string& name) {
  auto it = m.layers.find(name);
  if (it == m.layers.end()) throw std::runtime_error("global layer '" + name + "' not found");
  auto p = std::dynamic_pointer_cast<HashedVoxelPointCloud>(it->second);
  if (!p) throw std::runtime_error("global layer '" + name + "' is not NearestNeighborsCapable on the device");
  return *p;
}

void Matcher_Points_DistanceThreshold::impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal,
                                                  const CPose3D& localPose, const MatchContext&, Pairings& out) const {
  for (const auto& lm : pointLayerMatches) {
    const PointCloud& loc = local_layer(pcLocal, lm.local);
    const HashedVoxelPointCloud& glob = global_layer(pcGlobal, lm.global);
    const size_t n = loc.size();
    out.potential_pairings += n * pairingsPerPoint;
    if (out.pt2pt_weight_set && out.pt2pt_weight != lm.weight)
      throw std::runtime_error("pointLayerMatches with different weights in one pairing set: not a device input (one weight per kind of pair)");
    out.pt2pt_weight = lm.weight;
    out.pt2pt_weight_set = true;
    if (!n) continue;
    mh_scan* scan = nullptr;
    check(mh_scan_create(glob.context()->get(), loc.x.data(), loc.y.data(), loc.z.data(), n, MH_MEM_HOST, &scan), "mh_scan_create");
    const size_t cap = n * pairingsPerPoint;  // nn_multiple_search(k): up to k pairs per local point (rgbd.yaml:138)
    std::vector<uint32_t> li(cap), gi(cap);
    std::vector<float> gx(cap), gy(cap), gz(cap), d2(cap);
    mh_pairs_out po{li.data(), gi.data(), gx.data(), gy.data(), gz.data(), d2.data()};
    mh_match_info info{};
    const mh_status st = mh_nn_search_k(glob.handle(), scan, localPose.T, threshold, thresholdAngularDeg, pairingsPerPoint, &po,
                                        MH_MEM_HOST, &info);
    mh_scan_destroy(scan);
    check(st, "mh_nn_search_k");
    // U12 (MOLA_HIP_MATCHED_POINTS=skip): local points an earlier matcher of this iteration has paired are left out [U]
    const bool skip_paired = molahip_host::plugin_switches().matched_points == MH_MATCHED_POINTS_SKIP;
    auto& paired = out.local_paired[lm.local];
    if (paired.size() < n) paired.resize(n, 0);
    std::vector<uint8_t> paired_before;
    if (skip_paired) paired_before = paired;  // (a point's own second pair is not "already paired by an earlier matcher")
    for (size_t k = 0; k < info.n_pairs; k++) {
      if (skip_paired && paired_before[li[k]]) continue;
      paired[li[k]] = 1;
      out.localIdx.push_back(li[k]);
      out.globalIdx.push_back(gi[k]);
      out.lx.push_back(loc.x[li[k]]);
      out.ly.push_back(loc.y[li[k]]);
      out.lz.push_back(loc.z[li[k]]);
      out.gx.push_back(gx[k]);
      out.gy.push_back(gy[k]);
      out.gz.push_back(gz[k]);
      out.errSq.push_back(d2[k]);
    }
  }
}

void Matcher_Point2Plane::initialize(const Config& c) {
  parameterFromConfig(c, "distanceThreshold", &distanceThreshold, true);
  parameterFromConfig(c, "planeEigenThreshold", &planeEigenThreshold, false);
  parameterFromConfig(c, "searchRadius", &searchRadius, false);
  if (c.has("knn")) knn = to_u32(c["knn"].asString());
  if (c.has("minimumPlanePoints")) minimumPlanePoints = to_u32(c["minimumPlanePoints"].asString());
  if (c.has("allowMatchAlreadyMatchedGlobalPoints"))
    allowMatchAlreadyMatchedGlobalPoints = to_bool(c["allowMatchAlreadyMatchedGlobalPoints"].asString());
  if (c.has("runFromIteration")) runFromIteration = to_u32(c["runFromIteration"].asString());
  if (c.has("runUpToIteration")) runUpToIteration = to_u32(c["runUpToIteration"].asString());
  pointLayerMatches.clear();
  if (c.has("pointLayerMatches")) {
    const Config& s = c["pointLayerMatches"];
    for (size_t i = 0; i < s.size(); i++) {
      Matcher_Points_DistanceThreshold::LayerMatch lm;
      lm.global = s.at(i)["global"].asString();
      lm.local = s.at(i)["local"].asString();
      lm.weight = strtod(s.at(i).getOr("weight", "1.0").c_str(), nullptr);
      pointLayerMatches.push_back(lm);
    }
  }
}

static void append_pl_pairs(const PointCloud& loc, const std::vector<uint32_t>& li, const std::vector<float>* a, size_t k,
                            Pairings& out) {
  for (size_t i = 0; i < k; i++) {
    out.pl_lx.push_back(loc.x[li[i]]);
    out.pl_ly.push_back(loc.y[li[i]]);
    out.pl_lz.push_back(loc.z[li[i]]);
    out.pl_cx.push_back(a[0][i]);
    out.pl_cy.push_back(a[1][i]);
    out.pl_cz.push_back(a[2][i]);
    out.pl_nx.push_back(a[3][i]);
    out.pl_ny.push_back(a[4][i]);
    out.pl_nz.push_back(a[5][i]);
  }
}

void Matcher_Point2Plane::impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose,
                                     const MatchContext&, Pairings& out) const {
  for (const auto& lm : pointLayerMatches) {
    const PointCloud& loc = local_layer(pcLocal, lm.local);
    const HashedVoxelPointCloud& glob = global_layer(pcGlobal, lm.global);
    const size_t n = loc.size();
    out.potential_pairings += n;
    if (out.pt2pl_weight_set && out.pt2pl_weight != lm.weight)
      throw std::runtime_error("Matcher_Point2Plane layers with different weights in one pairing set: not a device input");
    out.pt2pl_weight = lm.weight;  // [U] whether upstream applies a layer weight to plane pairs is unverified; every shipped pipeline has 1.0
    out.pt2pl_weight_set = true;
    if (!n) continue;
    mh_scan* scan = nullptr;
    check(mh_scan_create(glob.context()->get(), loc.x.data(), loc.y.data(), loc.z.data(), n, MH_MEM_HOST, &scan), "mh_scan_create");
    std::vector<uint32_t> li(n);
    std::vector<float> a[6];
    for (auto& v : a) v.resize(n);
    mh_pairs_pl_out po{li.data(), a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), a[5].data()};
    mh_match_info info{};
    mh_status st;
    if (dynamic_cast<const NDT*>(&glob)) {  // per-voxel planes
      st = mh_nn_search_pt2pl(glob.handle(), scan, localPose.T, distanceThreshold, molahip_host::plugin_switches().pt2pl_mode, &po,
                              MH_MEM_HOST, &info);
    } else {  // a plain point layer: k nearest neighbours + PCA (rgbd.yaml:143-151)
      mh_pt2pl_knn_params kp{distanceThreshold, planeEigenThreshold, searchRadius, knn, minimumPlanePoints};
      st = mh_nn_search_pt2pl_knn(glob.handle(), scan, localPose.T, &kp, &po, MH_MEM_HOST, &info);
    }
    mh_scan_destroy(scan);
    check(st, "mh_nn_search_pt2pl");
    append_pl_pairs(loc, li, a, info.n_pairs, out);
    auto& paired = out.local_paired[lm.local];  // (what MatchState::localPairedBitField records upstream [U])
    if (paired.size() < n) paired.resize(n, 0);
    for (size_t k = 0; k < info.n_pairs; k++) paired[li[k]] = 1;
  }
}

// ================================================================== solver
static RobustKernel parse_kernel(std::string s) {
  const size_t p = s.rfind("::");
  if (p != std::string::npos) s = s.substr(p + 2);
  if (s == "None") return RobustKernel::None;
  if (s == "GemanMcClure") return RobustKernel::GemanMcClure;
  if (s == "Cauchy") return RobustKernel::Cauchy;
  if (s == "GemanMcClure_KISS") return RobustKernel::GemanMcClure_KISS;
  if (s == "GemanMcClure_Barron") return RobustKernel::GemanMcClure_Barron;
  if (s == "GemanMcClure_C2") return RobustKernel::GemanMcClure_C2;
  throw std::runtime_error("unknown robustKernel '" + s + "'");
}

void Solver_GaussNewton::initialize(const Config& c) {
  if (c.has("maxIterations")) maxIterations = to_u32(c["maxIterations"].asString());
  if (c.has("robustKernel")) robustKernel = parse_kernel(c["robustKernel"].asString());
  parameterFromConfig(c, "robustKernelParam", &robustKernelParam, false);
  if (c.has("minDelta")) minDelta = strtod(c["minDelta"].asString().c_str(), nullptr);
  if (c.has("maxCost")) maxCost = strtod(c["maxCost"].asString().c_str(), nullptr);
}

static mh_gn_params gn_params_of(const Solver_GaussNewton& s) {
  mh_gn_params p{};
  p.max_inner_iterations = s.maxIterations;
  p.robust_kernel = (uint32_t)s.robustKernel;
  p.robust_kernel_param = s.robustKernelParam;
  p.min_delta = s.minDelta;
  p.max_cost = s.maxCost;
  p.weight_pt2pt = 1.0;
  p.weight_pt2pl = 1.0;
  return p;
}

static void fill_prior(const std::optional<CPose3DPDFGaussianInf>& prior, mh_prior& out) {
  memcpy(out.mean, prior->mean.T, sizeof(out.mean));
  memcpy(out.info, prior->cov_inv, sizeof(out.info));
}

bool Solver_GaussNewton::optimal_pose(const Pairings& p, OptimalTF_Result& out, const SolverContext& sc) const {
  if (!sc.guessRelativePose) throw std::runtime_error("Solver_GaussNewton: guessRelativePose is required");
  mh_pairs_pt2pt pp{p.lx.data(), p.ly.data(), p.lz.data(), p.gx.data(), p.gy.data(), p.gz.data(), p.localIdx.size()};
  mh_pairs_pt2pl pl{p.pl_lx.data(), p.pl_ly.data(), p.pl_lz.data(), p.pl_cx.data(), p.pl_cy.data(), p.pl_cz.data(),
                    p.pl_nx.data(), p.pl_ny.data(), p.pl_nz.data(), p.pl_lx.size()};
  mh_gn_params gp = gn_params_of(*this);
  gp.weight_pt2pt = p.pt2pt_weight;  // the layer weight of the matcher that produced the pairs (Pairings::point_weights [U])
  gp.weight_pt2pl = p.pt2pl_weight;
  mh_prior pr;
  if (sc.prior) fill_prior(sc.prior, pr);
  double T[12];
  memcpy(T, sc.guessRelativePose->T, sizeof(T));
  int32_t n_steps = 0, ok = 1;
  check(mh_gn_solve(DeviceContext::Default()->get(), &pp, &pl, MH_MEM_HOST, &gp, sc.prior ? &pr : nullptr, T, &n_steps, &ok, nullptr),
        "mh_gn_solve");
  memcpy(out.optimalPose.T, T, sizeof(T));
  return ok != 0;
}

// ================================================================== factory
Matcher::Ptr create_matcher(const std::string& cn) {
  if (cn == "mp2p_icp::Matcher_Points_DistanceThreshold" || cn == "mp2p_icp_hip::Matcher_Points_DistanceThreshold")
    return std::make_shared<Matcher_Points_DistanceThreshold>();
  if (cn == "mp2p_icp::Matcher_Point2Plane" || cn == "mp2p_icp_hip::Matcher_Point2Plane")
    return std::make_shared<Matcher_Point2Plane>();
  throw std::runtime_error("matcher class '" + cn + "' is not available in mp2p_icp_hip");
}
Solver::Ptr create_solver(const std::string& cn) {
  if (cn == "mp2p_icp::Solver_GaussNewton" || cn == "mp2p_icp_hip::Solver_GaussNewton")
    return std::make_shared<Solver_GaussNewton>();
  throw std::runtime_error("solver class '" + cn + "' is not available in mp2p_icp_hip");
}

// ================================================================== ICP
ICP::ICP(std::shared_ptr<DeviceContext> ctx) : ctx_(std::move(ctx)) {}
ICP::~ICP() {
  if (scan_) mh_scan_destroy(scan_);
}

void ICP::setDeviceHook(double min_trans, double min_rot_rad, const CPose3D& checkpoint) {
  dev_hook_ = true;
  dev_hook_trans_ = min_trans;
  dev_hook_rot_ = min_rot_rad;
  dev_hook_chk_ = checkpoint;
}
void ICP::clearHooks() {
  dev_hook_ = false;
  iteration_hook_ = nullptr;
}

void ICP::attachToParameterSource(ParameterSource& s) {
  source_ = &s;
  for (auto& m : matchers_) m->attachToParameterSource(s);
  for (auto& v : solvers_) v->attachToParameterSource(s);
}

void ICP::initialize_matchers(const Config& seq) {
  matchers_.clear();
  for (size_t i = 0; i < seq.size(); i++) {
    auto m = create_matcher(seq.at(i)["class"].asString());
    m->initialize(seq.at(i)["params"]);
    m->attachToParameterSource(own_source_);
    matchers_.push_back(m);
  }
}
void ICP::initialize_solvers(const Config& seq) {
  solvers_.clear();
  for (size_t i = 0; i < seq.size(); i++) {
    auto s = create_solver(seq.at(i)["class"].asString());
    s->initialize(seq.at(i)["params"]);
    s->attachToParameterSource(own_source_);
    solvers_.push_back(s);
  }
}

// ICP::align updates "ICP_ITERATION" in the attached sources and re-realizes them every iteration [U]
void ICP::realize_iteration(uint32_t k) {
  std::map<std::string, double> vars = source_ ? source_->getVariableValues() : own_source_.getVariableValues();
  vars["ICP_ITERATION"] = (double)k;
  for (auto& m : matchers_) m->realizeWith(vars);
  for (auto& s : solvers_) s->realizeWith(vars);
}

// the two pipeline shapes the device loop implements: [Points_DistanceThreshold] (lidar3d-default.yaml:195-204) and
// [Point2Plane, Points_DistanceThreshold] on the same layers (lidar3d-ndt.yaml:195-210), with one Solver_GaussNewton
bool ICP::can_fuse() const {
  if (force_generic_ || (iteration_hook_ && !hook_replay_)) return false;
  if (matchers_.empty() || matchers_.size() > 2 || solvers_.size() != 1) return false;
  auto m = std::dynamic_pointer_cast<Matcher_Points_DistanceThreshold>(matchers_.back());
  auto s = std::dynamic_pointer_cast<Solver_GaussNewton>(solvers_[0]);
  if (!m || !s) return false;
  if (!(m->enabled && m->runFromIteration == 0 && m->runUpToIteration == 0 && m->pairingsPerPoint == 1 &&
        m->pointLayerMatches.size() == 1))  // (its weight: any -- the fused loop's solver takes it, round 5)
    return false;
  if (matchers_.size() == 2) {
    auto pl = std::dynamic_pointer_cast<Matcher_Point2Plane>(matchers_[0]);
    if (!pl || !pl->enabled || pl->runFromIteration || pl->runUpToIteration || pl->pointLayerMatches.size() != 1) return false;
    const auto &a = pl->pointLayerMatches[0], &b = m->pointLayerMatches[0];
    if (a.global != b.global || a.local != b.local) return false;
  }
  return true;
}

// ---------------------------------------------------------------- AlignBatcher
AlignBatcher::AlignBatcher(size_t participants) : active_(participants) {
  if (const char* e = getenv("MOLA_HIP_BATCH_SPLIT")) split_ = (size_t)std::max(1, atoi(e));
  solo_ = getenv("MOLA_HIP_BATCH_SOLO") != nullptr;  // every alignment on its own, as soon as it is asked for (A/B against the lock-step batches)
  no_solo_ = getenv("MOLA_HIP_BATCH_NO_SOLO") != nullptr;
}

// how many waiting requests make a batch: all active participants (split 1), or a share of them -- then two or more
// batches are in flight at once, each led by the thread that completed it, and nobody waits for the slowest alignment of
// ALL sequences (iteration counts differ by 5x from scan to scan), only for the slowest of its own batch
size_t AlignBatcher::threshold_locked() const {
  if (active_ < 4 || split_ <= 1) return active_;
  return std::max<size_t>(2, (active_ + split_ - 1) / split_);
}

// Everything that leaves waiting_ leaves it here, and leaves without its `lead` flag: a request whose flag is still set when its
// thread wakes is therefore still waiting (leave() used to swap the vector bare, a woken leader could then find it
// empty -- or holding NEWER requests -- and run a batch that did not contain itself).
void AlignBatcher::take_waiting_locked(std::vector<Request*>& batch) {
  batch.clear();
  batch.swap(waiting_);
  for (Request* r : batch) r->lead = false;
  in_flight_ += batch.size();
}

// a batch is due when enough requests wait -- or when everybody who is not waiting is inside a running batch already (then
// waiting longer only idles the device)
bool AlignBatcher::batch_due_locked() const {
  return !waiting_.empty() && (waiting_.size() >= threshold_locked() || waiting_.size() + in_flight_ >= active_);
}

void AlignBatcher::run_batch(std::vector<Request*>& batch) {
  if (batch.empty()) return;
  // the device work of this batch is issued by this thread alone; the mutex is NOT held (other participants queue up and
  // may start the next batch on their own contexts meanwhile: the C ABI is re-entrant across contexts)
  const size_t n = batch.size();
  const auto t_start = std::chrono::steady_clock::now();
  std::vector<const mh_map*> maps(n);
  std::vector<const mh_scan*> scans(n);
  std::vector<mh_icp_params> params(n);
  std::vector<double> T(12 * n);
  std::vector<const mh_prior*> priors(n);
  std::vector<mh_icp_result> results(n);
  bool any_prior = false;
  for (size_t i = 0; i < n; i++) {
    maps[i] = batch[i]->map;
    scans[i] = batch[i]->scan;
    params[i] = *batch[i]->params;
    memcpy(&T[12 * i], batch[i]->T, 12 * sizeof(double));
    priors[i] = batch[i]->prior;
    any_prior = any_prior || batch[i]->prior;
  }
  mh_status st = MH_OK;
  std::string err;
  if (n == 1) {
    st = mh_icp_align(maps[0], scans[0], &params[0], &T[0], priors[0], &results[0], nullptr, nullptr, MH_MEM_HOST);
  } else {
    st = mh_icp_align_batch(n, maps.data(), scans.data(), params.data(), 1, T.data(), any_prior ? priors.data() : nullptr,
                            results.data(), nullptr, MH_MEM_HOST);
  }
  if (st != MH_OK) err = mh_last_error_string();  // (thread-local in the library: read it on the thread that made the call)
  std::vector<mh_status> sts(n, st);
  std::vector<std::string> errs(n, err);
  if (st != MH_OK && n > 1) {
    // one job's bad argument or allocation failure must not fail the other sequences: once more, one by one, so that
    // every request gets its own status
    for (size_t i = 0; i < n; i++) {
      sts[i] = mh_icp_align(maps[i], scans[i], &params[i], &T[12 * i], priors[i], &results[i], nullptr, nullptr, MH_MEM_HOST);
      errs[i] = sts[i] != MH_OK ? mh_last_error_string() : "";
    }
  }
  std::lock_guard<std::mutex> lk(mtx_);
  n_batches_++;
  n_jobs_ += n;
  t_run_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
  t_assemble_ += std::chrono::duration<double>(t_start - batch[0]->arrived).count();
  in_flight_ -= n;
  if (batch_due_locked()) waiting_.front()->lead = true;  // (e.g. a participant left while this batch ran; woken by the notify below)
  for (size_t i = 0; i < n; i++) {
    *batch[i]->result = results[i];
    batch[i]->status = sts[i];
    batch[i]->error = errs[i];
    batch[i]->done = true;
  }
  cv_.notify_all();
}

mh_status AlignBatcher::align(const void* owner, const mh_map* map, const mh_scan* scan, const mh_icp_params* params,
                              const double T_guess[12], const mh_prior* prior, mh_icp_result* result, std::string* error) {
  Request rq;
  rq.map = map; rq.scan = scan; rq.params = params; rq.T = T_guess; rq.prior = prior; rq.result = result;
  // An alignment the library runs as ONE launch (the default pipeline's ICP layers: <= 2048 points) gains nothing from a
  // lock-step batch and loses the wait for the others: it is issued on its own at once, as long as the loops of all active
  // participants fit the device together (4 sequences: 4450 against 3620 scans/s; with 8, whose loops would take turns, the
  // batches win: 4950 against 4100).
  int32_t solo = solo_ ? 1 : 0;
  if (!solo && !no_solo_) {
    size_t callers;
    {
      std::lock_guard<std::mutex> g(mtx_);
      callers = active_;
    }
    if (mh_icp_align_prefers_solo(scan, params, (uint32_t)callers, &solo) != MH_OK) solo = 0;
  }
  std::unique_lock<std::mutex> lk(mtx_);
  rq.arrived = std::chrono::steady_clock::now();
  {
    OwnerState& os = owners_[owner ? owner : (const void*)scan];
    os.aligns++;  // this participant is past the filter set of its previous alignment count
    // A participant whose alignments run on their own is not in step with anybody: its prefetch worker's filter chain runs at
    // once instead of waiting for a set that only lock-step rounds complete (16 free-running sequences: 1.6 ms of every scan
    // were spent in `prefetch_wait`, 208 sets forced by the 2 ms net).
    static const bool filter_sets_always = getenv("MOLA_HIP_FILTER_SETS_ALWAYS") != nullptr;  // (A/B: the rendezvous also for solo callers)
    os.free_running = solo != 0 && !filter_sets_always;
    if (solo) last_solo_ = rq.arrived;
  }
  if (!pp_waiting_.empty()) cv_.notify_all();           // (a waiting filter worker may find its set complete now)
  if (solo) {
    in_flight_ += 1;
    // whoever waits for a batch waits for those who are not in flight: if that is nobody now, one of them leads it
    if (batch_due_locked()) {
      waiting_.front()->lead = true;
      cv_.notify_all();
    }
    std::vector<Request*> one{&rq};
    lk.unlock();
    run_batch(one);
    lk.lock();
    if (error) *error = rq.error;
    return rq.status;
  }
  waiting_.push_back(&rq);
  if (batch_due_locked()) {
    std::vector<Request*> batch;
    take_waiting_locked(batch);
    lk.unlock();
    run_batch(batch);
    lk.lock();
  } else {
    // Bounded: with three or more participants whose other alignments all run solo and never overlap, nobody's arrival makes
    // this request's batch due -- after the limit it leads whatever waits (a smaller batch: same results).
    static const auto limit_lockstep = std::chrono::microseconds([] {
      const char* e = getenv("MOLA_HIP_BATCH_WAIT_US");
      return e ? std::max(100, atoi(e)) : 3000;
    }());
    while (!rq.done) {
      // (while others' alignments run on their own -- one seen within the last few milliseconds -- nobody is coming to complete
      //  a round: this request leads what waits after a fraction of an alignment's duration)
      const bool free_run = (rq.arrived - last_solo_) < std::chrono::milliseconds(5);
      const auto limit = free_run ? std::chrono::microseconds(100) : limit_lockstep;
      const bool woke = cv_.wait_for(lk, limit, [&] { return rq.done || rq.lead; });
      if (rq.done) break;
      // `lead` set (or the limit reached) AND still among those waiting: whoever takes a batch clears the flags of its requests,
      // so a set flag means nobody has taken this one
      const bool still_waiting = std::find(waiting_.begin(), waiting_.end(), &rq) != waiting_.end();
      if (!still_waiting) continue;  // (taken by another leader meanwhile: its batch will set `done`)
      if (!woke) n_wait_timeouts_++;
      std::vector<Request*> batch;
      take_waiting_locked(batch);
      lk.unlock();
      run_batch(batch);
      lk.lock();
    }
  }
  if (error) *error = rq.error;
  return rq.status;
}

void AlignBatcher::run_filter_batch(std::vector<FilterRequest*>& batch) {
  const size_t n = batch.size();
  std::vector<const mh_scan*> raws(n);
  std::vector<mh_preprocess_params> params(n);
  std::vector<mh_scan*> maps(n), icps(n);
  for (size_t i = 0; i < n; i++) {
    raws[i] = batch[i]->raw;
    params[i] = *batch[i]->params;
    maps[i] = batch[i]->out_map;
    icps[i] = batch[i]->out_icp;
  }
  mh_status st = n == 1 ? mh_scan_preprocess(raws[0], &params[0], maps[0], icps[0])
                        : mh_scan_preprocess_batch(n, raws.data(), params.data(), sizeof(mh_preprocess_params), maps.data(), icps.data());
  std::vector<mh_status> sts(n, st);
  std::vector<std::string> errs(n, st != MH_OK ? mh_last_error_string() : "");
  if (st != MH_OK && n > 1)  // whose scan it was: one by one
    for (size_t i = 0; i < n; i++) {
      sts[i] = mh_scan_preprocess(raws[i], &params[i], maps[i], icps[i]);
      errs[i] = sts[i] != MH_OK ? mh_last_error_string() : "";
    }
  std::lock_guard<std::mutex> lk(mtx_);
  n_pp_batches_++;
  n_pp_jobs_ += n;
  for (size_t i = 0; i < n; i++) {
    batch[i]->status = sts[i];
    batch[i]->error = errs[i];
    batch[i]->done = true;
  }
  cv_.notify_all();
}

// complete = every active participant is past the set and every announced request of it (and of earlier sets) is here
bool AlignBatcher::filter_set_ready_locked(size_t set) const {
  if (set < pp_set_) return true;  // (its set went without it: a request that arrives late runs at once)
  for (const auto& kv : pp_pending_)
    if (kv.first <= set && kv.second > 0) return false;
  size_t past = 0;
  for (const auto& kv : owners_)
    past += (kv.second.free_running || kv.second.aligns > set || (kv.second.announced && kv.second.announced_set >= set)) ? 1 : 0;
  return past >= active_;
}

void AlignBatcher::take_filter_sets_upto(std::unique_lock<std::mutex>& lk, size_t set) {
  std::vector<FilterRequest*> batch, rest;
  for (auto* r : pp_waiting_) (r->set <= set ? batch : rest).push_back(r);
  pp_waiting_.swap(rest);
  if (set + 1 > pp_set_) pp_set_ = set + 1;
  for (auto* r : batch) r->taken = true;
  if (batch.empty()) return;
  lk.unlock();
  run_filter_batch(batch);
  lk.lock();
}

size_t AlignBatcher::announceFilter(const void* owner) {
  std::lock_guard<std::mutex> lk(mtx_);
  OwnerState& o = owners_[owner];
  if (o.free_running) return kNoFilterSet;  // (runs alone, at once: nobody waits for it, it waits for nobody)
  o.announced = true;
  o.announced_set = o.aligns;
  pp_pending_[o.aligns]++;
  return o.aligns;
}

void AlignBatcher::cancelAnnouncedFilter(const void* owner, size_t set) {
  (void)owner;
  if (set == kNoFilterSet) return;
  std::lock_guard<std::mutex> lk(mtx_);
  auto it = pp_pending_.find(set);
  if (it != pp_pending_.end() && it->second > 0 && --it->second == 0) pp_pending_.erase(it);
  cv_.notify_all();
}

mh_status AlignBatcher::preprocess(const void* owner, size_t set, const mh_scan* raw, const mh_preprocess_params* params,
                                   mh_scan* out_map, mh_scan* out_icp, std::string* error) {
  (void)owner;
  if (set == kNoFilterSet) {  // a free-running participant's request: its own launches, now
    const mh_status st = mh_scan_preprocess(raw, params, out_map, out_icp);
    if (st != MH_OK && error) *error = mh_last_error_string();
    std::lock_guard<std::mutex> lk(mtx_);
    n_pp_batches_++;
    n_pp_jobs_++;
    return st;
  }
  FilterRequest rq;
  rq.raw = raw; rq.params = params; rq.out_map = out_map; rq.out_icp = out_icp; rq.set = set;
  std::unique_lock<std::mutex> lk(mtx_);
  {
    auto it = pp_pending_.find(set);  // announced -> arrived
    if (it != pp_pending_.end() && it->second > 0 && --it->second == 0) pp_pending_.erase(it);
  }
  pp_waiting_.push_back(&rq);
  cv_.notify_all();  // (the arrival may be what another worker's set was waiting for)
  static const auto limit = std::chrono::microseconds([] {
    const char* e = getenv("MOLA_HIP_FILTER_SET_WAIT_US");
    return e ? std::max(0, atoi(e)) : 2000;
  }());
  while (!rq.done) {
    if (!rq.taken && filter_set_ready_locked(rq.set)) {
      take_filter_sets_upto(lk, rq.set);
      continue;
    }
    const bool woke = cv_.wait_for(lk, limit, [&] { return rq.done || (!rq.taken && filter_set_ready_locked(rq.set)); });
    if (woke || rq.taken) continue;
    n_pp_timeouts_++;  // nobody completed the set in time: everything up to this request's set runs now
    take_filter_sets_upto(lk, rq.set);
  }
  if (error) *error = rq.error;
  return rq.status;
}

void AlignBatcher::forgetOwner(const void* owner) {
  std::lock_guard<std::mutex> lk(mtx_);
  owners_.erase(owner);
  cv_.notify_all();
}

void AlignBatcher::leave() {
  std::unique_lock<std::mutex> lk(mtx_);
  if (active_ > 0) active_--;
  cv_.notify_all();  // (fewer participants: a waiting filter set may be complete now)
  if (batch_due_locked()) {
    // the others were only waiting for this one
    std::vector<Request*> batch;
    take_waiting_locked(batch);
    lk.unlock();
    run_batch(batch);
  }
}

void ICP::align(const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const TPose3D& guess, const Parameters& p,
                Results& result, const std::optional<CPose3DPDFGaussianInf>& prior) {
  result = Results();
  const CPose3D g(guess);
  if (can_fuse()) {
    auto m = std::static_pointer_cast<Matcher_Points_DistanceThreshold>(matchers_.back());
    last_fused_ = true;
    const std::string& lname = m->pointLayerMatches[0].local;
    auto it = pcLocal.layers.find(lname);
    auto dev = it != pcLocal.layers.end() ? std::dynamic_pointer_cast<DevicePointCloud>(it->second) : nullptr;
    align_fused(dev ? nullptr : &local_layer(pcLocal, lname), dev.get(), global_layer(pcGlobal, m->pointLayerMatches[0].global),
                g, p, result, prior);
  } else {
    last_fused_ = false;
    align_generic(pcLocal, pcGlobal, g, p, result, prior);
  }
}

// the per-align trace file of Parameters::generateDebugFiles (see the header)
static void write_debug_file(const Parameters& p, const CPose3D& guess, const mh_icp_result& r, const std::vector<mh_icp_iter>& trace,
                             size_t n_local) {
  static std::atomic<uint64_t> counter{0};
  std::string path = p.debugFileNameFormat;
  const std::string id = std::to_string(counter++);
  for (size_t pos; (pos = path.find("$UNIQUE_ID")) != std::string::npos;) path.replace(pos, 10, id);
  for (const char* var : {"$LOCAL_ID", "$LOCAL_LABEL", "$GLOBAL_ID", "$GLOBAL_LABEL"})
    for (size_t pos; (pos = path.find(var)) != std::string::npos;) path.erase(pos, strlen(var));
  const size_t slash = path.find_last_of('/');
  if (slash != std::string::npos) {
    std::string dir;
    for (size_t i = 0; i <= slash; i++) {  // mkdir -p
      dir += path[i];
      if (path[i] == '/') (void)mkdir(dir.c_str(), 0755);
    }
  }
  FILE* f = fopen(path.c_str(), "w");
  if (!f) throw std::runtime_error("generateDebugFiles: cannot write " + path);
  auto pose = [&](const double* T) {
    fprintf(f, "[");
    for (int i = 0; i < 12; i++) fprintf(f, "%s%.17g", i ? ", " : "", T[i]);
    fprintf(f, "]");
  };
  fprintf(f, "{\n \"format\": \"molahip-icplog-json-1\", \"n_local_points\": %zu, \"max_iterations\": %u,\n \"initial_guess\": ", n_local, p.maxIterations);
  pose(guess.T);
  fprintf(f, ",\n \"iterations\": [\n");
  const uint32_t cnt = std::min<uint32_t>(p.maxIterations, r.n_iterations + 1);
  for (uint32_t k = 0; k < cnt; k++) {
    fprintf(f, "  {\"iteration\": %u, \"n_pairs\": %u, \"threshold\": %.17g, \"kernel_param\": %.17g, \"delta_trans\": %.17g, \"delta_rot\": %.17g, \"pose\": ",
            k, trace[k].n_pairs, trace[k].threshold, trace[k].kernel_param, trace[k].delta_trans, trace[k].delta_rot);
    pose(trace[k].T);
    fprintf(f, "}%s\n", k + 1 < cnt ? "," : "");
  }
  fprintf(f, " ],\n \"n_iterations\": %u, \"termination\": \"%s\", \"quality\": %.17g, \"n_final_pairs\": %u,\n \"final_pose\": ",
          r.n_iterations, enum2str((IterTermReason)r.termination_reason), r.quality, r.n_final_pairs);
  pose(r.T);
  fprintf(f, "\n}\n");
  fclose(f);
}

// The thresholds are functions of ICP_ITERATION only once the caller's variables are fixed (LidarOdometry.cpp:1571-1635
// publishes them before align): all iterations evaluated up front.  The result is kept with the VALUES of the variables
// the formulas read, so that a caller who knows them early (the odometry driver, while the device is busy with the
// key-frame update of the previous scan) can have the work done before align() needs it: 300 iterations x 2-3 formulas
// were 40 us of a 1 ms scan with the device idle.
void ICP::prepare_schedule(uint32_t n_iterations) {
  auto m = std::static_pointer_cast<Matcher_Points_DistanceThreshold>(matchers_.back());
  auto s = std::static_pointer_cast<Solver_GaussNewton>(solvers_[0]);
  auto mpl = matchers_.size() == 2 ? std::static_pointer_cast<Matcher_Point2Plane>(matchers_[0]) : nullptr;
  // formulas compiled once and bound to one copy of the variables; only ICP_ITERATION is swept (in place)
  std::map<std::string, double> vars = source_ ? source_->getVariableValues() : own_source_.getVariableValues();
  double& it_var = vars["ICP_ITERATION"];
  it_var = 0.0;
  const auto bm = m->bind(vars);
  const auto bs = s->bind(vars);
  Parameterizable::Binding bp;
  if (mpl) bp = mpl->bind(vars);
  std::vector<double> key;
  for (const Parameterizable::Binding* b : {&bm, &bs, (const Parameterizable::Binding*)&bp})
    for (const auto& item : b->items)
      for (const double* v : item.second) key.push_back(*v);
  if (sched_.valid && sched_.key == key && sched_.thr.size() >= n_iterations) return;
  sched_.valid = false;
  sched_.key = key;
  sched_.thr.assign(n_iterations, 0.0);
  sched_.kp.assign(n_iterations, 0.0);
  sched_.plthr.assign(n_iterations, 0.0);
  for (uint32_t k = 0; k < n_iterations; k++) {
    it_var = (double)k;
    bm.realize();
    bs.realize();
    bp.realize();
    sched_.thr[k] = m->threshold;
    sched_.kp[k] = s->robustKernelParam;
    if (mpl) sched_.plthr[k] = mpl->distanceThreshold;
  }
  sched_.valid = true;
}

void ICP::precomputeSchedule(uint32_t n_iterations) {
  if (!can_fuse() || !n_iterations) return;
  prepare_schedule(n_iterations);
  realize_iteration(0);  // (the members the formulas write are left as align() leaves them)
}

void ICP::align_fused(const PointCloud* host_local, const DevicePointCloud* dev_local, const HashedVoxelPointCloud& global,
                      const CPose3D& guess, const Parameters& p, Results& result,
                      const std::optional<CPose3DPDFGaussianInf>& prior) {
  auto m = std::static_pointer_cast<Matcher_Points_DistanceThreshold>(matchers_.back());
  auto s = std::static_pointer_cast<Solver_GaussNewton>(solvers_[0]);
  auto mpl = matchers_.size() == 2 ? std::static_pointer_cast<Matcher_Point2Plane>(matchers_[0]) : nullptr;
  // the thresholds are functions of ICP_ITERATION only once the caller's variables are fixed for this call
  // (LidarOdometry.cpp:1571-1635 publishes them before align): evaluate them for every iteration up front
  const auto t_setup0 = std::chrono::steady_clock::now();
  prepare_schedule(p.maxIterations);  // (a no-op when precomputeSchedule() ran on the same values of the variables)
  const std::vector<double>&thr = sched_.thr, &kp = sched_.kp, &plthr = sched_.plthr;
  if (p.maxIterations) realize_iteration(0);
  mh_icp_params ip{};
  ip.max_iterations = p.maxIterations;
  ip.min_abs_step_trans = p.minAbsStep_trans;
  ip.min_abs_step_rot = p.minAbsStep_rot;
  ip.threshold = thr.data();
  ip.kernel_param = kp.data();
  ip.threshold_angular_deg = m->thresholdAngularDeg;
  ip.pt2pl_threshold = mpl ? plthr.data() : nullptr;
  ip.gn = gn_params_of(*s);
  ip.gn.weight_pt2pt = m->pointLayerMatches[0].weight;  // pointLayerMatches {..., weight} (yaml:203-204)
  if (mpl) ip.gn.weight_pt2pl = mpl->pointLayerMatches[0].weight;
  ip.hook_enabled = dev_hook_ ? 1u : 0u;
  ip.hook_min_trans = dev_hook_trans_;
  ip.hook_min_rot = dev_hook_rot_;
  memcpy(ip.hook_checkpoint, dev_hook_chk_.T, sizeof(ip.hook_checkpoint));
  ip.compute_covariance = 1;
  ip.cov_findif_xyz = 1e-7;
  ip.cov_findif_ang = 1e-7;
  molahip_host::apply_switches(ip, molahip_host::plugin_switches());  // MOLA_HIP_* overrides, if any are set
  if (p.maxIterations > full_budget_) full_budget_ = p.maxIterations;
  const int call_kind = p.maxIterations < full_budget_ ? 1 : 0;
  ip.expected_iterations = last_iterations_[call_kind];
  mh_scan* scan = nullptr;
  PointCloud downloaded;  // only when a device layer's final pairings are requested
  if (dev_local) {
    scan = dev_local->handle();  // already resident: no upload
  } else {
    const PointCloud& hl = *host_local;
    if (scan_ && scan_ctx_ != global.context()) {  // the staging layer lives in the map's context: follow the map
      mh_scan_destroy(scan_);
      scan_ = nullptr;
    }
    if (!scan_) {
      scan_ctx_ = global.context();  // (kept alive: the layer must not outlive its context)
      check(mh_scan_create(scan_ctx_->get(), hl.x.data(), hl.y.data(), hl.z.data(), hl.size(), MH_MEM_HOST, &scan_),
            "mh_scan_create");
    } else
      check(mh_scan_update(scan_, hl.x.data(), hl.y.data(), hl.z.data(), hl.size(), MH_MEM_HOST), "mh_scan_update");
    scan = scan_;
  }
  mh_prior pr;
  if (prior) fill_prior(prior, pr);
  last_setup_seconds_ = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_setup0).count();
  mh_icp_result r{};
  const size_t n = dev_local ? dev_local->size() : host_local->size();
  const bool want_pairs = keep_pairings_;
  std::vector<uint32_t> li(want_pairs ? n : 0), gi(li.size());
  std::vector<float> gx(li.size()), gy(li.size()), gz(li.size()), d2(li.size());
  mh_pairs_out po{li.data(), gi.data(), gx.data(), gy.data(), gz.data(), d2.data()};
  std::vector<mh_icp_iter> trace(p.generateDebugFiles ? p.maxIterations : 0);
  if (iteration_hook_) {
    // an opaque host hook on the fused loop: replay it on the traced poses (molahip_host/hook_replay.h)
    ip.hook_enabled = 0;
    auto run = [&](uint32_t budget, mh_icp_iter* tr) {
      mh_icp_params q = ip;
      q.max_iterations = budget;
      mh_icp_result rr{};
      check(mh_icp_align(global.handle(), scan, &q, guess.T, prior ? &pr : nullptr, &rr, tr, want_pairs ? &po : nullptr,
                         MH_MEM_HOST), "mh_icp_align");
      if (tr && !trace.empty()) std::copy(tr, tr + budget, trace.begin());
      return rr;
    };
    auto hook = [&](uint32_t k, const double* T) {
      OptimalTF_Result cur;
      memcpy(cur.optimalPose.T, T, sizeof(cur.optimalPose.T));
      IterationHook_Input in;
      in.currentIteration = k;
      in.currentSolution = &cur;
      return iteration_hook_(in).request_stop;
    };
    r = molahip_host::align_with_replayed_hook(p.maxIterations, run, hook);
  } else if (batcher_ && trace.empty() && !want_pairs) {
    // several sequences in one process: this alignment joins the others' (AlignBatcher)
    std::string err;
    const mh_status st = batcher_->align(batch_owner_, global.handle(), scan, &ip, guess.T, prior ? &pr : nullptr, &r, &err);
    if (st != MH_OK) throw std::runtime_error(std::string("mh_icp_align_batch: ") + mh_status_string(st) + ": " + err);
  } else {
    check(mh_icp_align(global.handle(), scan, &ip, guess.T, prior ? &pr : nullptr, &r, trace.empty() ? nullptr : trace.data(),
                       want_pairs ? &po : nullptr, MH_MEM_HOST), "mh_icp_align");
  }
  if (p.generateDebugFiles) write_debug_file(p, guess, r, trace, n);
  last_iterations_[call_kind] = r.n_iterations + (r.termination_reason == MH_TERM_MAX_ITERATIONS ? 0u : 1u);
  last_polls_ = r.n_host_polls;
  last_enqueued_ = r.n_enqueued_iterations;
  memcpy(result.optimal_tf.mean.T, r.T, sizeof(r.T));
  memcpy(result.optimal_tf.cov, r.cov, sizeof(r.cov));
  result.quality = r.quality;
  result.nIterations = r.n_iterations;
  result.terminationReason = (IterTermReason)r.termination_reason;
  Pairings& fp = result.finalPairings;
  fp.potential_pairings = r.potential_pairings;
  if (!want_pairs) return;
  if (dev_local) dev_local->download(downloaded.x, downloaded.y, downloaded.z);
  const PointCloud& local = dev_local ? downloaded : *host_local;
  if (r.n_final_pairs_pt2pl) {
    std::vector<uint32_t> qli(n);
    std::vector<float> a[6];
    for (auto& v : a) v.resize(n);
    mh_pairs_pl_out qo{qli.data(), a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), a[5].data()};
    uint64_t nq = 0;
    check(mh_icp_get_pt2pl_pairs(scan, &qo, MH_MEM_HOST, &nq), "mh_icp_get_pt2pl_pairs");
    append_pl_pairs(local, qli, a, nq, fp);
  }
  for (uint32_t k = 0; k < r.n_final_pairs - r.n_final_pairs_pt2pl; k++) {
    fp.localIdx.push_back(li[k]);
    fp.globalIdx.push_back(gi[k]);
    fp.lx.push_back(local.x[li[k]]);
    fp.ly.push_back(local.y[li[k]]);
    fp.lz.push_back(local.z[li[k]]);
    fp.gx.push_back(gx[k]);
    fp.gy.push_back(gy[k]);
    fp.gz.push_back(gz[k]);
    fp.errSq.push_back(d2[k]);
  }
}

// matcher/solver-granular loop: the structure of mp2p_icp::ICP::align [U] with every numeric step on
// the device; used for custom pipelines and arbitrary host iteration hooks
void ICP::align_generic(const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const CPose3D& guess, const Parameters& p,
                        Results& result, const std::optional<CPose3DPDFGaussianInf>& prior) {
  OptimalTF_Result cur;
  cur.optimalPose = guess;
  CPose3D prev = guess;
  Pairings pairings;
  for (result.nIterations = 0; result.nIterations < p.maxIterations; result.nIterations++) {
    const uint32_t k = (uint32_t)result.nIterations;
    realize_iteration(k);
    MatchContext mc;
    mc.icpIteration = k;
    pairings = Pairings();
    for (auto& m : matchers_) m->match(pcGlobal, pcLocal, cur.optimalPose, mc, pairings);
    if (pairings.empty()) {
      result.terminationReason = IterTermReason::NoPairings;
      break;
    }
    SolverContext sc;
    sc.guessRelativePose = cur.optimalPose;
    sc.prior = prior;
    sc.icpIteration = k;
    bool ok = false;
    for (auto& s : solvers_) {
      ok = s->optimal_pose(pairings, cur, sc);
      if (ok) break;
    }
    if (!ok) {
      result.terminationReason = IterTermReason::SolverError;
      break;
    }
    // stall test on log_SE3(prev^-1 (+) cur): |V^-1 t| and |w| (lidar3d-default.yaml:174-175)
    double xi_t, xi_r;
    se3_log_norms(cur.optimalPose - prev, xi_t, xi_r);
    if (xi_t < p.minAbsStep_trans && xi_r < p.minAbsStep_rot) {
      result.terminationReason = IterTermReason::Stalled;
      break;
    }
    if (iteration_hook_) {
      IterationHook_Input hi;
      hi.currentIteration = k;
      hi.currentSolution = &cur;
      if (iteration_hook_(hi).request_stop) {
        result.terminationReason = IterTermReason::HookRequest;
        break;
      }
    } else if (dev_hook_) {
      const CPose3D hd = cur.optimalPose - dev_hook_chk_;
      if (hd.translationNorm() > dev_hook_trans_ || hd.rotationAngle() > dev_hook_rot_) {
        result.terminationReason = IterTermReason::HookRequest;
        break;
      }
    }
    prev = cur.optimalPose;
  }
  if (result.nIterations >= p.maxIterations) result.terminationReason = IterTermReason::MaxIterations;
  result.quality = pairings.empty() ? 0.0 : quality_.evaluate(pairings);
  result.optimal_tf.mean = cur.optimalPose;
  mh_pairs_pt2pt pp{pairings.lx.data(), pairings.ly.data(), pairings.lz.data(), pairings.gx.data(), pairings.gy.data(),
                    pairings.gz.data(), pairings.localIdx.size()};
  mh_pairs_pt2pl pl{pairings.pl_lx.data(), pairings.pl_ly.data(), pairings.pl_lz.data(), pairings.pl_cx.data(),
                    pairings.pl_cy.data(), pairings.pl_cz.data(), pairings.pl_nx.data(), pairings.pl_ny.data(),
                    pairings.pl_nz.data(), pairings.pl_lx.size()};
  if (!ctx_) ctx_ = DeviceContext::Default();
  check(mh_covariance(ctx_->get(), &pp, &pl, MH_MEM_HOST, cur.optimalPose.T, 1e-7, 1e-7, result.optimal_tf.cov), "mh_covariance");
  result.finalPairings = std::move(pairings);
}

std::tuple<ICP::Ptr, Parameters> icp_pipeline_from_yaml(const Config& c, std::shared_ptr<DeviceContext> ctx) {
  const std::string cn = c.getOr("class_name", "mp2p_icp::ICP");
  if (cn != "mp2p_icp::ICP" && cn != "mp2p_icp_hip::ICP") throw std::runtime_error("ICP class '" + cn + "' is not available");
  auto icp = std::make_shared<ICP>(std::move(ctx));
  Parameters p;
  if (c.has("params")) p.load_from(c["params"]);
  icp->initialize_solvers(c["solvers"]);
  icp->initialize_matchers(c["matchers"]);
  if (c.has("quality")) {
    const Config& q = c["quality"];
    for (size_t i = 0; i < q.size(); i++) {
      const std::string qc = q.at(i)["class"].asString();
      if (qc != "mp2p_icp::QualityEvaluator_PairedRatio" && qc != "mp2p_icp_hip::QualityEvaluator_PairedRatio")
        throw std::runtime_error("quality evaluator '" + qc + "' is not available");
    }
  }
  return {icp, p};
}

void reload_plugin_switches() { molahip_host::reload_plugin_switches(); }
uint32_t plugin_switch_matched_points() { return molahip_host::plugin_switches().matched_points; }

}  // namespace mp2p_icp_hip
