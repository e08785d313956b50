// lidar_odometry.cpp -- see mola_lidar_odometry_hip/LidarOdometry.h.  Control logic only: the arithmetic on points
// is behind include/molahip.h.  Citations are module/src/LidarOdometry.cpp unless another file is named.
#include "mola_lidar_odometry_hip/LidarOdometry.h"
#include "molahip_host/plugin_switches.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <stdexcept>

namespace mola_hip {

using namespace mp2p_icp_hip;

namespace {

double to_double(const std::string& s) { return strtod(s.c_str(), nullptr); }
bool to_bool(const std::string& s) { return s == "true" || s == "True" || s == "1" || s == "yes"; }
constexpr double kDeg2Rad = M_PI / 180.0;

// "$f{expr}" (evaluated once, at object creation) or a plain number / expression
double eval_now(std::string s, const std::map<std::string, double>& vars) {
  if (s.size() > 4 && s.compare(0, 3, "$f{") == 0 && s.back() == '}') s = s.substr(3, s.size() - 4);
  return evaluate_expression(s, vars);
}

struct StageTimer {  // adds the wall time of its scope to a profile entry
  std::map<std::string, double>& prof;
  const char* name;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  StageTimer(std::map<std::string, double>& p, const char* n) : prof(p), name(n) {}
  ~StageTimer() { prof[name] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

std::string class_of(const Config& entry) { return entry["class_name"].asString(); }
bool ends_with(const std::string& s, const std::string& suffix) {
  return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

}  // namespace

// ================================================================== motion model
void NavStateFuse::initialize(const Config& c) {
  auto num = [&](const char* k, double& v) { if (c.has(k)) v = to_double(c[k].asString()); };
  num("max_time_to_use_velocity_model", max_time_to_use_velocity_model);
  num("sigma_random_walk_acceleration_linear", sigma_random_walk_acceleration_linear);
  num("sigma_random_walk_acceleration_angular", sigma_random_walk_acceleration_angular);
  if (c.has("motion_model_prior")) motion_model_prior = to_bool(c["motion_model_prior"].asString());
  initial_twist.reset();
  if (c.has("initial_twist") && c["initial_twist"].size() == 6) {
    double v[6];
    bool any = false;
    for (size_t i = 0; i < 6; i++) {
      v[i] = to_double(c["initial_twist"].at(i).asString());
      any = any || v[i] != 0.0;
    }
    if (any) {
      Twist tw;
      tw.vx = v[0]; tw.vy = v[1]; tw.vz = v[2]; tw.wx = v[3]; tw.wy = v[4]; tw.wz = v[5];
      initial_twist = tw;
    }
  }
  reset();
}
void NavStateFuse::reset() {
  last_pose_.reset();
  twist_.reset();
  last_t_ = 0;
  for (double& v : last_cov_) v = 0;
}
void NavStateFuse::fuse_pose(double t, const CPose3D& pose, const double* cov) {
  if (last_pose_) {
    const double dt = t - last_t_;
    if (dt > 0 && dt <= max_time_to_use_velocity_model) {
      const CPose3D incr = pose - *last_pose_;  // increment in the frame of the previous pose
      double w[3];
      incr.so3Log(w);
      Twist tw;
      tw.vx = incr.T[3] / dt; tw.vy = incr.T[7] / dt; tw.vz = incr.T[11] / dt;
      tw.wx = w[0] / dt; tw.wy = w[1] / dt; tw.wz = w[2] / dt;
      twist_ = tw;
    } else {
      twist_.reset();  // a gap or a stamp that does not advance: no velocity from this pair
    }
  } else if (initial_twist) {
    twist_ = initial_twist;
  }
  last_pose_ = pose;
  for (int i = 0; i < 36; i++) last_cov_[i] = cov ? cov[i] : ((i % 7 == 0) ? 1e-12 : 0.0);  // (:834-836: "cannot be zero")
  last_t_ = t;
}

namespace {
// inverse of a symmetric positive-definite 6x6 (Cholesky); false when it is not
bool spd_inverse6(const double* A, double* Ainv) {
  double L[36] = {0};
  for (int i = 0; i < 6; i++)
    for (int j = 0; j <= i; j++) {
      double s = A[i * 6 + j];
      for (int k = 0; k < j; k++) s -= L[i * 6 + k] * L[j * 6 + k];
      if (i == j) {
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        L[i * 6 + i] = std::sqrt(s);
      } else {
        L[i * 6 + j] = s / L[j * 6 + j];
      }
    }
  double Li[36] = {0};  // inverse of the lower-triangular factor
  for (int c = 0; c < 6; c++) {
    Li[c * 6 + c] = 1.0 / L[c * 6 + c];
    for (int r = c + 1; r < 6; r++) {
      double s = 0;
      for (int k = c; k < r; k++) s -= L[r * 6 + k] * Li[k * 6 + c];
      Li[r * 6 + c] = s / L[r * 6 + r];
    }
  }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double s = 0;
      for (int k = 0; k < 6; k++) s += Li[k * 6 + i] * Li[k * 6 + j];
      Ainv[i * 6 + j] = s;
    }
  return true;
}
}  // namespace

std::optional<NavStateFuse::NavState> NavStateFuse::estimated_navstate(double t) const {
  if (!last_pose_ || !twist_) return std::nullopt;
  const double dt = t - last_t_;
  if (dt < 0 || dt > max_time_to_use_velocity_model) return std::nullopt;
  const double w[3] = {twist_->wx * dt, twist_->wy * dt, twist_->wz * dt};
  const double v[3] = {twist_->vx * dt, twist_->vy * dt, twist_->vz * dt};
  NavState ns;
  ns.pose.mean = *last_pose_ + CPose3D::FromRotVecAndTranslation(w, v);
  ns.twist = *twist_;
  if (motion_model_prior) {
    // covariance of the last pose, grown by the random-walk acceleration over dt, in the solver's tangent order:
    // (x,y,z,yaw,pitch,roll) -> [v; w] with w = (roll, pitch, yaw) to first order
    static const int perm[6] = {0, 1, 2, 5, 4, 3};
    double C[36];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) C[i * 6 + j] = last_cov_[perm[i] * 6 + perm[j]];
    const double sl = sigma_random_walk_acceleration_linear * dt, sa = sigma_random_walk_acceleration_angular * dt;
    for (int i = 0; i < 3; i++) C[i * 7] += sl * sl;
    for (int i = 3; i < 6; i++) C[i * 7] += sa * sa;
    double Ci[36];
    if (spd_inverse6(C, Ci))
      for (int i = 0; i < 36; i++) ns.pose.cov_inv[i] = Ci[i];
  }
  return ns;
}

// ================================================================== key-frame list
std::pair<bool, CPose3D> SearchablePoseList::check(const CPose3D& p) const {
  if (poses_.empty()) return {true, CPose3D()};
  size_t best = 0;
  double best_d2 = INFINITY;
  for (size_t i = 0; i < poses_.size(); i++) {
    const double dx = poses_[i].T[3] - p.T[3], dy = poses_[i].T[7] - p.T[7], dz = poses_[i].T[11] - p.T[11];
    const double d2 = dx * dx + dy * dy + dz * dz;
    if (d2 < best_d2) {
      best_d2 = d2;
      best = i;
    }
  }
  return {false, p - poses_[best]};
}
void SearchablePoseList::removeAllFartherThan(const CPose3D& p, double max_dist) {
  std::vector<CPose3D> kept;
  for (const auto& q : poses_) {
    const double dx = q.T[3] - p.T[3], dy = q.T[7] - p.T[7], dz = q.T[11] - p.T[11];
    if (std::sqrt(dx * dx + dy * dy + dz * dz) <= max_dist) kept.push_back(q);
  }
  poses_.swap(kept);
}

// ================================================================== parameters
void LidarOdometry::Params::load_from(const Config& c) {
  auto num = [&](const Config& n, const char* k, double& v) { if (n.has(k)) v = to_double(n[k].asString()); };
  auto flag = [&](const Config& n, const char* k, bool& v) { if (n.has(k)) v = to_bool(n[k].asString()); };
  num(c, "min_time_between_scans", min_time_between_scans);
  num(c, "max_sensor_range_filter_coefficient", max_sensor_range_filter_coefficient);
  num(c, "absolute_minimum_sensor_range", absolute_minimum_sensor_range);
  flag(c, "optimize_twist", optimize_twist);
  num(c, "optimize_twist_rerun_min_trans", optimize_twist_rerun_min_trans);
  num(c, "optimize_twist_rerun_min_rot_deg", optimize_twist_rerun_min_rot_deg);
  if (c.has("optimize_twist_max_corrections"))
    optimize_twist_max_corrections = (size_t)to_double(c["optimize_twist_max_corrections"].asString());
  num(c, "min_icp_goodness", min_icp_goodness);
  if (c.has("local_map_updates")) {
    const Config& l = c["local_map_updates"];
    flag(l, "enabled", local_map_updates_enabled);
    // DECLARE_PARAMETER_IN_REQ: formulas over wx,wy,wz and ESTIMATED_SENSOR_MAX_RANGE (yaml:44-46)
    parameterFromConfig(l, "min_translation_between_keyframes", &min_translation_between_keyframes, true);
    parameterFromConfig(l, "min_rotation_between_keyframes", &min_rotation_between_keyframes, true);
    parameterFromConfig(l, "max_distance_to_keep_keyframes", &max_distance_to_keep_keyframes, false);
    if (l.has("check_for_removal_every_n")) check_for_removal_every_n = (uint32_t)to_double(l["check_for_removal_every_n"].asString());
  }
  if (c.has("adaptive_threshold")) {
    const Config& a = c["adaptive_threshold"];
    flag(a, "enabled", adaptive_threshold_enabled);
    num(a, "initial_sigma", initial_sigma);
    num(a, "min_motion", min_motion);
    num(a, "maximum_sigma", maximum_sigma);
    num(a, "kp", kp);
    num(a, "alpha", alpha);
  }
  if (c.has("observation_validity_checks")) {
    const Config& v = c["observation_validity_checks"];
    flag(v, "enabled", validity_check_enabled);
    if (v.has("minimum_point_count")) validity_minimum_point_count = (uint32_t)to_double(v["minimum_point_count"].asString());
  }
}

// The observation filter chain the device implements, recognised from the pipeline file:
//   1st pass  Decimate(raw -> A) -> [ByRange(A -> B)] -> [BoundingBox(B -> C)] -> Decimate(C -> D)      (yaml:278-319)
//   2nd pass  [DeleteLayer] -> Deskew(C -> for_map) -> Deskew(D -> for_icp)                               (yaml:322-350)
//   merge     FilterMerge(for_map -> local map layer)                                                     (yaml:362-368)
struct LidarOdometry::FilterPlan : public Parameterizable {
  double decim_map_res = 0, decim_icp_res = 0, range_min = 0, range_max = 0;
  double bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};
  double time_offset = 0;
  uint32_t min_points_to_filter = 0;
  int32_t decim_map_method = MH_DECIMATE_FIRST_POINT, decim_icp_method = MH_DECIMATE_FIRST_POINT;
  int32_t bbox_mode = MH_BBOX_OFF, timestamp_method = MH_TS_NONE;
  bool skip_deskew = false;
  std::string layer_for_map, layer_for_icp, map_layer;

  static void unsupported(const std::string& what) {
    throw std::runtime_error("LidarOdometry (HIP): unsupported observation filter chain: " + what +
                             ". Implemented on the device: FilterAdjustTimestamps; FilterDecimateVoxels(FirstPoint | ClosestToAverage) -> "
                             "[FilterByRange] -> [FilterBoundingBox] -> FilterDecimateVoxels(FirstPoint | ClosestToAverage); FilterDeskew x2; "
                             "FilterMerge (the chain of pipelines/lidar3d-default.yaml)");
  }
  void decimate(const Config& p, double* res) {
    int32_t method = MH_DECIMATE_FIRST_POINT;  // (the default of FilterDecimateVoxels; lidar3d-default.yaml:291 spells it out)
    if (p.has("decimate_method")) {
      const std::string m = p["decimate_method"].asString();
      if (ends_with(m, "ClosestToAverage")) method = MH_DECIMATE_CLOSEST_TO_AVERAGE;  // (yaml:292, the commented alternative)
      else if (!ends_with(m, "FirstPoint")) unsupported("decimate_method " + m);
    }
    (res == &decim_map_res ? decim_map_method : decim_icp_method) = method;
    parameterFromConfig(p, "voxel_filter_resolution", res, true);
    const uint32_t mp = p.has("minimum_input_points_to_filter") ? (uint32_t)to_double(p["minimum_input_points_to_filter"].asString()) : 0;
    if (res == &decim_map_res) min_points_to_filter = mp;
    else if (mp != min_points_to_filter) unsupported("different minimum_input_points_to_filter in the two decimations");
  }
  void load(const Config& cfg) {
    if (cfg.has("observations_filter_adjust_timestamps")) {
      const Config& s = cfg["observations_filter_adjust_timestamps"];
      for (size_t i = 0; i < s.size(); i++) {
        if (!ends_with(class_of(s.at(i)), "FilterAdjustTimestamps")) unsupported(class_of(s.at(i)));
        const Config& p = s.at(i)["params"];
        const std::string m = p.getOr("method", "TimestampAdjustMethod::MiddleIsZero");
        timestamp_method = ends_with(m, "MiddleIsZero") ? MH_TS_MIDDLE_IS_ZERO : ends_with(m, "EarliestIsZero") ? MH_TS_EARLIEST_IS_ZERO : -1;
        if (timestamp_method < 0) unsupported("timestamp method " + m);
        if (p.has("time_offset")) parameterFromConfig(p, "time_offset", &time_offset, false);
      }
    }
    // ---- 1st pass
    const Config& f1 = cfg["observations_filter_1st_pass"];
    std::string cur = "raw";
    size_t i = 0;
    auto params_of = [&](size_t k) -> const Config& { return f1.at(k)["params"]; };
    if (i >= f1.size() || !ends_with(class_of(f1.at(i)), "FilterDecimateVoxels")) unsupported("1st pass must start with FilterDecimateVoxels");
    if (params_of(i)["input_pointcloud_layer"].asString() != cur) unsupported("first decimation must read layer 'raw'");
    decimate(params_of(i), &decim_map_res);
    cur = params_of(i)["output_pointcloud_layer"].asString();
    i++;
    if (i < f1.size() && ends_with(class_of(f1.at(i)), "FilterByRange")) {
      const Config& p = params_of(i);
      if (p["input_pointcloud_layer"].asString() != cur || !p.has("output_layer_between")) unsupported("FilterByRange wiring");
      parameterFromConfig(p, "range_min", &range_min, true);
      parameterFromConfig(p, "range_max", &range_max, true);
      cur = p["output_layer_between"].asString();
      i++;
    }
    if (i < f1.size() && ends_with(class_of(f1.at(i)), "FilterBoundingBox")) {
      const Config& p = params_of(i);
      if (p["input_pointcloud_layer"].asString() != cur) unsupported("FilterBoundingBox wiring");
      if (p.has("outside_pointcloud_layer")) { bbox_mode = MH_BBOX_KEEP_OUTSIDE; cur = p["outside_pointcloud_layer"].asString(); }
      else if (p.has("inside_pointcloud_layer")) { bbox_mode = MH_BBOX_KEEP_INSIDE; cur = p["inside_pointcloud_layer"].asString(); }
      else unsupported("FilterBoundingBox without an output layer");
      for (int a = 0; a < 3; a++) {
        declareParameter("bounding_box_min", p["bounding_box_min"].at(a).asString(), &bbox_min[a]);
        declareParameter("bounding_box_max", p["bounding_box_max"].at(a).asString(), &bbox_max[a]);
      }
      i++;
    }
    const std::string skewed_map = cur;
    if (i >= f1.size() || !ends_with(class_of(f1.at(i)), "FilterDecimateVoxels")) unsupported("1st pass must end with FilterDecimateVoxels");
    if (params_of(i)["input_pointcloud_layer"].asString() != cur) unsupported("second decimation wiring");
    decimate(params_of(i), &decim_icp_res);
    const std::string skewed_icp = params_of(i)["output_pointcloud_layer"].asString();
    if (++i != f1.size()) unsupported("extra filters after the second decimation: " + class_of(f1.at(i)));
    // ---- 2nd pass
    const Config& f2 = cfg["observations_filter_2nd_pass"];
    for (size_t k = 0; k < f2.size(); k++) {
      const std::string cn = class_of(f2.at(k));
      if (ends_with(cn, "FilterDeleteLayer")) continue;
      if (!ends_with(cn, "FilterDeskew")) unsupported(cn);
      const Config& p = f2.at(k)["params"];
      const std::string in = p["input_pointcloud_layer"].asString(), out = p["output_pointcloud_layer"].asString();
      if (in == skewed_map) layer_for_map = out;
      else if (in == skewed_icp) layer_for_icp = out;
      else unsupported("FilterDeskew reads unknown layer " + in);
      if (p.has("skip_deskew")) skip_deskew = to_bool(p["skip_deskew"].asString());
    }
    if (layer_for_map.empty() || layer_for_icp.empty()) unsupported("both skewed layers need a FilterDeskew");
    // ---- merge
    const Config& mg = cfg["insert_observation_into_local_map"];
    if (mg.size() != 1 || !ends_with(class_of(mg.at(0)), "FilterMerge")) unsupported("insert_observation_into_local_map must be one FilterMerge");
    const Config& mp = mg.at(0)["params"];
    if (mp["input_pointcloud_layer"].asString() != layer_for_map) unsupported("FilterMerge must read the de-skewed map layer");
    if (mp.has("input_layer_in_local_coordinates") && !to_bool(mp["input_layer_in_local_coordinates"].asString()))
      unsupported("FilterMerge with input_layer_in_local_coordinates: false");
    map_layer = mp["target_layer"].asString();
  }
};

// ================================================================== driver
struct LidarOdometry::RawInput {
  size_t n = 0;
  const float *x = nullptr, *y = nullptr, *z = nullptr, *t = nullptr;  // channel arrays, or ...
  const void* data = nullptr;                                           // ... interleaved records
  size_t point_step = 0, off_x = 0, off_y = 0, off_z = 0;
  long long off_t = -1;
  bool same(const RawInput& o) const {
    return n == o.n && x == o.x && y == o.y && z == o.z && t == o.t && data == o.data && point_step == o.point_step &&
           off_x == o.off_x && off_y == o.off_y && off_z == o.off_z && off_t == o.off_t;
  }
};

struct LidarOdometry::Prefetch {
  RawInput req;  // announced, waiting for the current scan to reach its launch point
  RawInput in;   // handed to the worker
  bool requested = false, launched = false;
  int slot = 0;               // which of the two prefetch sets the worker fills / filled
  mh_preprocess_params pp{};  // the filter parameters the worker used
  // the worker: ONE thread that lives as long as the driver and takes a task per scan (a std::async per scan created a
  // thread per scan: ~20 us of the main thread's time right before its alignment)
  struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> task;
    bool has_task = false, busy = false, stop = false;
    std::exception_ptr error;
    void start() {
      th = std::thread([this] {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
          cv.wait(lk, [this] { return has_task || stop; });
          if (stop) return;
          std::function<void()> t = std::move(task);
          has_task = false;
          lk.unlock();
          std::exception_ptr e;
          try { t(); } catch (...) { e = std::current_exception(); }
          lk.lock();
          error = e;
          busy = false;
          cv.notify_all();
        }
      });
    }
    void submit(std::function<void()> t) {
      if (!th.joinable()) start();
      std::lock_guard<std::mutex> lk(m);
      task = std::move(t);
      has_task = busy = true;
      error = nullptr;
      cv.notify_all();
    }
    void wait() {  // rethrows what the task threw
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [this] { return !busy; });
      if (error) {
        std::exception_ptr e = error;
        error = nullptr;
        std::rethrow_exception(e);
      }
    }
    ~Worker() {
      if (th.joinable()) {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        cv.notify_all();
        th.join();
      }
    }
  } worker;
  bool on_worker = false;
  void join() {
    if (on_worker) {
      on_worker = false;
      worker.wait();
    }
  }
};

LidarOdometry::LidarOdometry(std::shared_ptr<DeviceContext> ctx) : ctx_(std::move(ctx)), pf_(new Prefetch) {}
LidarOdometry::~LidarOdometry() {
  cancel_prefetch();
  if (batcher_) batcher_->forgetOwner(this);
}

void LidarOdometry::initialize(const Config& cfg) {
  if (plan_) throw std::runtime_error("LidarOdometry::initialize() called twice; create a new object instead");
  params_.load_from(cfg["params"]);
  params_.attachToParameterSource(source_);
  if (cfg.has("navstate_fuse_params")) navstate_.initialize(cfg["navstate_fuse_params"]);
  plan_ = std::make_unique<FilterPlan>();
  plan_->load(cfg);
  plan_->attachToParameterSource(source_);

  // ICP pipelines (:340-358)
  auto t0 = icp_pipeline_from_yaml(cfg["icp_settings_with_vel"], ctx_);
  icp_[0] = std::get<0>(t0);
  icp_params_[0] = std::get<1>(t0);
  if (cfg.has("icp_settings_without_vel")) {
    auto t1 = icp_pipeline_from_yaml(cfg["icp_settings_without_vel"], ctx_);
    icp_[1] = std::get<0>(t1);
    icp_params_[1] = std::get<1>(t1);
  } else {
    icp_[1] = icp_[0];
    icp_params_[1] = icp_params_[0];
  }
  for (auto& icp : icp_) {
    icp->attachToParameterSource(source_);
    icp->setKeepFinalPairings(false);
  }
  // local map definition (yaml:213-242), instantiated at the first key-frame when its $f{} formulas can be evaluated
  const Config& gen = cfg["localmap_generator"];
  if (gen.size() < 1) throw std::runtime_error("localmap_generator is empty");
  map_def_ = gen.at(0)["params"]["metric_map_definition"];

  reset();  // device objects are created at the first scan: a pipeline can be loaded and checked without a GPU
}

void LidarOdometry::ensure_device() {
  if (raw_) return;
  if (!ctx_) ctx_ = DeviceContext::Default();
  raw_ = std::make_shared<DevicePointCloud>(ctx_);
  map_skewed_ = std::make_shared<DevicePointCloud>(ctx_);
  icp_skewed_ = std::make_shared<DevicePointCloud>(ctx_);
  for_map_ = std::make_shared<DevicePointCloud>(ctx_);
  for_icp_ = std::make_shared<DevicePointCloud>(ctx_);
}

void LidarOdometry::resolve_map_counts() const {
  if (!map_counts_pending_) return;
  map_points_cached_ = local_map_ ? local_map_->size() : 0;  // (mh_map_get_info: waits for the update if it still runs)
  map_voxels_cached_ = local_map_ ? local_map_->voxelCount() : 0;
  for (size_t i = map_counts_from_; i < records_.size(); i++) {
    if (records_[i].dropped) continue;  // (those returned before the map was looked at)
    records_[i].n_map_points = map_points_cached_;
    records_[i].n_map_voxels = map_voxels_cached_;
  }
  map_counts_pending_ = false;
  // what map_is_empty() answers from now on follows the device's own count (an update with far-voxel removal may leave
  // the map empty: local_map_->empty() is what the reference asks every scan, LidarOdometry.cpp:817)
  map_known_nonempty_ = map_points_cached_ != 0;
}

bool LidarOdometry::map_is_empty() {
  if (!local_map_) return true;
  if (map_known_nonempty_) return false;
  map_known_nonempty_ = local_map_->size() != 0;
  return !map_known_nonempty_;
}

void LidarOdometry::reset() {
  cancel_prefetch();
  map_counts_pending_ = false;
  map_counts_from_ = 0;
  map_points_cached_ = map_voxels_cached_ = 0;
  map_known_nonempty_ = false;
  navstate_.reset();
  local_map_.reset();
  last_lidar_pose_ = CPose3D();
  last_icp_was_good_ = true;
  last_icp_quality_ = 0;
  last_obs_tim_.reset();
  last_icp_timestamp_.reset();
  first_ever_timestamp_.reset();
  last_obs_timestamp_.reset();
  last_motion_model_output_.reset();
  adapt_thres_sigma_ = 0;
  estimated_sensor_max_range_.reset();
  instantaneous_sensor_max_range_.reset();
  distance_checker_local_map_.clear();
  localmap_check_removal_counter_ = 0;
  trajectory_.clear();
  records_.clear();
}

void LidarOdometry::updatePipelineTwistVariables(const Twist& tw) {  // :1571-1579
  source_.updateVariable("vx", tw.vx); source_.updateVariable("vy", tw.vy); source_.updateVariable("vz", tw.vz);
  source_.updateVariable("wx", tw.wx); source_.updateVariable("wy", tw.wy); source_.updateVariable("wz", tw.wz);
}

void LidarOdometry::updatePipelineDynamicVariables() {  // :1581-1635
  updatePipelineTwistVariables(last_motion_model_output_ ? last_motion_model_output_->twist : Twist());
  const TPose3D p = last_lidar_pose_.asTPose();
  source_.updateVariable("robot_x", p.x); source_.updateVariable("robot_y", p.y); source_.updateVariable("robot_z", p.z);
  source_.updateVariable("robot_yaw", p.yaw); source_.updateVariable("robot_pitch", p.pitch); source_.updateVariable("robot_roll", p.roll);
  source_.updateVariable("ADAPTIVE_THRESHOLD_SIGMA", adapt_thres_sigma_ != 0 ? adapt_thres_sigma_ : params_.initial_sigma);
  source_.updateVariable("ICP_ITERATION", 0);
  const auto& vars = source_.getVariableValues();
  for (const char* v : {"icp_iterations", "SENSOR_TIME_OFFSET", "twistCorrectionCount"})
    if (!vars.count(v)) source_.updateVariable(v, 0);
  if (estimated_sensor_max_range_) source_.updateVariable("ESTIMATED_SENSOR_MAX_RANGE", *estimated_sensor_max_range_);
  source_.updateVariable("INSTANTANEOUS_SENSOR_MAX_RANGE", instantaneous_sensor_max_range_ ? *instantaneous_sensor_max_range_ : 20.0);
  if (last_obs_timestamp_ && first_ever_timestamp_)
    source_.updateVariable("current_relative_timestamp", *last_obs_timestamp_ - *first_ever_timestamp_);
  source_.realize();
}

// bounding-box "radius" used by the sensor range estimate (:1503-1508, 1523-1527): float norms, like TPoint3Df::norm()
static double bbox_radius(const float mn[3], const float mx[3]) {
  const float a = std::sqrt((mx[0] * mx[0] + mx[1] * mx[1]) + mx[2] * mx[2]);
  const float b = std::sqrt((mn[0] * mn[0] + mn[1] * mn[1]) + mn[2] * mn[2]);
  return (double)std::max(a, b);
}

static mh_preprocess_params make_pp(double decim_map_res, double decim_icp_res, uint32_t min_points_to_filter, double range_min,
                                    double range_max, int32_t bbox_mode, const double bbox_min[3], const double bbox_max[3],
                                    int32_t timestamp_method, double time_offset, int32_t decim_map_method = MH_DECIMATE_FIRST_POINT,
                                    int32_t decim_icp_method = MH_DECIMATE_FIRST_POINT) {
  mh_preprocess_params pp;
  memset(&pp, 0, sizeof(pp));  // (compared bytewise with the parameters a prefetch used)
  pp.decim_map_resolution = (float)decim_map_res;
  pp.decim_icp_resolution = (float)decim_icp_res;
  pp.min_points_to_filter = min_points_to_filter;
  pp.index_mode = (int32_t)molahip_host::plugin_switches().index_mode;  // MOLA_HIP_INDEX_MODE (default floor)
  pp.range_min = (float)range_min;
  pp.range_max = (float)range_max;
  pp.bbox_mode = bbox_mode;
  for (int a = 0; a < 3; a++) {
    pp.bbox_min[a] = (float)bbox_min[a];
    pp.bbox_max[a] = (float)bbox_max[a];
  }
  pp.timestamp_method = timestamp_method;
  pp.time_offset = (float)time_offset;
  pp.decim_map_method = decim_map_method;
  pp.decim_icp_method = decim_icp_method;
  return pp;
}

void LidarOdometry::run_first_pass() {
  const FilterPlan& f = *plan_;
  const mh_preprocess_params pp = make_pp(f.decim_map_res, f.decim_icp_res, f.min_points_to_filter, f.range_min, f.range_max,
                                          f.bbox_mode, f.bbox_min, f.bbox_max, f.timestamp_method, f.time_offset, f.decim_map_method, f.decim_icp_method);
  // (not through the batcher even when there is one: its filter sets are made of the PREFETCH requests, one action per
  // alignment and participant -- this call is the first scan of a sequence, or a prepared scan that has to be redone)
  check(mh_scan_preprocess(raw_->handle(), &pp, map_skewed_->handle(), icp_skewed_->handle()), "mh_scan_preprocess");
}

void LidarOdometry::setAlignBatcher(std::shared_ptr<mp2p_icp_hip::AlignBatcher> b) {
  // the instances of a batch run on their own host threads: each needs a context (stream + scratch) of its own, the
  // process-wide default one would be shared between threads (molahip.h: one context, one thread at a time)
  if (b && (!ctx_ || ctx_ == DeviceContext::Default()))
    throw std::runtime_error("LidarOdometry::setAlignBatcher: this instance uses the process-wide default context; construct "
                             "it with a DeviceContext of its own");
  for (auto& i : icp_)
    if (i) i->setAlignBatcher(b, this);  // (both ICP objects align for this one participant)
  batcher_ = std::move(b);
}

// ---- the announced next observation: upload + first pass on a second stream while this scan is in its ICP loop
void LidarOdometry::prefetch(const float* x, const float* y, const float* z, const float* t, size_t n) {
  pf_->req = RawInput();  // (a prepared scan that is still waiting to be picked up stays untouched)
  pf_->req.n = n; pf_->req.x = x; pf_->req.y = y; pf_->req.z = z; pf_->req.t = t;
  pf_->requested = n > 0;
}

void LidarOdometry::prefetchInterleaved(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                                        size_t off_z, long long off_t, const float* t) {
  pf_->req = RawInput();
  pf_->req.n = n; pf_->req.data = data; pf_->req.point_step = point_step; pf_->req.off_x = off_x; pf_->req.off_y = off_y;
  pf_->req.off_z = off_z; pf_->req.off_t = off_t; pf_->req.t = t;
  pf_->requested = n > 0;
}

void LidarOdometry::cancel_prefetch() {
  if (pf_->launched) {
    try { pf_->join(); } catch (...) {}
  }
  pf_->launched = pf_->requested = false;
}

void LidarOdometry::launch_prefetch() {
  if (!pf_->requested || !plan_ || !estimated_sensor_max_range_) {
    return;  // (the last scan of the sequence, or nothing announced: the batcher sees this sequence's next alignment)
  }
  if (pf_->launched) {  // prepared but never picked up: drop it
    try { pf_->join(); } catch (...) {}
    pf_->launched = false;
  }
  pf_->in = pf_->req;
  if (!ctx_b_) {
    // the next scan's upload and filters run beside the current alignment on a stream of their own.  (Measured in round 3 and
    // removed in round 4: a stream of the LOW priority class -- 8 sequences 2180 scans/s against 2870 -- and a stream
    // restricted to a quarter or an eighth of the compute units -- 4530-4650 against 4780: whatever the filters and uploads
    // take from the alignment's kernels, it is neither dispatch priority nor wave slots.)
    ctx_b_ = std::make_shared<DeviceContext>(ctx_->device());
    for (int i = 0; i < 2; i++) {
      raw_b_[i] = std::make_shared<DevicePointCloud>(ctx_b_);
      map_skewed_b_[i] = std::make_shared<DevicePointCloud>(ctx_b_);
      icp_skewed_b_[i] = std::make_shared<DevicePointCloud>(ctx_b_);
    }
  }
  pf_->slot ^= 1;  // not the set the current scan may still be reading
  // the filter parameters as the next onLidar will publish them (:692): only the sensor range differs from now
  std::map<std::string, double> vars = source_.getVariableValues();
  vars["ESTIMATED_SENSOR_MAX_RANGE"] = *estimated_sensor_max_range_;
  vars["INSTANTANEOUS_SENSOR_MAX_RANGE"] = instantaneous_sensor_max_range_ ? *instantaneous_sensor_max_range_ : 20.0;
  plan_->realizeWith(vars);
  const FilterPlan& f = *plan_;
  pf_->pp = make_pp(f.decim_map_res, f.decim_icp_res, f.min_points_to_filter, f.range_min, f.range_max, f.bbox_mode,
                    f.bbox_min, f.bbox_max, f.timestamp_method, f.time_offset, f.decim_map_method, f.decim_icp_method);
  plan_->realizeWith(source_.getVariableValues());  // and back: this scan goes on with what it started with
  const RawInput in = pf_->in;
  const mh_preprocess_params pp = pf_->pp;
  auto raw = raw_b_[pf_->slot], ms = map_skewed_b_[pf_->slot], is = icp_skewed_b_[pf_->slot];
  auto ctx = ctx_b_;
  const bool pinned = input_pinned_;
  auto batcher = batcher_;
  const void* owner = this;
  // several sequences in one process: the request is announced HERE, on the thread that aligns next (the batcher then
  // knows it is coming before it sees that alignment), and delivered by the worker
  const size_t filter_set = batcher ? batcher->announceFilter(owner) : 0;
  auto work = [in, pp, raw, ms, is, ctx, pinned, batcher, owner, filter_set]() {
    bool delivered = false;
    try {
      if (in.data) raw->setPointsInterleaved(in.data, in.n, in.point_step, in.off_x, in.off_y, in.off_z, in.off_t, pinned);
      else raw->setPoints(in.x, in.y, in.z, in.n);
      if (in.t) raw->setTimestamps(in.t, in.n);
      if (batcher) {
        std::string err;
        delivered = true;
        const mh_status st = batcher->preprocess(owner, filter_set, raw->handle(), &pp, ms->handle(), is->handle(), &err);
        if (st != MH_OK) throw std::runtime_error("mh_scan_preprocess (prefetch, batched): " + err);
      } else {
        check(mh_scan_preprocess(raw->handle(), &pp, ms->handle(), is->handle()), "mh_scan_preprocess (prefetch)");
      }
      ctx->synchronize();
    } catch (...) {
      if (batcher && !delivered) batcher->cancelAnnouncedFilter(owner, filter_set);  // (the others must not wait for it)
      throw;
    }
  };
  pf_->worker.submit(work);
  pf_->on_worker = true;
  pf_->launched = true;
  pf_->requested = false;
}

void LidarOdometry::run_second_pass() {
  const auto& v = source_.getVariableValues();
  const double tw[6] = {v.at("vx"), v.at("vy"), v.at("vz"), v.at("wx"), v.at("wy"), v.at("wz")};
  const double* twp = plan_->skip_deskew ? nullptr : tw;
  // both layers and the bounding box of the de-skewed ICP layer (read by the sensor-range estimate right below) at once
  check(mh_scan_deskew_pair(cur_map_skewed_->handle(), cur_icp_skewed_->handle(), twp, for_map_->handle(), for_icp_->handle(),
                            icp_bb_min_, icp_bb_max_, nullptr), "mh_scan_deskew_pair");
}

void LidarOdometry::doUpdateAdaptiveThreshold(const CPose3D& err) {  // :1449-1485 (KISS-ICP's scheme)
  if (!estimated_sensor_max_range_) return;
  const double max_range = *estimated_sensor_max_range_;
  const double theta = err.rotationAngle();
  const double model_error = err.translationNorm() + 2.0 * max_range * std::sin(theta / 2.0);
  double rot_error = 0;
  if (last_motion_model_output_) {
    const Twist& tw = last_motion_model_output_->twist;
    rot_error = 0.1 * std::sqrt(tw.wx * tw.wx + tw.wy * tw.wy + tw.wz * tw.wz) * max_range;
  }
  const double KP = params_.kp;
  if (!(KP > 1.0)) throw std::runtime_error("adaptive_threshold.kp must be > 1");
  const double gain = std::min(KP, std::max(0.1, KP * (1.0 - last_icp_quality_)));
  const double new_sigma = (model_error + rot_error) * gain;
  if (adapt_thres_sigma_ == 0) adapt_thres_sigma_ = params_.initial_sigma;
  adapt_thres_sigma_ = params_.alpha * adapt_thres_sigma_ + (1.0 - params_.alpha) * new_sigma;
  adapt_thres_sigma_ = std::min(params_.maximum_sigma, std::max(params_.min_motion, adapt_thres_sigma_));
}

void LidarOdometry::create_local_map() {  // :1165-1171 with yaml:228-242
  const auto& vars = source_.getVariableValues();
  const std::string cls = map_def_["class"].asString();
  const Config& co = map_def_["creationOpts"];
  const Config& io = map_def_["insertOpts"];
  mh_map_params mp{};
  map_voxel_size_ = eval_now(co["voxel_size"].asString(), vars);
  mp.voxel_size = (float)map_voxel_size_;
  mp.max_points_per_voxel = io.has("max_points_per_voxel") ? (uint32_t)eval_now(io["max_points_per_voxel"].asString(), vars) : 0;
  mp.index_mode = molahip_host::plugin_switches().index_mode;
  mp.far_voxel_metric = molahip_host::plugin_switches().far_voxel_metric;  // MOLA_HIP_FAR_VOXEL_METRIC
  mp.min_distance_between_points = io.has("min_distance_between_points") ? (float)eval_now(io["min_distance_between_points"].asString(), vars) : 0.f;
  remove_voxels_farther_than_ = io.has("remove_voxels_farther_than") ? (float)eval_now(io["remove_voxels_farther_than"].asString(), vars) : 0.f;
  if (ends_with(cls, "NDT")) {
    mp.ndt_max_eigen_ratio = io.has("max_eigen_ratio_for_planes") ? (float)eval_now(io["max_eigen_ratio_for_planes"].asString(), vars) : 0.05f;
    mp.ndt_min_points = 4;
  } else if (!ends_with(cls, "HashedVoxelPointCloud")) {
    throw std::runtime_error("local map class '" + cls + "' has no device implementation (HashedVoxelPointCloud, NDT)");
  }
  local_map_ = std::make_shared<HashedVoxelPointCloud>(mp, ctx_);
}

const LidarOdometry::ScanRecord& LidarOdometry::onLidar(double this_obs_tim, const float* x, const float* y, const float* z,
                                                        const float* t, size_t n) {
  RawInput in;
  in.n = n; in.x = x; in.y = y; in.z = z; in.t = t;
  return process(this_obs_tim, in);
}

const LidarOdometry::ScanRecord& LidarOdometry::onLidarInterleaved(double this_obs_tim, const void* data, size_t n,
                                                                   size_t point_step, size_t off_x, size_t off_y,
                                                                   size_t off_z, long long off_t, const float* t) {
  RawInput in;
  in.n = n; in.data = data; in.point_step = point_step; in.off_x = off_x; in.off_y = off_y; in.off_z = off_z;
  in.off_t = off_t; in.t = t;
  return process(this_obs_tim, in);
}

const LidarOdometry::ScanRecord& LidarOdometry::process(double this_obs_tim, const RawInput& in) {
  const size_t n = in.n;
  const bool has_t = in.t != nullptr || (in.data && in.off_t >= 0);
  (void)has_t;
  if (!plan_) throw std::runtime_error("LidarOdometry::onLidar called before initialize()");
  records_.emplace_back();
  ScanRecord& rec = records_.back();
  rec.timestamp = this_obs_tim;
  rec.n_raw = n;
  rec.pose = last_lidar_pose_;

  // drop scans too close in time (:644-657)
  if (last_obs_tim_ && (this_obs_tim - *last_obs_tim_) < params_.min_time_between_scans) {
    rec.dropped = true;
    return rec;
  }
  StageTimer t_all(profile_, "onLidar");
  ensure_device();
  cur_raw_ = raw_;
  cur_map_skewed_ = map_skewed_;
  cur_icp_skewed_ = icp_skewed_;
  bool prepared = false;  // upload + first pass already done by the prefetch worker?
  if (pf_->launched) {
    StageTimer tt(profile_, "onLidar.0.prefetch_wait");
    bool ok = true;
    try { pf_->join(); } catch (...) { ok = false; }  // (a failed prefetch is simply redone below, and reports there)
    pf_->launched = false;
    if (ok && pf_->in.same(in)) prepared = true;
  }
  if (pf_->requested && pf_->req.same(in)) pf_->requested = false;  // due before it could be launched
  if (!prepared) {
    StageTimer tt(profile_, "onLidar.0.upload_raw");
    if (in.data) raw_->setPointsInterleaved(in.data, n, in.point_step, in.off_x, in.off_y, in.off_z, in.off_t, input_pinned_);
    else raw_->setPoints(in.x, in.y, in.z, n);
    if (in.t) raw_->setTimestamps(in.t, n);
  }

  // first call: sensor range from the raw cloud (:660, 1487-1513)
  if (!estimated_sensor_max_range_ && n) {
    float mn[3], mx[3];
    cur_raw_->boundingBox(mn, mx);
    estimated_sensor_max_range_ = std::max(bbox_radius(mn, mx), params_.absolute_minimum_sensor_range);
  }
  {
    StageTimer tt(profile_, "onLidar.0.dynamic_variables");
    updatePipelineDynamicVariables();  // :692
  }
  rec.twist = last_motion_model_output_ ? last_motion_model_output_->twist : Twist();

  if (prepared) {  // valid only if the parameters published just now are the ones the worker used
    const FilterPlan& f = *plan_;
    const mh_preprocess_params now = make_pp(f.decim_map_res, f.decim_icp_res, f.min_points_to_filter, f.range_min, f.range_max,
                                             f.bbox_mode, f.bbox_min, f.bbox_max, f.timestamp_method, f.time_offset, f.decim_map_method, f.decim_icp_method);
    if (memcmp(&now, &pf_->pp, sizeof(now)) == 0) {
      cur_raw_ = raw_b_[pf_->slot];
      cur_map_skewed_ = map_skewed_b_[pf_->slot];
      cur_icp_skewed_ = icp_skewed_b_[pf_->slot];
      profile_["prefetch_hits"] += 1.0;
    } else {
      prepared = false;
      profile_["prefetch_misses"] += 1.0;
      StageTimer tt(profile_, "onLidar.0.upload_raw");
      if (in.data) raw_->setPointsInterleaved(in.data, n, in.point_step, in.off_x, in.off_y, in.off_z, in.off_t, input_pinned_);
      else raw_->setPoints(in.x, in.y, in.z, n);
      if (in.t) raw_->setTimestamps(in.t, n);
    }
  }
  if (!prepared) {
    StageTimer tt(profile_, "onLidar.1.filter_1st");
    run_first_pass();  // :734
  }
  {
    StageTimer tt(profile_, "onLidar.1.filter_2nd");
    run_second_pass();  // :739
  }
  rec.decim_map_resolution = plan_->decim_map_res;
  rec.decim_icp_resolution = plan_->decim_icp_res;
  rec.n_for_map = for_map_->size();
  rec.n_for_icp = for_icp_->size();

  // sensor range low-pass from the first point layer of the observation, 'decimated_for_icp' (:744, 1515-1545)
  if (estimated_sensor_max_range_) {
    StageTimer tt(profile_, "onLidar.2.sensor_range");
    const double radius = std::max(bbox_radius(icp_bb_min_, icp_bb_max_), params_.absolute_minimum_sensor_range);  // (run_second_pass)
    instantaneous_sensor_max_range_ = radius;
    const double a = params_.max_sensor_range_filter_coefficient;
    estimated_sensor_max_range_ = *estimated_sensor_max_range_ * a + radius * (1.0 - a);
  }
  launch_prefetch();  // the announced next scan: its first pass only needs the range estimate, which is final now
  rec.estimated_sensor_max_range = estimated_sensor_max_range_.value_or(0);
  rec.instantaneous_sensor_max_range = instantaneous_sensor_max_range_.value_or(0);

  if (params_.validity_check_enabled && !(n > params_.validity_minimum_point_count)) {  // :749-757, 1548-1568
    rec.dropped = true;
    return rec;
  }
  last_obs_tim_ = this_obs_tim;
  last_obs_timestamp_ = this_obs_tim;
  if (!first_ever_timestamp_) first_ever_timestamp_ = this_obs_tim;
  if (n == 0) {  // :769-775
    rec.dropped = true;
    return rec;
  }

  bool updateLocalMap = false;
  {
    StageTimer tt(profile_, "onLidar.2.navstate");
    last_motion_model_output_ = navstate_.estimated_navstate(this_obs_tim);  // :810-811
  }
  const bool hasMotionModel = last_motion_model_output_.has_value();
  rec.had_motion_model = hasMotionModel;

  const bool map_empty = map_is_empty();
  if (map_empty) {
    // first point cloud: no ICP, it becomes the map (:817-838)
    rec.first_scan = true;
    updateLocalMap = true;
    trajectory_.emplace_back(this_obs_tim, last_lidar_pose_);
    navstate_.fuse_pose(this_obs_tim, CPose3D());
  } else {
    // ---- ICP (:840-1024)
    TPose3D init_guess = hasMotionModel ? last_motion_model_output_->pose.mean.asTPose() : last_lidar_pose_.asTPose();
    std::optional<CPose3DPDFGaussianInf> prior;
    if (hasMotionModel) {
      bool any = false;
      for (double v : last_motion_model_output_->pose.cov_inv) any = any || v != 0.0;
      if (any) prior = last_motion_model_output_->pose;
    }
    const int kind = hasMotionModel ? 0 : 1;
    const CPose3D last_keyframe_pose = last_lidar_pose_;
    double time_since_last_keyframe = 0;
    if (last_icp_timestamp_) time_since_last_keyframe = this_obs_tim - *last_icp_timestamp_;
    last_icp_timestamp_ = this_obs_tim;

    TPose3D current_solution = init_guess;
    rec.init_guess = CPose3D(init_guess);
    ICP& icp = *icp_[kind];
    StageTimer t_icp_all(profile_, "onLidar.3.icp_with_setup");
    mp2p_icp_hip::Parameters icp_params = icp_params_[kind];
    size_t remaining = icp_params.maxIterations;
    mp2p_icp_hip::Results res;
    mp2p_icp_hip::metric_map_t obs, glob;
    obs.layers[plan_->layer_for_icp] = for_icp_;
    obs.layers[plan_->layer_for_map] = for_map_;
    glob.layers[plan_->map_layer] = local_map_;
    std::optional<StageTimer> t_icp;
    t_icp.emplace(profile_, "onLidar.3.run_icp");
    do {
      icp_params.maxIterations = (uint32_t)remaining;
      // the in-tree hook (:919-952) only compares the running solution with its check point: evaluated on the device.
      // NOTE the reference increments optimize_twist_max_corrections instead of its counter (:941), so the number of
      // corrections is bounded by the shared iteration budget only; kept.
      if (params_.optimize_twist)
        icp.setDeviceHook(params_.optimize_twist_rerun_min_trans, params_.optimize_twist_rerun_min_rot_deg * kDeg2Rad,
                          CPose3D(current_solution));
      else
        icp.clearHooks();
      icp.align(obs, glob, current_solution, icp_params, res, prior);  // :961-962
      rec.align_calls++;
      profile_["icp.host_polls"] += icp.lastAlignHostPolls();
      profile_["icp.enqueued_iterations"] += icp.lastAlignEnqueuedIterations();
      profile_["icp.executed_iterations"] += (double)res.nIterations;
      profile_["icp.align_calls"] += 1.0;
      profile_["onLidar.3.icp_host_setup"] += icp.lastAlignSetupSeconds();
      remaining -= std::min(remaining, res.nIterations);
      rec.icp_iterations += (uint32_t)res.nIterations;
      if (res.terminationReason == IterTermReason::HookRequest) {
        current_solution = res.optimal_tf.mean.asTPose();  // what the hook stored (:949)
        rec.twist_corrections++;
        if (time_since_last_keyframe > 0) {
          // re-estimate the twist from the running solution and de-skew again (:973-1004), all on the device
          const CPose3D incr = res.optimal_tf.mean - last_keyframe_pose;
          double w[3];
          incr.so3Log(w);
          const double At = time_since_last_keyframe;
          Twist tw;
          tw.vx = incr.T[3] / At; tw.vy = incr.T[7] / At; tw.vz = incr.T[11] / At;
          tw.wx = w[0] / At; tw.wy = w[1] / At; tw.wz = w[2] / At;
          updatePipelineTwistVariables(tw);
          source_.realize();
          run_second_pass();
          rec.twist = tw;
        }
      }
    } while (res.terminationReason == IterTermReason::HookRequest);
    t_icp.reset();
    icp.clearHooks();
    rec.icp_run = true;
    rec.termination = (int)res.terminationReason;
    rec.goodness = res.quality;

    // ---- gate, motion model, trajectory (:1026-1045)
    StageTimer t_post(profile_, "onLidar.3.post_icp");
    const bool icpIsGood = res.quality >= params_.min_icp_goodness;
    last_icp_was_good_ = icpIsGood;
    last_icp_quality_ = res.quality;
    rec.icp_good = icpIsGood;
    if (icpIsGood) {
      last_lidar_pose_ = res.optimal_tf.mean;
      navstate_.fuse_pose(this_obs_tim, res.optimal_tf.mean, res.optimal_tf.cov);
      trajectory_.emplace_back(this_obs_tim, last_lidar_pose_);
    } else {
      navstate_.reset();
    }
    source_.updateVariable("icp_iterations", (double)res.nIterations);
    source_.updateVariable("twistCorrectionCount", 0);  // always 0 in the reference (App. C.2)

    // ---- adaptive threshold, also after a rejected ICP (:1052-1064)
    if (params_.adaptive_threshold_enabled) doUpdateAdaptiveThreshold(res.optimal_tf.mean - CPose3D(init_guess));

    // ---- key-frame decision (:1066-1118)
    const auto [isFirstPoseInChecker, distanceToClosest] = distance_checker_local_map_.check(last_lidar_pose_);
    const double dist_eucl_since_last = distanceToClosest.translationNorm();
    const double rot_since_last = distanceToClosest.rotationAngle();
    updateLocalMap = icpIsGood && params_.local_map_updates_enabled && hasMotionModel &&
                     (isFirstPoseInChecker || dist_eucl_since_last > params_.min_translation_between_keyframes ||
                      rot_since_last > params_.min_rotation_between_keyframes * kDeg2Rad);
    if (updateLocalMap) {
      distance_checker_local_map_.insert(last_lidar_pose_);
      if (params_.max_distance_to_keep_keyframes > 0 &&
          (localmap_check_removal_counter_++ >= params_.check_for_removal_every_n)) {
        localmap_check_removal_counter_ = 0;
        distance_checker_local_map_.removeAllFartherThan(last_lidar_pose_, params_.max_distance_to_keep_keyframes);
      }
    }
  }

  // a bad ICP right after the start: begin again from an empty map (:1146-1156)
  if (!last_icp_was_good_ && trajectory_.size() == 1) {
    resolve_map_counts();  // (earlier records keep the counts of the map they saw)
    if (local_map_) local_map_->clear();
    map_known_nonempty_ = false;
    map_points_cached_ = map_voxels_cached_ = 0;
    trajectory_.clear();
    updateLocalMap = false;
    last_icp_was_good_ = true;
    rec.restarted = true;
  }

  // ---- local map update (:1158-1206): FilterMerge of the de-skewed map layer at the current pose, on the device
  if (updateLocalMap) {
    StageTimer tt(profile_, "onLidar.4.update_local_map");
    if (!local_map_) create_local_map();
    updatePipelineDynamicVariables();  // robot_x..robot_roll (:1194)
    resolve_map_counts();  // the previous update's counters (it finished before this scan's alignment started: no wait)
    // asynchronous: the update runs on the map's own stream and is waited for by the next use of the map only
    local_map_->insertPointCloud(*for_map_, last_lidar_pose_, remove_voxels_farther_than_);
    rec.map_updated = true;
    if (rec.n_for_map == 0) map_known_nonempty_ = false;  // (nothing offered: ask the device next time)
    map_counts_pending_ = true;
    map_counts_from_ = records_.size() - 1;
  }
  // the next alignment's threshold schedules while the device works on the map update (the formulas' variables -- the
  // adaptive sigma -- are final for this scan; align() checks the values and evaluates again if they differ after all)
  if (icp_[0] && local_map_) icp_[0]->precomputeSchedule(icp_params_[0].maxIterations);
  rec.pose = last_lidar_pose_;
  rec.sigma = adapt_thres_sigma_;
  rec.map_voxel_size = map_voxel_size_;
  if (!map_counts_pending_) {  // (otherwise filled by resolve_map_counts())
    rec.n_map_points = map_points_cached_;
    rec.n_map_voxels = map_voxels_cached_;
  }
  return rec;
}

std::map<std::string, std::string> LidarOdometry::describePipeline() const {
  std::map<std::string, std::string> d;
  if (!plan_) return d;
  d["layer_for_map"] = plan_->layer_for_map;
  d["layer_for_icp"] = plan_->layer_for_icp;
  d["map_layer"] = plan_->map_layer;
  d["map_class"] = map_def_["class"].asString();
  d["bbox_mode"] = std::to_string(plan_->bbox_mode);
  d["timestamp_method"] = std::to_string(plan_->timestamp_method);
  d["min_points_to_filter"] = std::to_string(plan_->min_points_to_filter);
  d["skip_deskew"] = plan_->skip_deskew ? "true" : "false";
  for (const auto& p : plan_->declaredParameters()) d["formula:" + p.name + (d.count("formula:" + p.name) ? "#" + std::to_string(d.size()) : "")] = p.expr;
  for (const auto& p : params_.declaredParameters()) d["formula:" + p.name] = p.expr;
  return d;
}

void LidarOdometry::saveTrajectoryTUM(const std::string& path) const {
  FILE* f = fopen(path.c_str(), "w");
  if (!f) throw std::runtime_error("cannot write " + path);
  for (const auto& [t, p] : trajectory_) {
    // rotation matrix -> unit quaternion (w >= 0)
    const double* T = p.T;
    const double tr = T[0] + T[5] + T[10];
    double qw, qx, qy, qz;
    if (tr > 0) {
      const double s = std::sqrt(tr + 1.0) * 2;
      qw = 0.25 * s; qx = (T[9] - T[6]) / s; qy = (T[2] - T[8]) / s; qz = (T[4] - T[1]) / s;
    } else if (T[0] > T[5] && T[0] > T[10]) {
      const double s = std::sqrt(1.0 + T[0] - T[5] - T[10]) * 2;
      qw = (T[9] - T[6]) / s; qx = 0.25 * s; qy = (T[1] + T[4]) / s; qz = (T[2] + T[8]) / s;
    } else if (T[5] > T[10]) {
      const double s = std::sqrt(1.0 + T[5] - T[0] - T[10]) * 2;
      qw = (T[2] - T[8]) / s; qx = (T[1] + T[4]) / s; qy = 0.25 * s; qz = (T[6] + T[9]) / s;
    } else {
      const double s = std::sqrt(1.0 + T[10] - T[0] - T[5]) * 2;
      qw = (T[4] - T[1]) / s; qx = (T[2] + T[8]) / s; qy = (T[6] + T[9]) / s; qz = 0.25 * s;
    }
    if (qw < 0) { qw = -qw; qx = -qx; qy = -qy; qz = -qz; }
    fprintf(f, "%.9f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", t, T[3], T[7], T[11], qx, qy, qz, qw);
  }
  fclose(f);
}

}  // namespace mola_hip
