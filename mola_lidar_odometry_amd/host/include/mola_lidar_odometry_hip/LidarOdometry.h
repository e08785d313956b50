// LidarOdometry.h -- stand-alone LiDAR odometry driver over libmolahip.
//
// Restates the per-scan control flow of mola::LidarOdometry::onLidarImpl (module/src/LidarOdometry.cpp:622-1313,
// all "file:line" relative to /root/reference) so that a whole sequence can run on a box without the MOLA stack,
// configured by the reference's own pipeline file (pipelines/lidar3d-default.yaml / lidar3d-ndt.yaml):
//
//   raw scan -> [device] time-stamp adjust, decimate, range / box filters, de-skew      (yaml:267-350; :730-741)
//            -> constant-velocity guess                                                  (:808-811, 854-877)
//            -> [device] ICP with the twist-re-estimation hook loop                      (:919-1007)
//            -> goodness gate, motion-model update, trajectory                           (:1026-1045)
//            -> adaptive sigma                                                           (:1052-1064, 1437-1485)
//            -> key-frame decision, [device] local-map update                            (:1066-1118, 1160-1206)
//
// Host code here is control logic only; every point touches the GPU through include/molahip.h.  What is NOT here:
// MOLA module plumbing, multi-LiDAR sync, IMU/GNSS/wheel inputs, simplemap generation, visualisation, ROS.
#pragma once
#include <map>
#include <memory>
#include <optional>
#include <string>
#include <vector>

#include "mp2p_icp_hip/mp2p_icp_hip.h"

namespace mola_hip {

using mp2p_icp_hip::Config;
using mp2p_icp_hip::CPose3D;
using mp2p_icp_hip::CPose3DPDFGaussianInf;
using mp2p_icp_hip::DeviceContext;
using mp2p_icp_hip::DevicePointCloud;
using mp2p_icp_hip::HashedVoxelPointCloud;
using mp2p_icp_hip::ICP;
using mp2p_icp_hip::Parameterizable;
using mp2p_icp_hip::ParameterSource;
using mp2p_icp_hip::TPose3D;

struct Twist {
  double vx = 0, vy = 0, vz = 0, wx = 0, wy = 0, wz = 0;  // vehicle frame, the frame FilterDeskew integrates in
};

// Role of mola::NavStateFuse [U] (mola_navstate_fuse, not vendored; LidarOdometry.cpp:338, 810-811, 838, 1035-1038)
// restated as a constant-velocity model: the twist is the increment between the last two fused poses over their time
// difference (or `initial_twist`, yaml:137 / MOLA_INITIAL_VX of eval/cli_kitti.sh:25, while only one pose is known and
// that twist is not zero); the extrapolation composes the last pose with (Exp_SO3(w dt), v dt).  The upstream module is
// a sliding-window factor graph whose marginal also yields the prior information matrix that LidarOdometry.cpp:859-861
// hands to align(); here the prior is the simple propagation upstream used before the factor graph [U]: covariance
// of the last fused pose (the previous ICP result) + (sigma_random_walk_acceleration_{linear,angular} * dt)^2 on the
// diagonal (yaml:132-133), inverted.  It is switched by `motion_model_prior` (this implementation's own key; env
// MOLA_HIP_MOTION_MODEL_PRIOR in the -hip pipelines) and OFF by default: its strength relative to upstream's marginal
// is unverified, and a prior of the wrong strength pulls the solution towards the constant-velocity prediction (the
// synthetic drive that pulls away from rest at 13 m/s^2 ends 0.31 m off with it, 0.15 m without).  Poses older than
// max_time_to_use_velocity_model, or not newer than the last one, do not produce a twist.
class NavStateFuse {
 public:
  struct NavState {
    CPose3DPDFGaussianInf pose;  // cov_inv in the solver's tangent order [v; w]
    Twist twist;
  };
  double max_time_to_use_velocity_model = 2.0;  // [s] (yaml:130)
  double sigma_random_walk_acceleration_linear = 1.0;    // [m/s^2] (yaml:132)
  double sigma_random_walk_acceleration_angular = 10.0;  // [rad/s^2] (yaml:133)
  bool motion_model_prior = false;
  std::optional<Twist> initial_twist;
  void initialize(const Config& c);
  void reset();
  // cov: 6x6 row-major covariance of `pose` in (x,y,z,yaw,pitch,roll) (Results::optimal_tf.cov), or null = exact
  void fuse_pose(double t, const CPose3D& pose, const double* cov = nullptr);
  std::optional<NavState> estimated_navstate(double t) const;

 private:
  std::optional<CPose3D> last_pose_;
  double last_cov_[36] = {0};
  double last_t_ = 0;
  std::optional<Twist> twist_;
};

// Role of mola::SearchablePoseList [U] (mola_pose_list) with measure_from_last_kf_only = false: the key-frame poses,
// queried for the closest one (LidarOdometry.cpp:1066-1118).
class SearchablePoseList {
 public:
  bool empty() const { return poses_.empty(); }
  size_t size() const { return poses_.size(); }
  void insert(const CPose3D& p) { poses_.push_back(p); }
  // {is_first, p (-) closest key-frame}
  std::pair<bool, CPose3D> check(const CPose3D& p) const;
  void removeAllFartherThan(const CPose3D& p, double max_dist);
  void clear() { poses_.clear(); }

 private:
  std::vector<CPose3D> poses_;
};

class LidarOdometry {
 public:
  // the params: block of the pipeline file (yaml:6-122); formulas are re-evaluated every scan
  struct Params : public Parameterizable {
    double min_time_between_scans = 0.2;
    double max_sensor_range_filter_coefficient = 0.999;
    double absolute_minimum_sensor_range = 5.0;
    bool optimize_twist = false;
    double optimize_twist_rerun_min_trans = 0.1, optimize_twist_rerun_min_rot_deg = 0.5;
    size_t optimize_twist_max_corrections = 8;
    bool local_map_updates_enabled = true;
    double min_translation_between_keyframes = 1.0, min_rotation_between_keyframes = 30.0;  // [m] [deg] formulas
    double max_distance_to_keep_keyframes = 0.0;                                            // formula
    uint32_t check_for_removal_every_n = 100;
    double min_icp_goodness = 0.4;
    bool adaptive_threshold_enabled = true;
    double initial_sigma = 0.5, min_motion = 0.1, maximum_sigma = 5.0, kp = 5.0, alpha = 0.99;
    bool validity_check_enabled = false;
    uint32_t validity_minimum_point_count = 1000;
    void load_from(const Config& c);
  };

  // one line of the per-scan log (what the reference spreads over debug traces / the parameter source)
  struct ScanRecord {
    double timestamp = 0;
    bool dropped = false, first_scan = false, icp_run = false, icp_good = false, had_motion_model = false,
         map_updated = false, restarted = false;
    CPose3D pose;             // state_.last_lidar_pose after this scan
    CPose3D init_guess;       // what ICP started from
    double goodness = 0, sigma = 0, estimated_sensor_max_range = 0, instantaneous_sensor_max_range = 0;
    uint32_t icp_iterations = 0, twist_corrections = 0, align_calls = 0;
    int termination = 0;
    // n_map_points / n_map_voxels: the map AFTER this scan.  In the reference returned by onLidar*() they are still 0 for a
    // key-frame scan (and for the scans up to the next read-back): the update is asynchronous; records() fills them in.
    uint64_t n_raw = 0, n_for_map = 0, n_for_icp = 0, n_map_points = 0, n_map_voxels = 0;
    Twist twist;              // twist used for the (last) de-skew of this scan
    double decim_map_resolution = 0, decim_icp_resolution = 0, map_voxel_size = 0;
  };

  // ctx == nullptr: the process-wide default device context, taken at initialize()
  explicit LidarOdometry(std::shared_ptr<DeviceContext> ctx = nullptr);
  ~LidarOdometry();

  // cfg = the whole pipeline file (keys: params, navstate_fuse_params, icp_settings_with_vel[, icp_settings_without_vel],
  // localmap_generator, observations_filter_adjust_timestamps, observations_filter_1st_pass, observations_filter_2nd_pass,
  // insert_observation_into_local_map).  Throws std::runtime_error on a filter chain the device path does not implement.
  void initialize(const Config& cfg);
  void reset();

  // One LiDAR observation: points in the vehicle frame, optional per-point time stamps [s] relative to `timestamp`.
  // Returns the record of this scan (also appended to records()).
  const ScanRecord& onLidar(double timestamp, const float* x, const float* y, const float* z, const float* t, size_t n);
  // The same for raw interleaved records as sensors and data sets deliver them (KITTI velodyne .bin: point_step 16,
  // offsets 0/4/8; PointCloud2: its own): the bytes go to the device as they are and are split into channels there
  // (the job of observations_generator, lidar3d-default.yaml:250-262).  Time stamps: float32 field at off_t (>= 0), or
  // the separate array `t` (may be nullptr).
  const ScanRecord& onLidarInterleaved(double timestamp, const void* data, size_t n, size_t point_step, size_t off_x,
                                       size_t off_y, size_t off_z, long long off_t = -1, const float* t = nullptr);

  // Off-line replay (data sets, eval/cli_kitti.sh): announce the NEXT observation before calling onLidar* for the
  // current one.  Its upload and first filter pass then run on a second stream of the same device, in a worker thread,
  // while the current scan is in its ICP loop; the next onLidar* call (same buffer, same layout) picks the result up.
  // The buffers must stay valid until that call.  Results are identical to the sequential flow: the first pass only
  // depends on the sensor-range estimate, which is final before ICP starts; if any filter parameter turns out
  // different when the scan is really due, the prepared layers are dropped and the pass runs again.
  // several sequences in one process: the ICP of this instance joins the others' in one lock-step batch per round
  // (mp2p_icp_hip::AlignBatcher); call after initialize(), drive every instance from its own host thread
  void setAlignBatcher(std::shared_ptr<mp2p_icp_hip::AlignBatcher> b);
  void prefetchInterleaved(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                           long long off_t = -1, const float* t = nullptr);
  void prefetch(const float* x, const float* y, const float* z, const float* t, size_t n);

  // (n_map_points / n_map_voxels of the records since the last key-frame are read back from the device lazily: here, and
  // before the next key-frame update -- by then the update has long finished, so the replay itself never waits for it)
  const std::vector<ScanRecord>& records() const { resolve_map_counts(); return records_; }
  const std::vector<std::pair<double, CPose3D>>& estimatedTrajectory() const { return trajectory_; }
  // TUM format "t x y z qx qy qz qw" (estimated_trajectory.output_file, yaml:79-81; eval/cli_kitti.sh:41-50)
  void saveTrajectoryTUM(const std::string& path) const;
  const Params& params() const { return params_; }
  std::shared_ptr<HashedVoxelPointCloud> localMap() const { return local_map_; }
  std::map<std::string, double> dynamicVariables() const { return source_.getVariableValues(); }
  // what initialize() recognised in the pipeline file (for tests / logs)
  std::map<std::string, std::string> describePipeline() const;
  // accumulated host wall time [s] per stage of onLidar (the role of the reference's profiler_ sections "onLidar.*")
  const std::map<std::string, double>& profile() const { return profile_; }
  void resetProfile() { profile_.clear(); }
  // The interleaved buffers handed to onLidarInterleaved / prefetchInterleaved are page-locked (mh_host_alloc_pinned) and
  // stay valid and unmodified until the scan AFTER the one they hold has been registered: uploads are then asynchronous.
  void setInputPinned(bool pinned) { input_pinned_ = pinned; }  // e.g. after the warm-up scans of a replay: steady-state stage times

 private:
  struct FilterPlan;  // the recognised observation filter chain, as data for mh_scan_preprocess / mh_scan_deskew
  struct RawInput;  // where the points of the current observation come from
  struct Prefetch;  // the announced next observation and its worker
  const ScanRecord& process(double timestamp, const RawInput& in);
  void launch_prefetch();
  void cancel_prefetch();
  void updatePipelineDynamicVariables();
  void updatePipelineTwistVariables(const Twist& tw);
  void run_first_pass();
  void run_second_pass();
  void doUpdateAdaptiveThreshold(const CPose3D& motionModelError);
  void create_local_map();
  void ensure_device();
  void resolve_map_counts() const;  // fills n_map_points / n_map_voxels of the records that still wait for them
  bool map_is_empty();              // local_map_->empty() without a device round trip once the map is known to hold points

  std::shared_ptr<DeviceContext> ctx_;
  Params params_;
  std::unique_ptr<FilterPlan> plan_;
  ParameterSource source_;
  NavStateFuse navstate_;
  ICP::Ptr icp_[2];  // [0] RegularOdometry, [1] NoMotionModel
  mp2p_icp_hip::Parameters icp_params_[2];
  Config map_def_;

  // observation layers (device)
  std::shared_ptr<DevicePointCloud> raw_, map_skewed_, icp_skewed_, for_map_, for_icp_;
  // prefetch: a second context (stream + scratch) used by the worker thread only, with two sets of raw / skewed layers
  // filled alternately (the current scan may still re-de-skew from its set while the next one is being prepared)
  std::shared_ptr<DeviceContext> ctx_b_;
  float icp_bb_min_[3] = {0, 0, 0}, icp_bb_max_[3] = {0, 0, 0};  // bounding box of for_icp_ (run_second_pass)
  std::shared_ptr<mp2p_icp_hip::AlignBatcher> batcher_;  // set: filters and alignments join those of the other sequences
  std::shared_ptr<DevicePointCloud> raw_b_[2], map_skewed_b_[2], icp_skewed_b_[2];
  std::shared_ptr<DevicePointCloud> cur_raw_, cur_map_skewed_, cur_icp_skewed_;  // the set the current scan reads
  std::unique_ptr<Prefetch> pf_;
  std::shared_ptr<HashedVoxelPointCloud> local_map_;
  float remove_voxels_farther_than_ = 0.f;
  double map_voxel_size_ = 0;

  // state_ (LidarOdometry.h: struct MethodState)
  CPose3D last_lidar_pose_;
  bool last_icp_was_good_ = true;
  double last_icp_quality_ = 0;
  std::optional<double> last_obs_tim_, last_icp_timestamp_, first_ever_timestamp_, last_obs_timestamp_;
  std::optional<NavStateFuse::NavState> last_motion_model_output_;
  double adapt_thres_sigma_ = 0;
  std::optional<double> estimated_sensor_max_range_, instantaneous_sensor_max_range_;
  SearchablePoseList distance_checker_local_map_;
  uint32_t localmap_check_removal_counter_ = 0;
  std::vector<std::pair<double, CPose3D>> trajectory_;
  mutable std::vector<ScanRecord> records_;
  mutable bool map_counts_pending_ = false;  // records_[map_counts_from_ ...] wait for the map's counters
  mutable size_t map_counts_from_ = 0;
  mutable uint64_t map_points_cached_ = 0, map_voxels_cached_ = 0;
  mutable bool map_known_nonempty_ = false;
  bool input_pinned_ = false;
  std::map<std::string, double> profile_;
};

}  // namespace mola_hip
