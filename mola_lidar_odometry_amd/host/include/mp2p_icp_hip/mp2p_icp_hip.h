// mp2p_icp_hip.h -- C++17 host layer above the C ABI (include/molahip.h).
//
// It mirrors, name for name and argument for argument, the part of the mp2p_icp plugin API [U] that
// mola::LidarOdometry drives for the ICP hot path, so that code written against the reference's interface
// reads the same here (the reference's own dependencies -- mp2p_icp, MRPT -- are absent from this image, so
// this layer is self-contained; the adapter that derives from the REAL mp2p_icp classes is
// host/adapters/mp2p_icp_plugin.cpp):
//
//   reference (file:line in /root/reference)                       here
//   -------------------------------------------------------------  ------------------------------------------
//   mp2p_icp::icp_pipeline_from_yaml   LidarOdometry.cpp:115-122    mp2p_icp_hip::icp_pipeline_from_yaml
//   mp2p_icp::ICP::align               LidarOdometry.cpp:961-962    ICP::align (same argument order)
//   ICP::setIterationHook              LidarOdometry.cpp:923-952    ICP::setIterationHook / setDeviceHook
//   mp2p_icp::Parameters               lidar3d-default.yaml:172-182 Parameters
//   mp2p_icp::Results / IterTermReason LidarOdometry.cpp:1009-1019  Results / IterTermReason
//   Matcher_Points_DistanceThreshold   lidar3d-default.yaml:195-204 same name
//   Solver_GaussNewton                 lidar3d-default.yaml:184-190 same name
//   QualityEvaluator_PairedRatio       lidar3d-default.yaml:206-209 same name
//   ParameterSource / Parameterizable  LidarOdometry.cpp:356,1571-1635  same names (run-time formulas)
//   mola::HashedVoxelPointCloud        lidar3d-default.yaml:228-242 HashedVoxelPointCloud (device resident)
//
// Nothing here computes on the CPU: every numeric step is a call into libmolahip.
#pragma once
#include <cstdint>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <chrono>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "molahip.h"

namespace mp2p_icp_hip {

// ---------------------------------------------------------------- poses (mrpt::poses stand-ins)
struct TPose3D {
  double x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0;
  static TPose3D Identity() { return {}; }
};
struct CPose3D {
  double T[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};  // row-major [R|t]
  CPose3D() = default;
  explicit CPose3D(const TPose3D& p);
  TPose3D asTPose() const;
  CPose3D operator+(const CPose3D& b) const;  // composition a (+) b
  CPose3D operator-(const CPose3D& b) const;  // inverse composition: b^-1 (+) a
  double translationNorm() const;
  double rotationAngle() const;  // |SO(3) log|
  void so3Log(double w[3]) const;  // mrpt::poses::Lie::SO<3>::log of the rotation part
  // (Exp_SO3(w), t): the pose a constant twist reaches after one unit of time in FilterDeskew's model
  static CPose3D FromRotVecAndTranslation(const double w[3], const double t[3]);
};
struct CPose3DPDFGaussian {
  CPose3D mean;
  double cov[36] = {0};  // (x,y,z,yaw,pitch,roll)
};
struct CPose3DPDFGaussianInf {
  CPose3D mean;
  double cov_inv[36] = {0};
};

// ---------------------------------------------------------------- tiny config tree (YAML subset)
class Config {
 public:
  enum class Kind { Null, Scalar, Map, Seq };
  Kind kind = Kind::Null;
  std::string scalar;
  std::vector<std::pair<std::string, Config>> map;
  std::vector<Config> seq;

  static Config FromYamlText(const std::string& text);  // comments, nesting, "- " sequences, {flow maps}, ${ENV|default}
  static Config FromYamlFile(const std::string& path);
  bool has(const std::string& key) const;
  const Config& operator[](const std::string& key) const;  // throws if absent
  const Config& at(size_t i) const { return seq.at(i); }
  size_t size() const { return kind == Kind::Seq ? seq.size() : map.size(); }
  std::string asString() const { return scalar; }
  bool isNull() const { return kind == Kind::Null || (kind == Kind::Scalar && (scalar == "~" || scalar.empty())); }
  std::string getOr(const std::string& key, const std::string& def) const;
};

// ---------------------------------------------------------------- run-time formulas (mp2p_icp::Parameterizable)
double evaluate_expression(const std::string& expr, const std::map<std::string, double>& vars);

// A formula compiled once to a small stack program.  ICP::align evaluates the matcher / solver formulas for every
// ICP_ITERATION of a call (up to 300 of them, lidar3d-default.yaml:172): re-parsing the text each time cost more
// host time than the device needed for the whole alignment of a decimated scan.
class CompiledExpression {
 public:
  explicit CompiledExpression(const std::string& text);
  const std::vector<std::string>& variables() const { return vars_; }
  // values[i] = pointer to the current value of variables()[i]
  double evaluate(const std::vector<const double*>& values) const;
  double evaluate(const std::map<std::string, double>& vars) const;

 private:
  struct Op {
    int code;  // 0 const, 1 var, 2 add, 3 sub, 4 mul, 5 div, 6 neg, 7 pow, 8 call
    int arg = 0, nargs = 0;
    double value = 0;
  };
  std::vector<Op> prog_;
  std::vector<std::string> vars_;
  std::string text_;
  friend struct ExprCompiler;
};

class Parameterizable;
class ParameterSource {
 public:
  void updateVariable(const std::string& name, double value) { vars_[name] = value; }
  void updateVariables(const std::map<std::string, double>& v) {
    for (auto& kv : v) vars_[kv.first] = kv.second;
  }
  void attach(Parameterizable& p) { attached_.push_back(&p); }
  void realize();  // re-evaluates every declared formula of every attached object
  const std::map<std::string, double>& getVariableValues() const { return vars_; }

 private:
  std::map<std::string, double> vars_;
  std::vector<Parameterizable*> attached_;
};

class Parameterizable {
 public:
  virtual ~Parameterizable() = default;
  void attachToParameterSource(ParameterSource& s) { s.attach(*this); }
  void realizeWith(const std::map<std::string, double>& vars);
  struct Declared {
    std::string name, expr;
    double* target;
    std::shared_ptr<CompiledExpression> compiled;
  };
  const std::vector<Declared>& declaredParameters() const { return declared_; }
  // every formula bound to the entries of `vars` (which must outlive the binding and keep its keys): realize() then
  // only reads through the pointers, so a caller can sweep one variable in place (ICP_ITERATION) at ~0.1 us per formula
  struct Binding {
    std::vector<std::pair<const Declared*, std::vector<const double*>>> items;
    void realize() const;
  };
  Binding bind(const std::map<std::string, double>& vars) const;

 protected:
  void declareParameter(const std::string& name, const std::string& expr, double* target) {
    declared_.push_back({name, expr, target, std::make_shared<CompiledExpression>(expr)});
  }
  // DECLARE_PARAMETER_IN_REQ: the YAML value may be a number or a formula string
  void parameterFromConfig(const Config& c, const std::string& name, double* target, bool required);

 private:
  std::vector<Declared> declared_;
};

// ---------------------------------------------------------------- device handles
class DeviceContext {  // one HIP device + stream (mh_ctx)
 public:
  explicit DeviceContext(int device = 0);
  ~DeviceContext();
  DeviceContext(const DeviceContext&) = delete;
  mh_ctx* get() const { return ctx_; }
  int device() const { return device_; }
  void synchronize() const;
  static std::shared_ptr<DeviceContext> Default();  // process-wide context on device 0

 private:
  mh_ctx* ctx_ = nullptr;
  int device_ = 0;
};

[[noreturn]] void throw_status(mh_status s, const char* where);
inline void check(mh_status s, const char* where) {
  if (s != MH_OK) throw_status(s, where);  // std::runtime_error: what the reference catches at LidarOdometry.cpp:614-619
}

// ---------------------------------------------------------------- map layers
struct Layer {
  virtual ~Layer() = default;
};
// mrpt::maps::CPointsMap stand-in: host SoA, untransformed
struct PointCloud : Layer {
  std::vector<float> x, y, z;
  size_t size() const { return x.size(); }
  void insertPoint(float px, float py, float pz) { x.push_back(px); y.push_back(py); z.push_back(pz); }
};
// a point layer that lives on the device (mh_scan): what the device-side filters produce and align() consumes
// without any host copy
class DeviceContext;
class DevicePointCloud : public Layer {
 public:
  explicit DevicePointCloud(std::shared_ptr<DeviceContext> ctx);
  ~DevicePointCloud() override;
  DevicePointCloud(const DevicePointCloud&) = delete;
  size_t size() const;
  mh_scan* handle() const { return scan_; }
  void setPoints(const float* x, const float* y, const float* z, size_t n);
  // interleaved records (KITTI .bin, PointCloud2 payload): float32 x/y/z [and time stamp when off_t >= 0] at byte offsets
  // pinned: `data` is page-locked memory (mh_host_alloc_pinned) that stays valid and unmodified until the context's
  // stream has passed the copy -- the upload is then asynchronous (MH_MEM_HOST_PINNED, include/molahip.h)
  void setPointsInterleaved(const void* data, size_t n, size_t point_step, size_t off_x, size_t off_y, size_t off_z,
                            long long off_t = -1, bool pinned = false);
  void setTimestamps(const float* t, size_t n);
  void boundingBox(float mn[3], float mx[3]) const;
  void download(std::vector<float>& x, std::vector<float>& y, std::vector<float>& z) const;

 private:
  std::shared_ptr<DeviceContext> ctx_;
  mh_scan* scan_ = nullptr;
};
// mola::HashedVoxelPointCloud stand-in, device resident (NearestNeighborsCapable role only)
class HashedVoxelPointCloud : public Layer {
 public:
  HashedVoxelPointCloud(float voxel_size, uint32_t max_points_per_voxel,
                        std::shared_ptr<DeviceContext> ctx = DeviceContext::Default());
  // full parameter set (also what the NDT subclass uses)
  HashedVoxelPointCloud(const mh_map_params& p, std::shared_ptr<DeviceContext> ctx);
  ~HashedVoxelPointCloud() override;
  void setPoints(const float* x, const float* y, const float* z, size_t n);  // clear + insertPoint for each
  void insertPoints(const float* x, const float* y, const float* z, size_t n);  // keeps a host copy, rebuilds
  // FilterMerge + insertPointCloud + far-voxel removal, all on the device (mh_map_insert; lidar3d-default.yaml:362-368)
  void insertPointCloud(const DevicePointCloud& pc, const CPose3D& robot_pose, float remove_voxels_farther_than);
  void clear();
  size_t size() const;
  size_t voxelCount() const;
  mh_map* handle() const { return map_; }
  const std::shared_ptr<DeviceContext>& context() const { return ctx_; }

 private:
  std::shared_ptr<DeviceContext> ctx_;
  mh_map* map_ = nullptr;
};
// mola::NDT stand-in (lidar3d-ndt.yaml:236-254): the same device map plus per-voxel mean / covariance / eigen
// statistics, i.e. additionally NearestPlaneCapable for Matcher_Point2Plane
class NDT : public HashedVoxelPointCloud {
 public:
  NDT(float voxel_size, uint32_t max_points_per_voxel, float min_distance_between_points, float max_eigen_ratio_for_planes,
      std::shared_ptr<DeviceContext> ctx = DeviceContext::Default());
  size_t planeCount() const;
};
struct metric_map_t {
  std::map<std::string, std::shared_ptr<Layer>> layers;
};

// ---------------------------------------------------------------- pairings / results
struct Pairings {  // mp2p_icp::Pairings::paired_pt2pt as SoA (+ pt2pl)
  std::vector<uint32_t> localIdx, globalIdx;
  std::vector<float> lx, ly, lz, gx, gy, gz, errSq;
  // point-to-plane: local point, plane centroid, plane normal
  std::vector<float> pl_lx, pl_ly, pl_lz, pl_cx, pl_cy, pl_cz, pl_nx, pl_ny, pl_nz;
  size_t potential_pairings = 0;
  // role of Pairings::point_weights [U]: the `weight` of the pointLayerMatches entry that produced the point pairs (yaml:203-204).
  // The device solver takes ONE weight per kind of pair: layers with different weights in one pairing set are refused.
  double pt2pt_weight = 1.0, pt2pl_weight = 1.0;
  bool pt2pt_weight_set = false, pt2pl_weight_set = false;
  // role of MatchState::localPairedBitField [U]: per local layer, which points the matchers of THIS iteration have paired so
  // far -- read by later matchers when MOLA_HIP_MATCHED_POINTS=skip (allowMatchAlreadyMatchedPoints = false upstream, U12)
  std::map<std::string, std::vector<uint8_t>> local_paired;
  bool empty() const { return localIdx.empty() && pl_lx.empty(); }
  size_t size() const { return localIdx.size() + pl_lx.size(); }
};
enum class IterTermReason { Undefined = 0, NoPairings, SolverError, MaxIterations, Stalled, QualityCheckpointFailed, HookRequest };
const char* enum2str(IterTermReason r);
enum class RobustKernel { None = 0, GemanMcClure = 1, GemanMcClure_KISS = 2, GemanMcClure_Barron = 3, Cauchy = 4, GemanMcClure_C2 = 5 };

struct Parameters {  // mp2p_icp::Parameters (lidar3d-default.yaml:172-182)
  uint32_t maxIterations = 40;
  double minAbsStep_trans = 5e-4;
  double minAbsStep_rot = 1e-4;
  // MP2P_ICP_GENERATE_DEBUG_FILES (lidar3d-default.yaml:177-182): every align() writes its per-iteration trace.  The
  // upstream .icplog is an MRPT-serialised LogRecord and cannot be produced without MRPT; the file written here is
  // JSON with the same content an icp-log-viewer session starts from: initial guess, per-iteration pose / pairings /
  // thresholds, termination, final pose, quality.
  bool generateDebugFiles = false;
  std::string debugFileNameFormat = "icp-logs/icp-run-$UNIQUE_ID.icplog.json";  // $UNIQUE_ID = counter
  void load_from(const Config& c);
};
struct Results {
  CPose3DPDFGaussian optimal_tf;
  double quality = 0;
  size_t nIterations = 0;
  IterTermReason terminationReason = IterTermReason::Undefined;
  Pairings finalPairings;
};
struct OptimalTF_Result {
  CPose3D optimalPose;
};
struct MatchContext {
  uint32_t icpIteration = 0;
};
struct SolverContext {
  std::optional<CPose3D> guessRelativePose;
  std::optional<CPose3DPDFGaussianInf> prior;
  uint32_t icpIteration = 0;
};

// ---------------------------------------------------------------- plugins
class Matcher : public Parameterizable {
 public:
  using Ptr = std::shared_ptr<Matcher>;
  uint32_t runFromIteration = 0, runUpToIteration = 0;  // 0 = no limit
  bool enabled = true;
  virtual void initialize(const Config& params) = 0;
  // returns false if the matcher did not run for this iteration
  bool match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose, const MatchContext& mc,
             Pairings& out) const;

 protected:
  virtual void impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose,
                          const MatchContext& mc, Pairings& out) const = 0;
};

class Matcher_Points_DistanceThreshold : public Matcher {
 public:
  double threshold = 0.5;
  double thresholdAngularDeg = 0;
  uint32_t pairingsPerPoint = 1;
  bool allowMatchAlreadyMatchedGlobalPoints = true;
  struct LayerMatch {
    std::string global, local;
    double weight = 1.0;
  };
  std::vector<LayerMatch> pointLayerMatches;
  void initialize(const Config& params) override;

 protected:
  void impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose,
                  const MatchContext& mc, Pairings& out) const override;
};

// mp2p_icp::Matcher_Point2Plane [U]: against an NDT global layer (lidar3d-ndt.yaml:195-200) the per-voxel planes; against a plain
// HashedVoxelPointCloud layer (rgbd.yaml:143-151) the k nearest neighbours + PCA with the upstream knobs knn,
// planeEigenThreshold, minimumPlanePoints, searchRadius (round 5: mh_nn_search_pt2pl_knn; knn <= MH_MAX_PLANE_KNN).
class Matcher_Point2Plane : public Matcher {
 public:
  double distanceThreshold = 0.5;
  double planeEigenThreshold = 0.01, searchRadius = 1.0;  // [U] defaults; every pipeline that uses them sets them
  uint32_t knn = 5, minimumPlanePoints = 5;
  bool allowMatchAlreadyMatchedGlobalPoints = true;
  std::vector<Matcher_Points_DistanceThreshold::LayerMatch> pointLayerMatches;
  void initialize(const Config& params) override;

 protected:
  void impl_match(const metric_map_t& pcGlobal, const metric_map_t& pcLocal, const CPose3D& localPose,
                  const MatchContext& mc, Pairings& out) const override;
};

class Solver : public Parameterizable {
 public:
  using Ptr = std::shared_ptr<Solver>;
  virtual void initialize(const Config& params) = 0;
  virtual bool optimal_pose(const Pairings& p, OptimalTF_Result& out, const SolverContext& sc) const = 0;
};

class Solver_GaussNewton : public Solver {
 public:
  uint32_t maxIterations = 10;  // inner Gauss-Newton steps (lidar3d-default.yaml:187)
  RobustKernel robustKernel = RobustKernel::None;
  double robustKernelParam = 1.0;
  double minDelta = 1e-7, maxCost = 0;
  void initialize(const Config& params) override;
  bool optimal_pose(const Pairings& p, OptimalTF_Result& out, const SolverContext& sc) const override;
};

class QualityEvaluator_PairedRatio {
 public:
  double evaluate(const Pairings& pairingsFromICP) const {
    return pairingsFromICP.potential_pairings ? double(pairingsFromICP.size()) / double(pairingsFromICP.potential_pairings) : 0.0;
  }
};

// ---------------------------------------------------------------- several sequences on one GPU, one process
// N sequences = N LidarOdometry instances = N independent align() streams (eval/cli_kitti.sh:23-36 runs them as N
// processes).  One alignment of a decimated scan is a chain of short dependent kernels that leaves the device mostly
// idle, so the instances' alignments are merged: every instance runs on its own host thread, and where it would call
// mh_icp_align it hands its request to the batcher and blocks; when ALL active participants are waiting, the last one
// to arrive runs ONE mh_icp_align_batch over the requests (per-job parameters: every sequence has its own adaptive
// threshold, iteration budget, hook check point; jobs with the same kernel chain advance in lock step) and wakes the
// others.  Results are bitwise those of separate alignments, so every sequence's records are those of a solo run.
class AlignBatcher {
 public:
  explicit AlignBatcher(size_t participants);
  // blocks until the batch this request joined has run; returns its status (the message in `error` when not MH_OK).
  // `owner`: any address that identifies the participant (the same one in every call it makes; nullptr: the scan)
  mh_status align(const void* owner, const mh_map* map, const mh_scan* scan, const mh_icp_params* params, const double T_guess[12],
                  const mh_prior* prior, mh_icp_result* result, std::string* error);
  // The observation filters of the participants' NEXT scans (mh_scan_preprocess) merged the same way, in sets of their
  // own: a sequence's filter request for scan k+1 and its alignment of scan k are in flight together.  Which requests
  // belong together is bookkeeping, not timing.  A participant announces a request on the thread that will align next
  // (announceFilter: the set is the number of alignments it has requested so far) and a worker of its own delivers it
  // (preprocess).  A participant is PAST set j once it has requested its (j+1)-th alignment or announced a request for
  // set j; set j is complete when every active participant is past it and every announced request has arrived --
  // which is exactly when the lock-step alignment j+1 can start, so the filter batch runs beside it.  Participants that
  // re-align a scan, skip a scan or end early never make the others wait, and nobody has to say "not this time".
  // Whichever waiting worker sees its set complete runs ONE mh_scan_preprocess_batch over it (never the aligning
  // thread: the alignment must not wait for the filters).  A request that has waited MOLA_HIP_FILTER_SET_WAIT_US
  // (default 2 ms) runs with what waits by then: a net under the bookkeeping.
  size_t announceFilter(const void* owner);
  mh_status preprocess(const void* owner, size_t set, const mh_scan* raw, const mh_preprocess_params* params, mh_scan* out_map,
                       mh_scan* out_icp, std::string* error);
  void cancelAnnouncedFilter(const void* owner, size_t set);  // the announced request will not come (its upload failed)
  // Participation as a scope: leave() when the object goes away, however the sequence ended (an exception between
  // construction and the first align() must not leave the others waiting for ever).
  class Membership {
   public:
    explicit Membership(std::shared_ptr<AlignBatcher> b) : b_(std::move(b)) {}
    ~Membership() {
      if (b_) b_->leave();
    }
    Membership(const Membership&) = delete;
    Membership& operator=(const Membership&) = delete;

   private:
    std::shared_ptr<AlignBatcher> b_;
  };
  // this participant will not align any more (end of its sequence, or it failed)
  void leave();
  // ... and its bookkeeping entry goes with it (a driver calls this from its destructor)
  void forgetOwner(const void* owner);
  size_t filterBatches() const { return n_pp_batches_; }
  size_t filterJobs() const { return n_pp_jobs_; }
  size_t waitTimeouts() const { return n_wait_timeouts_; }  // batches led by a request that had waited MOLA_HIP_BATCH_WAIT_US
  size_t filterTimeouts() const { return n_pp_timeouts_; }  // sets forced because a request had waited MOLA_HIP_FILTER_SET_WAIT_US
  size_t batches() const { return n_batches_; }
  size_t jobs() const { return n_jobs_; }
  // where a batch's wall time goes: from the first request of a batch to its start (the sequences' other phases), and the
  // mh_icp_align_batch call itself; seconds, summed over the batches so far
  double secondsAssembling() const { return t_assemble_; }
  double secondsRunning() const { return t_run_; }

 private:
  struct Request {
    const mh_map* map = nullptr;
    const mh_scan* scan = nullptr;
    const mh_icp_params* params = nullptr;
    const double* T = nullptr;
    const mh_prior* prior = nullptr;
    mh_icp_result* result = nullptr;
    mh_status status = MH_OK;
    std::string error;
    bool done = false;
    bool lead = false;  // woken to run the batch of those waiting (the request that made it due ran on its own)
    std::chrono::steady_clock::time_point arrived{};
  };
  struct FilterRequest {
    const mh_scan* raw = nullptr;
    const mh_preprocess_params* params = nullptr;
    mh_scan* out_map = nullptr;
    mh_scan* out_icp = nullptr;
    size_t set = 0;
    mh_status status = MH_OK;
    std::string error;
    bool done = false, taken = false;
  };
  struct OwnerState {
    size_t aligns = 0;          // alignments requested so far
    bool announced = false;     // ... and the latest set it announced a filter request for
    size_t announced_set = 0;
    bool free_running = false;  // its latest alignment was issued on its own (a one-launch loop): its filter requests run at once, alone
  };
  static constexpr size_t kNoFilterSet = (size_t)-1;  // announceFilter's answer for a free-running participant
  void run_filter_batch(std::vector<FilterRequest*>& batch);  // called WITHOUT the mutex
  bool filter_set_ready_locked(size_t set) const;
  void take_filter_sets_upto(std::unique_lock<std::mutex>& lk, size_t set);
  std::vector<FilterRequest*> pp_waiting_;
  std::map<const void*, OwnerState> owners_;
  std::map<size_t, size_t> pp_pending_;  // set -> announced requests that have not arrived yet
  size_t pp_set_ = 0;                    // sets below this one have been taken
  size_t n_pp_batches_ = 0, n_pp_jobs_ = 0, n_pp_timeouts_ = 0, n_wait_timeouts_ = 0;
  std::chrono::steady_clock::time_point last_solo_{};  // when the latest alignment issued on its own arrived
  void run_batch(std::vector<Request*>& batch);  // called WITHOUT the mutex
  void take_waiting_locked(std::vector<Request*>& batch);  // the ONLY way out of waiting_: clears every taken request's `lead`
  bool batch_due_locked() const;
  size_t threshold_locked() const;
  std::mutex mtx_;
  std::condition_variable cv_;
  std::vector<Request*> waiting_;
  size_t active_;
  size_t in_flight_ = 0;  // requests inside running batches
  size_t split_ = 1;      // batches the active participants are spread over (MOLA_HIP_BATCH_SPLIT; 1 = one batch of all, the
                          // default: two or more batches in flight measured SLOWER -- 8 sequences 2590 -> 2160 scans/s -- the
                          // host threads then contend for the HIP runtime, which is what limits this runner in the first place)
  bool solo_ = false;     // MOLA_HIP_BATCH_SOLO: no batches at all, every request runs as a single alignment at once (A/B switch)
  bool no_solo_ = false;  // MOLA_HIP_BATCH_NO_SOLO: the library's hint (mh_icp_align_prefers_solo) is not asked: everything is batched
  size_t n_batches_ = 0, n_jobs_ = 0;
  double t_assemble_ = 0.0, t_run_ = 0.0;
};

// ---------------------------------------------------------------- ICP
class ICP {
 public:
  using Ptr = std::shared_ptr<ICP>;
  struct IterationHook_Input {
    uint32_t currentIteration = 0;
    const OptimalTF_Result* currentSolution = nullptr;
  };
  struct IterationHook_Output {
    bool request_stop = false;
  };
  using iteration_hook_t = std::function<IterationHook_Output(const IterationHook_Input&)>;

  // ctx == nullptr: the process-wide default context is taken lazily at the first align() (so that pipelines
  // can be built and inspected on a box without a GPU; computing always needs one)
  explicit ICP(std::shared_ptr<DeviceContext> ctx = nullptr);
  ~ICP();

  // Same argument order as the call at LidarOdometry.cpp:961-962.
  void align(const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const TPose3D& initialGuessLocalWrtGlobal,
             const Parameters& p, Results& result, const std::optional<CPose3DPDFGaussianInf>& prior = std::nullopt);

  // arbitrary host callback: forces the matcher/solver-granular loop (one host round trip per iteration) ...
  void setIterationHook(const iteration_hook_t& hook) { iteration_hook_ = hook; }
  // ... unless replay is on: the fused device loop runs with a per-iteration trace, the hook is replayed on the traced
  // poses and a requested stop is reproduced by a second run with that budget (molahip_host/hook_replay.h: what the
  // mp2p_icp adapter does, because it only ever sees an opaque std::function)
  void setHookReplay(bool v) { hook_replay_ = v; }
  // fused alignments (without trace / final pairings / host hook) go through the batcher instead of mh_icp_align
  // `owner`: the participant this ICP object aligns for (a driver with two ICP objects passes its own address to both)
  void setAlignBatcher(std::shared_ptr<AlignBatcher> b, const void* owner = nullptr) {
    batcher_ = std::move(b);
    batch_owner_ = owner ? owner : this;
  }
  // the in-tree hook (LidarOdometry.cpp:923-952) as data: evaluated on the device inside the fused loop
  void setDeviceHook(double min_trans, double min_rot_rad, const CPose3D& checkpoint);
  void clearHooks();

  std::vector<Matcher::Ptr>& matchers() { return matchers_; }
  std::vector<Solver::Ptr>& solvers() { return solvers_; }
  void attachToParameterSource(ParameterSource& s);
  void initialize_matchers(const Config& seq);
  void initialize_solvers(const Config& seq);
  bool lastAlignUsedFusedPath() const { return last_fused_; }
  uint32_t lastAlignHostPolls() const { return last_polls_; }              // fused path: host waits for the device loop
  uint32_t lastAlignEnqueuedIterations() const { return last_enqueued_; }  // ... and iterations worth of kernels enqueued
  double lastAlignSetupSeconds() const { return last_setup_seconds_; }     // host time before the device call (schedules, parameters)
  // false: Results::finalPairings stays empty in the fused path (the odometry driver never reads it)
  void setKeepFinalPairings(bool v) { keep_pairings_ = v; }
  void forceGenericPath(bool v) { force_generic_ = v; }
  // evaluate the per-iteration thresholds of the fused path NOW, on the variables' current values; the next align()
  // re-uses them if the variables its formulas read still have these values (anything else: evaluated again, as before)
  void precomputeSchedule(uint32_t n_iterations);

 private:
  bool can_fuse() const;
  void align_fused(const PointCloud* local, const DevicePointCloud* dev_local, const HashedVoxelPointCloud& global,
                   const CPose3D& guess, const Parameters& p, Results& result,
                   const std::optional<CPose3DPDFGaussianInf>& prior);
  void align_generic(const metric_map_t& pcLocal, const metric_map_t& pcGlobal, const CPose3D& guess, const Parameters& p,
                     Results& result, const std::optional<CPose3DPDFGaussianInf>& prior);
  void realize_iteration(uint32_t k);
  void prepare_schedule(uint32_t n_iterations);
  struct Schedule {
    bool valid = false;
    std::vector<double> key;  // values of the variables the formulas read (ICP_ITERATION = 0)
    std::vector<double> thr, kp, plthr;
  } sched_;

  std::shared_ptr<DeviceContext> ctx_;
  std::vector<Matcher::Ptr> matchers_;
  std::vector<Solver::Ptr> solvers_;
  QualityEvaluator_PairedRatio quality_;
  iteration_hook_t iteration_hook_;
  bool dev_hook_ = false;
  double dev_hook_trans_ = 0, dev_hook_rot_ = 0;
  CPose3D dev_hook_chk_;
  ParameterSource* source_ = nullptr;
  ParameterSource own_source_;
  mh_scan* scan_ = nullptr;                 // staging layer for host point clouds handed to the fused path ...
  std::shared_ptr<DeviceContext> scan_ctx_;  // ... and the (map's) context it lives in, kept alive until ~ICP has destroyed it
  bool last_fused_ = false, force_generic_ = false, keep_pairings_ = true, hook_replay_ = false;
  std::shared_ptr<AlignBatcher> batcher_;
  const void* batch_owner_ = nullptr;
  // how long the previous call of each kind ran: [0] calls with the full iteration budget, [1] re-entries after a hook
  // request (LidarOdometry.cpp:956-967 re-enters with what is left of it) -- the first chunk of the device loop is sized by it
  uint32_t full_budget_ = 0, last_iterations_[2] = {0, 0}, last_polls_ = 0, last_enqueued_ = 0;
  double last_setup_seconds_ = 0;
};

// class factory by name (mrpt::rtti::classFactory stand-in): "mp2p_icp::X" and "mp2p_icp_hip::X" both resolve
Matcher::Ptr create_matcher(const std::string& class_name);
Solver::Ptr create_solver(const std::string& class_name);

// mp2p_icp::icp_pipeline_from_yaml: expects the keys class_name, params, solvers, matchers, quality
// (the icp_settings_with_vel block of pipelines/lidar3d-default.yaml:162-209)
std::tuple<ICP::Ptr, Parameters> icp_pipeline_from_yaml(const Config& icpParams, std::shared_ptr<DeviceContext> ctx = nullptr);

// The MOLA_HIP_* switches (molahip_host/plugin_switches.h) as THIS library sees them: the header's cache is one per shared
// object, so a test module that changes the environment has to ask the library to read it again, not its own copy.
void reload_plugin_switches();
uint32_t plugin_switch_matched_points();

}  // namespace mp2p_icp_hip
