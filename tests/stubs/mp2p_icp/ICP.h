#pragma once  // stand-in for the mp2p_icp API the adapters use (tests/stubs/README.md): declarations only, from memory
#include <mrpt/core/optional_ref.h>
#include <mrpt/maps/CMetricMap.h>
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>
#include <mrpt/system/CTimeLogger.h>
#include <mrpt/tfest/TMatchingPair.h>
#include <mrpt/typemeta/TEnumType.h>
#include <cstdint>
#include <functional>
#include <map>
#include <optional>
#include <string>
#include <utility>
#include <vector>
namespace mp2p_icp {
using layer_name_t = std::string;
struct metric_map_t { std::map<layer_name_t, mrpt::maps::CMetricMap::Ptr> layers; };
enum class IterTermReason : uint8_t { Undefined = 0, NoPairings, SolverError, MaxIterations, Stalled, QualityCheckpointFailed, HookRequest };
enum class RobustKernel : uint8_t { None = 0, GemanMcClure, Cauchy };
struct plane_patch_t { mrpt::math::TPlane plane; mrpt::math::TPoint3Df centroid; };
struct point_plane_pair_t { plane_patch_t pl_global; mrpt::math::TPoint3Df pt_local; };
struct Pairings {
  mrpt::tfest::TMatchingPairList paired_pt2pt;
  std::vector<point_plane_pair_t> paired_pt2pl;
  std::vector<int> paired_pt2ln, paired_ln2ln, paired_pl2pl;
  std::vector<std::pair<std::size_t, double>> point_weights;
  uint64_t potential_pairings = 0;
};
struct LogRecord {};
struct Parameters { uint32_t maxIterations = 40; double minAbsStep_trans = 5e-4, minAbsStep_rot = 1e-4; bool generateDebugFiles = false; };
struct Results { mrpt::poses::CPose3DPDFGaussian optimal_tf; double quality = 0; size_t nIterations = 0; IterTermReason terminationReason = IterTermReason::Undefined; Pairings finalPairings; };
struct OptimalTF_Result { mrpt::poses::CPose3D optimalPose; double optimalScale = 1.0; };
struct SolverContext { std::optional<mrpt::poses::CPose3D> guessRelativePose; std::optional<uint32_t> icpIteration; std::optional<mrpt::poses::CPose3DPDFGaussianInf> prior; };
struct pointcloud_bitfield_t {
  struct bits { std::vector<bool> v; size_t size() const { return v.size(); } void resize(size_t n) { v.resize(n); } void mark_as_set(size_t i) { v[i] = true; }
                bool operator[](size_t i) const { return v[i]; } bool none() const { for (bool b : v) if (b) return false; return true; } };
  std::map<layer_name_t, bits> point_layers; };
struct MatchState { pointcloud_bitfield_t localPairedBitField, globalPairedBitField; };
struct MatchContext { uint32_t icpIteration = 0; };
class ParameterSource { public: void updateVariable(const std::string&, double) {} void realize() {} };
class Parameterizable { public: const std::vector<ParameterSource*>& attachedSources() const { return srcs_; } private: std::vector<ParameterSource*> srcs_; };
class Matcher : public mrpt::rtti::CObject, public Parameterizable { public: using Ptr = std::shared_ptr<Matcher>; uint32_t runFromIteration = 0, runUpToIteration = 0; bool enabled = true; };
class Solver : public mrpt::rtti::CObject, public Parameterizable { public: using Ptr = std::shared_ptr<Solver>;
 protected: virtual bool impl_optimal_pose(const Pairings&, OptimalTF_Result&, const SolverContext&) const = 0; };
class Matcher_Points_Base : public Matcher { public:
  std::map<std::string, std::map<std::string, double>> weight_pt2pt_layers;
  uint64_t maxLocalPointsPerLayer_ = 0, localPointsSampleSeed_ = 0;
  bool allowMatchAlreadyMatchedPoints_ = false, allowMatchAlreadyMatchedGlobalPoints_ = false;
 private:
  virtual void implMatchOneLayer(const mrpt::maps::CMetricMap&, const mrpt::maps::CPointsMap&, const mrpt::poses::CPose3D&, MatchState&, const layer_name_t&,
                                 const layer_name_t&, Pairings&) const = 0; };
class ICP : public mrpt::rtti::CObject, public Parameterizable {
  DEFINE_MRPT_OBJECT(ICP, mp2p_icp)
 public:
  struct IterationHook_Input { uint32_t currentIteration = 0; const metric_map_t* pcGlobal = nullptr; const metric_map_t* pcLocal = nullptr; const OptimalTF_Result* currentSolution = nullptr; };
  struct IterationHook_Output { bool request_stop = false; };
  using iteration_hook_t = std::function<IterationHook_Output(const IterationHook_Input&)>;
  virtual void align(const metric_map_t&, const metric_map_t&, const mrpt::math::TPose3D&, const Parameters&, Results&,
                     const std::optional<mrpt::poses::CPose3DPDFGaussianInf>& = std::nullopt, const mrpt::optional_ref<LogRecord>& = std::nullopt) {}
  const std::vector<Solver::Ptr>& solvers() const { return solvers_; }
  const std::vector<Matcher::Ptr>& matchers() const { return matchers_; }
  mrpt::system::CTimeLogger& profiler() { return profiler_; }
  void setIterationHook(const iteration_hook_t& h) { iteration_hook_ = h; }
 protected:
  std::vector<Solver::Ptr> solvers_; std::vector<Matcher::Ptr> matchers_; iteration_hook_t iteration_hook_; mrpt::system::CTimeLogger profiler_; };
}
