#pragma once  // stand-in
#include <mp2p_icp/ICP.h>
namespace mp2p_icp {
class Solver_GaussNewton : public Solver {
  DEFINE_MRPT_OBJECT(Solver_GaussNewton, mp2p_icp)
 public:
  uint32_t maxIterations = 2; RobustKernel robustKernel = RobustKernel::None; double robustKernelParam = 1.0;
  struct PairWeights { double pt2pt = 1.0, pt2ln = 1.0, pt2pl = 1.0, ln2ln = 1.0, pl2pl = 1.0; } pairWeights;  // [U] (stand-in: scaffolding, pins nothing)
 protected:
  bool impl_optimal_pose(const Pairings&, OptimalTF_Result&, const SolverContext&) const override { return false; } };
}
