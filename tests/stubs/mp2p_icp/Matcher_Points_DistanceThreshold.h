#pragma once  // stand-in
#include <mp2p_icp/ICP.h>
namespace mp2p_icp {
class Matcher_Points_DistanceThreshold : public Matcher_Points_Base {
  DEFINE_MRPT_OBJECT(Matcher_Points_DistanceThreshold, mp2p_icp)
 public:
  double threshold = 0.5, thresholdAngularDeg = 0; uint32_t pairingsPerPoint = 1;
  bool allowMatchAlreadyMatchedGlobalPoints = false;  // (spelled without the underscore in some versions)
 protected:
  void implMatchOneLayer(const mrpt::maps::CMetricMap&, const mrpt::maps::CPointsMap&, const mrpt::poses::CPose3D&, MatchState&, const layer_name_t&,
                         const layer_name_t&, Pairings&) const override {} };
}
