#pragma once  // stand-in
#include <mp2p_icp/ICP.h>
namespace mp2p_icp {
class Matcher_Point2Plane : public Matcher_Points_Base {
  DEFINE_MRPT_OBJECT(Matcher_Point2Plane, mp2p_icp)
 public:
  double distanceThreshold = 0.5, searchRadius = 1.0, planeEigenThreshold = 0.01; uint32_t knn = 5, minimumPlanePoints = 5;
 protected:
  void implMatchOneLayer(const mrpt::maps::CMetricMap&, const mrpt::maps::CPointsMap&, const mrpt::poses::CPose3D&, MatchState&, const layer_name_t&,
                         const layer_name_t&, Pairings&) const override {} };
}
