#pragma once  // stand-in
#include <mrpt/maps/CPointsMap.h>
#include <mrpt/obs/CObservation.h>
namespace mrpt::obs { class CObservationPointCloud : public CObservation { public: mrpt::maps::CPointsMap::Ptr pointcloud; mrpt::poses::CPose3D sensorPose; }; }
