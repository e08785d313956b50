#pragma once  // stand-in
#include <mrpt/math/types.h>
#include <mrpt/obs/CObservation.h>
#include <mrpt/opengl/CPointCloud.h>
#include <mrpt/poses/CPose3D.h>
#include <mrpt/rtti/CObject.h>
#include <mrpt/serialization/CArchive.h>
#include <optional>
#include <string>
namespace mrpt::maps {
class CMetricMap : public mrpt::serialization::CSerializable { public:
  using Ptr = std::shared_ptr<CMetricMap>;
  virtual bool isEmpty() const = 0;
  virtual mrpt::math::TBoundingBoxf boundingBox() const = 0;
  virtual std::string asString() const = 0;
  virtual void getVisualizationInto(mrpt::opengl::CSetOfObjects&) const = 0;
  virtual void saveMetricMapRepresentationToFile(const std::string&) const = 0;
 protected:
  virtual void internal_clear() = 0;
  virtual bool internal_insertObservation(const mrpt::obs::CObservation&, const std::optional<const mrpt::poses::CPose3D>&) = 0;
  virtual double internal_computeObservationLikelihood(const mrpt::obs::CObservation&, const mrpt::poses::CPose3D&) const = 0; };
}
