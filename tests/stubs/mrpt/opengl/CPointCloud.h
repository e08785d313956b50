#pragma once  // stand-in
#include <memory>
#include <vector>
namespace mrpt::opengl {
class CRenderizable { public: virtual ~CRenderizable() = default; };
class CPointCloud : public CRenderizable { public: using Ptr = std::shared_ptr<CPointCloud>; static Ptr Create() { return std::make_shared<CPointCloud>(); }
  void setAllPoints(const std::vector<float>&, const std::vector<float>&, const std::vector<float>&) {} };
class CSetOfObjects { public: void insert(const std::shared_ptr<CRenderizable>&) {} };
}
