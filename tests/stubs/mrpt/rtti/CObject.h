#pragma once  // stand-in
#include <memory>
#include <mrpt/core/exceptions.h>
namespace mrpt::rtti {
struct TRuntimeClassId { const char* className; };
class CObject { public: using Ptr = std::shared_ptr<CObject>; virtual ~CObject() = default; virtual const TRuntimeClassId* GetRuntimeClass() const { return nullptr; } };
inline void registerClass(const TRuntimeClassId*) {}
}
#define DEFINE_MRPT_OBJECT(cls, ns) public: using Ptr = std::shared_ptr<cls>; static const mrpt::rtti::TRuntimeClassId runtimeClassId; \
  const mrpt::rtti::TRuntimeClassId* GetRuntimeClass() const override { return &runtimeClassId; } static std::shared_ptr<cls> Create() { return std::make_shared<cls>(); } private:
#define IMPLEMENTS_MRPT_OBJECT(cls, base, ns) const mrpt::rtti::TRuntimeClassId cls::runtimeClassId = {#ns "::" #cls};
#define DEFINE_SERIALIZABLE(cls, ns) DEFINE_MRPT_OBJECT(cls, ns) protected: uint8_t serializeGetVersion() const override; \
  void serializeTo(mrpt::serialization::CArchive& out) const override; void serializeFrom(mrpt::serialization::CArchive& in, uint8_t version) override; private:
#define IMPLEMENTS_SERIALIZABLE(cls, base, ns) IMPLEMENTS_MRPT_OBJECT(cls, base, ns)
#define CLASS_ID(T) (&T::runtimeClassId)
