#pragma once  // stand-in
#include <cstdint>
#include <vector>
namespace mrpt::serialization {
class CArchive { public:
  template <class T> CArchive& operator<<(const T&) { return *this; }
  template <class T> CArchive& operator>>(T&) { return *this; } };
class CSerializable : public mrpt::rtti::CObject { protected:
  virtual uint8_t serializeGetVersion() const = 0; virtual void serializeTo(CArchive&) const = 0; virtual void serializeFrom(CArchive&, uint8_t) = 0; };
}
