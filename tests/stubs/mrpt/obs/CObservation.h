#pragma once  // stand-in
#include <mrpt/rtti/CObject.h>
namespace mrpt::obs { class CObservation : public mrpt::rtti::CObject { public: virtual ~CObservation() = default; }; }
