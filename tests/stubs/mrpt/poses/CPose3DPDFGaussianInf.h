#pragma once
#include <mrpt/poses/CPose3D.h>
