// Drop-in for the reference header of the same path; see ../mcgpu/dropin.hpp.
#pragma once
#include "../mcgpu/dropin.hpp"
