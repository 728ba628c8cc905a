// Compile-check stand-in (TEST INFRASTRUCTURE): lets the -DORBFE_HAVE_OPENCV branch of csrc/host/*.h,*.cc be syntax-checked
// in an image without OpenCV.  It forwards to cvlite.h, which spells the type codes as global macros exactly like OpenCV.
#pragma once
#include "../../../../../refactored_orb_slam2_amd/csrc/host/cvlite.h"
