// TEST INFRASTRUCTURE ONLY (oracle/): include/IKFoM_toolkit/esekfom/esekfom.hpp:41 includes <opencv/cv.h> but uses
// nothing from it; this empty stand-in lets the filter compile where OpenCV is not installed.
#pragma once
