// TEST INFRASTRUCTURE ONLY (oracle/): stand-in for <pcl/point_types.h> for the `ref_model` build, where the REAL Eigen
// is on the include path (so, unlike oracle/shim/pcl/point_types.h, nothing is declared in namespace Eigen).
// Field offsets mirror PCL's 16-byte-aligned PODs; PointXYZINormal (48 bytes) is `PointType` on the hot path
// (include/common_lib.h:161).  If the real PCL is installed, put it on the include path instead of this directory.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
namespace pcl {
struct alignas(16) PointXYZ { float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f; };
struct alignas(16) PointXYZI { float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f; float intensity = 0.f, _pad1[3] = {0.f, 0.f, 0.f}; };
struct alignas(16) PointXYZINormal {
    float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f;
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, _pad1 = 0.f;
    float intensity = 0.f, curvature = 0.f, _pad2[2] = {0.f, 0.f};
};
static_assert(sizeof(PointXYZINormal) == 48, "PointXYZINormal must be 48 bytes");
}  // namespace pcl
