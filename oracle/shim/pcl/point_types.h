// TEST INFRASTRUCTURE ONLY (oracle/): minimal stand-in for <pcl/point_types.h> so that the
// reference's include/ikd-Tree/ikd_Tree.{h,cpp} compile unmodified where they lie under
// /root/reference (PCL and Eigen are not installed in this image).
//
// ikd_Tree.h:11 includes <pcl/point_types.h>; ikd_Tree.h:62 uses Eigen::aligned_allocator;
// ikd_Tree.cpp:1460-1463 explicitly instantiate KD_TREE for pcl::PointXYZ / PointXYZI /
// PointXYZINormal (and ikdTree_PointType).  Field offsets mirror PCL's 16-byte-aligned PODs
// (x,y,z,pad | normal_xyz,pad | intensity,curvature,pad) = 48 bytes for PointXYZINormal, which
// is `PointType` on the hot path (include/common_lib.h:161).
#pragma once
#include <memory>
#include <cstdint>
#include <cstring>   // the real PCL header pulls these in transitively (ikd_Tree.cpp:122 uses memset)
#include <vector>
#include <cmath>

namespace pcl {
struct alignas(16) PointXYZ {
    float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f;
};
struct alignas(16) PointXYZI {
    float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f;
    float intensity = 0.f, _pad1[3] = {0.f, 0.f, 0.f};
};
struct alignas(16) PointXYZINormal {
    float x = 0.f, y = 0.f, z = 0.f, _pad0 = 1.f;
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, _pad1 = 0.f;
    float intensity = 0.f, curvature = 0.f, _pad2[2] = {0.f, 0.f};
};
static_assert(sizeof(PointXYZINormal) == 48, "PointXYZINormal must be 48 bytes");
}  // namespace pcl

namespace Eigen {
template <class T> using aligned_allocator = std::allocator<T>;
}
