// Stand-ins for the point / box types of include/ikd-Tree/ikd_Tree.h (pcl::PointXYZINormal layout, 48 bytes).
#pragma once
#include <vector>
struct PointType { float x, y, z, pad0; float normal_x, normal_y, normal_z, pad1; float intensity, curvature, pad2, pad3; };
typedef std::vector<PointType> PointVector;
struct BoxPointType { float vertex_min[3]; float vertex_max[3]; };
struct StubCloud { std::vector<PointType> points; };
