// Stand-ins for the reference's Eigen/MTK/ROS types, just rich enough to type-check the adaptor of
// INTEGRATION.md against include/lio_hip.h (Eigen, PCL and ROS are not installed in the build image).
#pragma once
#include <cstddef>
#include <cstdio>
#include <stdexcept>
#include <vector>

struct StubVec3 { double v[3]; double operator[](int i) const { return v[i]; } };
struct StubQuat { double q[4]; double x() const { return q[0]; } double y() const { return q[1]; } double z() const { return q[2]; } double w() const { return q[3]; } };
struct state_ikfom { StubVec3 pos; StubQuat rot; StubQuat offset_R_L_I; StubVec3 offset_T_L_I; };

struct StubMatrix {
    std::vector<double> a; long r = 0, c = 0;
    // Eigen's resize() does not initialise (and re-allocates when the size changes); the stand-in keeps its allocation
    void resize(long rows, long cols) { r = rows; c = cols; a.resize((size_t)(rows * cols)); }
    void resize(long rows) { resize(rows, 1); }
    double* data() { return a.data(); }
    long rows() const { return r; }
    long cols() const { return c; }
};
namespace esekfom {
template <class T> struct dyn_share_datastruct { bool valid = true, converge = true; StubMatrix h_x, h; };
}
#define ROS_WARN(...) std::printf(__VA_ARGS__)
