// TEST-ONLY stand-in (see tests/stubs/README.md): the slice of PCL the SurfelMapping adapter uses (a point cloud is a vector of points).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
struct PointXYZRGB { float x, y, z; std::uint8_t r, g, b; };
struct PointSurfel { float x, y, z, normal_x, normal_y, normal_z; std::uint8_t r, g, b; float radius, confidence, curvature; };
template <typename T> class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    std::vector<T> points;
    void push_back(const T &p) { points.push_back(p); }
};
}  // namespace pcl
