// TEST-ONLY stand-in (see tests/stubs/README.md): the members of ORB_SLAM2::Map / MapPlane the SurfelMapping adapter touches.
#pragma once
#include <vector>
#include <opencv2/opencv.hpp>
#include <pcl/point_types.h>
#include "Surfel.h"
namespace ORB_SLAM2 {
class MapPlane {
public:
    cv::Mat GetWorldPos() { return worldPos; }
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr mvPlanePoints;
    cv::Mat worldPos;
};
class Map {
public:
    std::vector<MapPlane *> GetAllMapPlanes() { return planes; }
    std::vector<Surfel> mvLocalSurfels;
    std::vector<Surfel> mvInactiveSurfels;
    std::vector<MapPlane *> planes;
};
}  // namespace ORB_SLAM2
