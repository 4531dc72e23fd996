// TEST-ONLY declarations (see tests/stubs/README.md): the members of ORB_SLAM2::Map / MapPlane the SurfelMapping adapter touches.
#pragma once
#include <vector>
#include <opencv2/opencv.hpp>
#include <pcl/point_types.h>
#include "Surfel.h"
namespace ORB_SLAM2 {
class MapPlane {
public:
    cv::Mat GetWorldPos();
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr mvPlanePoints;
};
class Map {
public:
    std::vector<MapPlane *> GetAllMapPlanes();
    std::vector<Surfel> mvLocalSurfels;
    std::vector<Surfel> mvInactiveSurfels;
};
}  // namespace ORB_SLAM2
