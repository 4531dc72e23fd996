// Drop-in replacement for the reference's include/PlaneExtractor.h: the class PlaneDetection with the members the rest of ManhattanSLAM
// touches (src/Frame.cc:606-640 reads plane_num_, plane_vertices_, cloud.vertices / verticesColour and plane_filter.extractedPlanes[i]->normal /
// ->center; src/Tracking.cc:228, 497 clones plane_filter.membershipImg), backed by msl_peac_extract_batch (include/msl.h) instead of the
// header-only PEAC library under include/peac/.  ahc::PlaneSeg and ahc::PlaneFitter shrink to the data members those call sites read.
#ifndef PLANEEXTRACTOR_H
#define PLANEEXTRACTOR_H

#include <memory>
#include <vector>
#include "opencv2/opencv.hpp"
#include <Eigen/Eigen>
#include "msl.h"
#include <cstdlib>

inline int msl_device_env() { const char *e = std::getenv("MSL_DEVICE"); return e ? std::atoi(e) : 0; }

typedef Eigen::Vector3d VertexType;
typedef cv::Vec3d VertexColour;

struct ImagePointCloud {   // include/PlaneExtractor.h:38-55
    std::vector<VertexType> vertices;   // 3D vertices
    std::vector<VertexColour> verticesColour;
    int w, h;
    inline int width() const { return w; }
    inline int height() const { return h; }
};

namespace ahc {
struct PlaneSeg {   // the members of include/peac/AHCPlaneSeg.hpp:50-125 a consumer of extractedPlanes reads
    typedef std::shared_ptr<PlaneSeg> shared_ptr;
    double normal[3], center[3], mse;
    int N;
};
template <class Image3D>
struct PlaneFitter {   // include/peac/AHCPlaneFitter.hpp:111-161: outputs and the parameters PlaneDetection could change
    cv::Mat membershipImg;                             // CV_32SC1, plane id >= 0, -1, or a region-growing visit counter -2..-6
    std::vector<PlaneSeg::shared_ptr> extractedPlanes;
    msl_peac_params params;                            // ahc::ParamSet + minSupport, windowWidth/Height, doRefine, erodeType
    PlaneFitter() { msl_peac_default_params(&params); }
};
}  // namespace ahc

class PlaneDetection {
public:
    ImagePointCloud cloud;
    ahc::PlaneFitter<ImagePointCloud> plane_filter;
    std::vector<std::vector<int>> plane_vertices_;   // vertex indices each plane contains
    cv::Mat seg_img_;                                // segmentation image (allocated, not coloured: nothing in the reference reads it)
    cv::Mat color_img_;                              // input color image
    int plane_num_;

public:
    PlaneDetection();
    ~PlaneDetection();
    bool readColorImage(cv::Mat RGBImg);
    bool readDepthImage(const cv::Mat depthImg, const cv::Mat &K, const float &depthMapFactor);
    void runPlaneDetection();

private:
    cv::Mat depth16_;   // the raw depth image of the last readDepthImage
    float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0, depthMapFactor_ = 0;
    int device_ = msl_device_env();   // MSL_DEVICE (default 0)
};

#endif  // PLANEEXTRACTOR_H
