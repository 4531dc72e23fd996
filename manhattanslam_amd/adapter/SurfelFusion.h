/**
 * Drop-in replacement for ManhattanSLAM's include/SurfelFusion.h (reference include/SurfelFusion.h:43-139): same
 * global-namespace class, constructor and fuseInitializeMap signature, so src/SurfelMapping.cpp compiles unchanged.
 * Build with SurfelFusion.cpp INSTEAD of the reference's src/SurfelFusion.cpp and link with -lmsl.
 *
 * fuseInitializeMap() is the host-vector mode (the caller's std::vector<Surfel> travels over PCIe on every call).
 * For a device-resident map use fuseMapResident(), which also performs the slot refill / tail compaction of
 * SurfelMapping::fuseMap (src/SurfelMapping.cpp:366-391) on the GPU -- see INTEGRATION.md for the three-line change
 * in SurfelMapping::fuseMap that enables it.
 */
#ifndef SURFEL_FUSION_H
#define SURFEL_FUSION_H

#include <Eigen/Eigen>
#include <opencv2/opencv.hpp>
#include <vector>

#include <Surfel.h>

#include "msl.h"

class SurfelFusion {
public:
    SurfelFusion(int width, int height, float _fx, float _fy, float _cx, float _cy, float _fuseFar, float _fuseNear);
    ~SurfelFusion();
    SurfelFusion(const SurfelFusion &) = delete;
    SurfelFusion &operator=(const SurfelFusion &) = delete;

    void fuseInitializeMap(const int referenceFrameIndex, const cv::Mat &inputImage, const cv::Mat &inputDepth,
                           const cv::Mat &inputPlaneMembershipImg, const Eigen::Matrix4f &pose,
                           std::vector<Surfel> &localSurfels, std::vector<Surfel> &newSurfels);

    // Device-resident variants (not in the reference): the map stays in HBM between keyframes.
    void uploadMap(const std::vector<Surfel> &localSurfels);
    void downloadMap(std::vector<Surfel> &localSurfels);
    void fuseMapResident(const int referenceFrameIndex, const cv::Mat &inputImage, const cv::Mat &inputDepth,
                         const cv::Mat &inputPlaneMembershipImg, const Eigen::Matrix4f &pose);
    // Waits for the enqueued keyframes and throws if a device-side bound was exceeded (errors of the asynchronous
    // resident calls are deferred to the next sync).
    void sync();
    void localSurfelsUnchangedSinceLastCall() { mLocalUnchanged = true; }   // hint for the NEXT fuseInitializeMap (MSL_SF_LOCAL_UNCHANGED)
    msl_sf *handle() const { return mHandle; }   // for adapter/SurfelMapping.cpp (map maintenance on the resident map)

private:
    msl_sf *mHandle;
    bool mLocalUnchanged = false;
    int imageWidth, imageHeight;
};

#endif  // SURFEL_FUSION_H
