// Adapter: SurfelFusion on top of the msl C ABI (replaces the reference's src/SurfelFusion.cpp).
#include "SurfelFusion.h"

#include <stdexcept>
#include <string>
#include <cstdlib>

// The GPU this process uses: the reference has no notion of a device, so the adapters take it from the environment (MSL_DEVICE, default 0);
// one process per GPU (bench.py's ranks, a multi-sequence server) sets it per process.
static int msl_device_from_env() { const char *e = std::getenv("MSL_DEVICE"); return e ? std::atoi(e) : 0; }


static_assert(sizeof(msl_surfel) == sizeof(Surfel), "msl_surfel must mirror struct Surfel (include/Surfel.h)");

static void check(int rc, const char *what) {
    if (rc != MSL_OK) throw std::runtime_error(std::string(what) + ": " + msl_last_error());
}

SurfelFusion::SurfelFusion(int width, int height, float _fx, float _fy, float _cx, float _cy, float _fuseFar, float _fuseNear)
    : mHandle(msl_sf_create(width, height, _fx, _fy, _cx, _cy, _fuseFar, _fuseNear, msl_device_from_env())), imageWidth(width),
      imageHeight(height) {
    if (!mHandle) throw std::runtime_error(std::string("msl_sf_create: ") + msl_last_error());
}

SurfelFusion::~SurfelFusion() { msl_sf_destroy(mHandle); }

void SurfelFusion::fuseInitializeMap(const int referenceFrameIndex, const cv::Mat &inputImage, const cv::Mat &inputDepth,
                                     const cv::Mat &inputPlaneMembershipImg, const Eigen::Matrix4f &pose,
                                     std::vector<Surfel> &localSurfels, std::vector<Surfel> &newSurfels) {
    // inputImage: CV_8UC1 (Tracking passes mImGray), inputDepth: CV_32FC1 metres, membership: CV_32SC1 half resolution
    const size_t cap = (size_t)(imageWidth / 8) * (imageHeight / 8);
    newSurfels.resize(cap);                                       // :289 clears it; at most one surfel per superpixel
    size_t nNew = 0;
    // A caller that has not touched localSurfels since the previous call may say so (localSurfelsUnchangedSinceLastCall(): one-shot, an
    // extension the reference does not have): the 56 bytes per surfel are then not uploaded again.  Only touched stretches come back either way.
    const unsigned flags = mLocalUnchanged ? MSL_SF_LOCAL_UNCHANGED : 0u;
    mLocalUnchanged = false;
    check(msl_sf_fuse_ex(mHandle, referenceFrameIndex, inputImage.data, inputImage.step, inputDepth.ptr<float>(), inputDepth.step,
                         inputPlaneMembershipImg.ptr<int32_t>(), inputPlaneMembershipImg.step, pose.data() /* column-major */,
                         reinterpret_cast<msl_surfel *>(localSurfels.data()), localSurfels.size(),
                         reinterpret_cast<msl_surfel *>(newSurfels.data()), cap, &nNew, flags),
          "msl_sf_fuse_ex");
    newSurfels.resize(nNew);
}

void SurfelFusion::uploadMap(const std::vector<Surfel> &localSurfels) {
    check(msl_sf_map_upload(mHandle, reinterpret_cast<const msl_surfel *>(localSurfels.data()), localSurfels.size()), "msl_sf_map_upload");
}

void SurfelFusion::downloadMap(std::vector<Surfel> &localSurfels) {
    size_t n = 0;
    check(msl_sf_map_size(mHandle, &n), "msl_sf_map_size");
    localSurfels.resize(n);
    check(msl_sf_map_download(mHandle, reinterpret_cast<msl_surfel *>(localSurfels.data()), n, &n), "msl_sf_map_download");
}

void SurfelFusion::fuseMapResident(const int referenceFrameIndex, const cv::Mat &inputImage, const cv::Mat &inputDepth,
                                   const cv::Mat &inputPlaneMembershipImg, const Eigen::Matrix4f &pose) {
    check(msl_sf_fuse_resident(mHandle, referenceFrameIndex, inputImage.data, inputImage.step, inputDepth.ptr<float>(),
                               inputDepth.step, inputPlaneMembershipImg.ptr<int32_t>(), inputPlaneMembershipImg.step, MSL_MEM_HOST,
                               pose.data()),
          "msl_sf_fuse_resident");
}

void SurfelFusion::sync() { check(msl_sf_sync(mHandle), "msl_sf_sync"); }
