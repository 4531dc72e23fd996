// Adapter: ORB_SLAM2::ORBextractor on top of the msl C ABI (replaces the reference's src/ORBextractor.cc).
#include "ORBextractor.h"

#include <cassert>
#include <cstring>
#include <stdexcept>
#include <string>
#include <cstdlib>

// The GPU this process uses: the reference has no notion of a device, so the adapters take it from the environment (MSL_DEVICE, default 0);
// one process per GPU (bench.py's ranks, a multi-sequence server) sets it per process.
static int msl_device_from_env() { const char *e = std::getenv("MSL_DEVICE"); return e ? std::atoi(e) : 0; }


namespace ORB_SLAM2 {

static_assert(sizeof(msl_keypoint) == sizeof(cv::KeyPoint), "msl_keypoint must mirror cv::KeyPoint");

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST),
      mHandle(nullptr), mW(0), mH(0) {
    mvImagePyramid.resize(nlevels);
    // The scale tables are needed before the first frame (Frame.cc:80-86 reads them right after construction):
    // same float arithmetic as src/ORBextractor.cc:416-430.
    mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
        mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    for (int i = 0; i < nlevels; i++) {
        mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
        mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
}

ORBextractor::~ORBextractor() { msl_orb_destroy(mHandle); }

void ORBextractor::ensureHandle(int w, int h) {
    if (mHandle && w <= mW && h <= mH) return;
    msl_orb_destroy(mHandle);
    mHandle = msl_orb_create(nfeatures, (float)scaleFactor, nlevels, iniThFAST, minThFAST, w, h, /*max_batch=*/1, msl_device_from_env());
    if (!mHandle) throw std::runtime_error(std::string("msl_orb_create: ") + msl_last_error());
    mW = w; mH = h;
    const int cap = msl_orb_capacity(mHandle);
    mKps.resize(cap); mDesc.resize((size_t)cap * 32);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint> &_keypoints,
                              cv::OutputArray _descriptors) {
    if (_image.empty()) return;                       // src/ORBextractor.cc:815-816
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);                  // :819
    ensureHandle(image.cols, image.rows);
    int n = 0;
    const int rc = msl_orb_extract(mHandle, image.data, image.cols, image.rows, image.step, mKps.data(), mDesc.data(),
                                   (int)mKps.size(), &n);
    if (rc != MSL_OK) throw std::runtime_error(std::string("msl_orb_extract: ") + msl_last_error());
    _keypoints.resize(n);
    if (n == 0) { _descriptors.release(); return; }   // :832-833
    std::memcpy(static_cast<void *>(_keypoints.data()), mKps.data(), sizeof(cv::KeyPoint) * n);
    _descriptors.create(n, 32, CV_8U);
    cv::Mat d = _descriptors.getMat();
    for (int i = 0; i < n; i++) std::memcpy(d.ptr(i), &mDesc[(size_t)i * 32], 32);
}

}  // namespace ORB_SLAM2
