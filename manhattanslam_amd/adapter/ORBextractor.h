/**
 * Drop-in replacement for ManhattanSLAM's include/ORBextractor.h (reference include/ORBextractor.h:43-110).
 *
 * Same namespace, class name, constructor, operator() and inline getters, so src/Frame.cc and src/Tracking.cc compile
 * and behave unchanged; the body forwards to the C ABI in include/msl.h (libmsl.so, hand-written HIP for gfx950).
 * Build this file and ORBextractor.cc INSTEAD of the reference's src/ORBextractor.cc and link with -lmsl.
 * Needs the OpenCV headers of the host project (they are not available in the build container of this repository,
 * so the adapters are compiled by the integrator).
 */
#ifndef ORBEXTRACTOR_H
#define ORBEXTRACTOR_H

#include <list>
#include <vector>

#include <opencv/cv.h>

#include "msl.h"

namespace ORB_SLAM2 {

class ORBextractor {
public:
    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image.  Mask is ignored (as in the reference).
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // Kept for source compatibility; the pyramid lives in HBM and is never mirrored to the host (no reference code
    // reads it: grep mvImagePyramid over src/ shows only ORBextractor.cc itself).
    std::vector<cv::Mat> mvImagePyramid;

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    msl_orb *mHandle;       // created lazily for the first frame size seen
    int mW, mH;
    std::vector<msl_keypoint> mKps;
    std::vector<unsigned char> mDesc;
    void ensureHandle(int w, int h);
};

}  // namespace ORB_SLAM2

#endif
