// CPU oracle (TEST INFRASTRUCTURE ONLY -- never linked into libmsl.so) for SURVEY.md 8(f) rank 2: literal restatement of
//   PlaneDetection::readDepthImage   /root/reference/src/PlaneExtractor.cpp:44-76
//   ImagePointCloud::get             /root/reference/include/PlaneExtractor.h:47-55
//   ahc::depthDisContinuous          /root/reference/include/peac/AHCPlaneSeg.hpp:41-43, ParamSet::T_dz AHCParamSet.hpp:140-142
//   ahc::PlaneSeg::PlaneSeg (init)   /root/reference/include/peac/AHCPlaneSeg.hpp:237-285, Stats::push :81-92
//   block loop of initGraph          /root/reference/include/peac/AHCPlaneFitter.hpp:756-776
// Parity unpinned: PEAC needs OpenCV / Eigen / boost and cannot be built in this image.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../include/msl.h"

#define MSLO_API extern "C" __attribute__((visibility("default")))

namespace {
struct Cloud {
    std::vector<double> v;   // x y z per vertex
    int w, h;
    bool get(int row, int col, double &x, double &y, double &z) const {
        const int pixIdx = row * w + col;
        z = v[3 * (size_t)pixIdx + 2];
        if (z == 0 || std::isnan(z)) return false;
        x = v[3 * (size_t)pixIdx]; y = v[3 * (size_t)pixIdx + 1];
        return true;
    }
};
}  // namespace

MSLO_API void mslo_peac_block_stats(const uint16_t *depth, size_t strideBytes, int cols, int rows, float fx, float fy, float cx, float cy,
                                    float depthMapFactor, int winWidth, int winHeight, double depthAlpha, double depthChangeTol, int initLoose,
                                    double *cloudOut, msl_peac_stats *statsOut) {
    Cloud cloud;
    const double width = std::ceil(cols / 2.0), height = std::ceil(rows / 2.0);
    cloud.w = (int)width; cloud.h = (int)height;
    cloud.v.assign((size_t)(height * width) * 3, 0.0);
    int vertex_idx = 0;
    for (int i = 0; i < rows; i += 2)
        for (int j = 0; j < cols; j += 2) {
            const uint16_t d = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(depth) + (size_t)i * strideBytes + 2 * (size_t)j);
            double z = (double)d * depthMapFactor;
            double x = ((double)j - cx) * z / fx;
            double y = ((double)i - cy) * z / fy;
            cloud.v[3 * (size_t)vertex_idx] = x; cloud.v[3 * (size_t)vertex_idx + 1] = y; cloud.v[3 * (size_t)vertex_idx + 2] = z;
            vertex_idx++;
        }
    if (cloudOut) for (size_t k = 0; k < cloud.v.size(); k++) cloudOut[k] = cloud.v[k];
    const int imgWidth = cloud.w, imgHeight = cloud.h;
    const int Nh = imgHeight / winHeight, Nw = imgWidth / winWidth;
    auto discontinuous = [&](double d0, double d1) { return std::fabs(d0 - d1) > depthAlpha * std::fabs(d0) + depthChangeTol; };
    for (int bi = 0; bi < Nh; ++bi)
        for (int bj = 0; bj < Nw; ++bj) {
            const int seed_row = bi * winHeight, seed_col = bj * winWidth;
            msl_peac_stats S{};
            bool windowValid = true;
            int nanCnt = 0, nanCntTh = winHeight * winWidth / 2;
            for (int i = seed_row, icnt = 0; icnt < winHeight && i < imgHeight; ++i, ++icnt) {
                for (int j = seed_col, jcnt = 0; jcnt < winWidth && j < imgWidth; ++j, ++jcnt) {
                    double x = 0, y = 0, z = 10000;
                    if (!cloud.get(i, j, x, y, z)) {
                        if (initLoose) {
                            ++nanCnt;
                            if (nanCnt < nanCntTh) continue;
                        }
                        windowValid = false;
                        break;
                    }
                    double xn = 0, yn = 0, zn = 10000;
                    if (j + 1 < imgWidth && (cloud.get(i, j + 1, xn, yn, zn) && discontinuous(z, zn))) { windowValid = false; break; }
                    if (i + 1 < imgHeight && (cloud.get(i + 1, j, xn, yn, zn) && discontinuous(z, zn))) { windowValid = false; break; }
                    S.sx += x; S.sy += y; S.sz += z;
                    S.sxx += x * x; S.syy += y * y; S.szz += z * z;
                    S.sxy += x * y; S.syz += y * z; S.sxz += x * z;
                    ++S.N;
                }
                if (!windowValid) break;
            }
            if (!windowValid) { S = msl_peac_stats{}; S.nouse = 1; }
            statsOut[bi * Nw + bj] = S;
        }
}
