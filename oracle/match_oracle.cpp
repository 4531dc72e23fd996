// match_oracle.cpp -- CPU restatement of ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, th)
// (reference src/ORBmatcher.cc:547-678) with the helpers it calls: Frame::GetFeaturesInArea (src/Frame.cc:332-381),
// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:835-849) and ORBmatcher::ComputeThreeMaxima (:799-830).
// TEST INFRASTRUCTURE ONLY (see orb_oracle.cpp header): never linked into the product.
//
// PARITY UNPINNED: the reference has no tests or golden vectors for this path and needs OpenCV, so it cannot be built here.
// This file follows the reference statement by statement on plain arrays (MapPoint* -> index of the last-frame point, -1 = NULL).
// Pinned third-party arithmetic:
//   * cv::Mat expressions on CV_32F 3x3 / 3x1 operands (`Rcw * x3Dw + tcw`, `-Rcw.t() * tcw`, `Rlw * twc + tlw`,
//     :560-568, :577) lower to cv::gemm, whose float kernel (GEMMSingleMul<float, double>) accumulates the products and the
//     beta * C term in double and rounds once to float.
//   * `round` (:649) is the C library round on the float product promoted per the usual rules (half away from zero).
//   * a NaN projection (depth exactly 0 after the invzc < 0 test) passes the reference's bounds tests but finds no feature
//     (GetFeaturesInArea converts NaN to an out-of-range cell); here it is skipped at the bounds test.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../include/msl.h"

namespace {

const int TH_HIGH = 100, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:33-35
const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;   // include/Frame.h:53-54

// ORBmatcher::DescriptorDistance (:835-849)
int DescriptorDistance(const uint8_t *a, const uint8_t *b) {
    int32_t pa[8], pb[8];
    memcpy(pa, a, 32); memcpy(pb, b, 32);
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = pa[i] ^ pb[i];
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}

// ORBmatcher::ComputeThreeMaxima (:799-830)
void ComputeThreeMaxima(const std::vector<int> *histo, const int L, int &ind1, int &ind2, int &ind3) {
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = (int)histo[i].size();
        if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
        else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
        else if (s > max3) { max3 = s; ind3 = i; }
    }
    if (max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
    else if (max3 < 0.1f * (float)max1) { ind3 = -1; }
}

struct CurFrame {   // the members of Frame the matcher reads
    int N;
    const msl_keypoint *mvKeysUn_src;   // octave, angle (mvKeysUn[i] = mvKeys[i] with pt replaced, src/Frame.cc:456-460)
    const float *un_xy;                 // mvKeysUn[i].pt
    const float *mvuRight;
    const uint8_t *mDescriptors;
    std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    float fx, fy, cx, cy, mbf, mb, mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv;
    const float *mvScaleFactors;

    // Frame::GetFeaturesInArea (src/Frame.cc:332-381)
    std::vector<size_t> GetFeaturesInArea(const float &x, const float &y, const float &r, const int minLevel, const int maxLevel) const {
        std::vector<size_t> vIndices;
        const int nMinCellX = std::max(0, (int)floor((x - mnMinX - r) * mfGridElementWidthInv));
        if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
        const int nMaxCellX = std::min((int)FRAME_GRID_COLS - 1, (int)ceil((x - mnMinX + r) * mfGridElementWidthInv));
        if (nMaxCellX < 0) return vIndices;
        const int nMinCellY = std::max(0, (int)floor((y - mnMinY - r) * mfGridElementHeightInv));
        if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
        const int nMaxCellY = std::min((int)FRAME_GRID_ROWS - 1, (int)ceil((y - mnMinY + r) * mfGridElementHeightInv));
        if (nMaxCellY < 0) return vIndices;
        const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
        for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
            for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
                const std::vector<size_t> &vCell = mGrid[ix][iy];
                if (vCell.empty()) continue;
                for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
                    const int octave = mvKeysUn_src[vCell[j]].octave;
                    if (bCheckLevels) {
                        if (octave < minLevel) continue;
                        if (maxLevel >= 0)
                            if (octave > maxLevel) continue;
                    }
                    const float distx = un_xy[2 * vCell[j]] - x;
                    const float disty = un_xy[2 * vCell[j] + 1] - y;
                    if (std::fabs(distx) < r && std::fabs(disty) < r) vIndices.push_back(vCell[j]);
                }
            }
        return vIndices;
    }
};

// d = A(3x3, row-major, leading dimension lda) * b + c, the cv::gemm float kernel: double accumulation, one rounding
void gemm3(const float *A, int lda, bool transA, float alpha, const float *b, const float *c, float *d) {
    for (int r = 0; r < 3; r++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (double)(transA ? A[k * lda + r] : A[r * lda + k]) * (double)b[k];
        d[r] = (float)(s * (double)alpha + (c ? (double)c[r] : 0.0));
    }
}

}  // namespace

extern "C" {
#define MSLO_API __attribute__((visibility("default")))

// One SearchByProjection(CurrentFrame, LastFrame, th) call.  Tcw_*: 3x4 row-major [R | t] (rows 0-2 of the CV_32F 4x4 mTcw).
// last_flags bit 0: LastFrame.mvpMapPoints[i] != NULL && !mvbOutlier[i]; bit 1: pMP->Observations() > 0.
// match_out[N_cur]: index of the last-frame point whose MapPoint CurrentFrame.mvpMapPoints[i2] holds at return, -1 = NULL
// (all NULL on entry, as Tracking fills it, src/Tracking.cc:1252).  Returns nmatches.
MSLO_API int mslo_search_by_projection(const msl_match_params *P, int n_cur, const msl_keypoint *cur_kps, const float *cur_un_xy,
                                       const float *cur_uright, const int32_t *cur_grid_cell, const uint8_t *cur_desc, int n_last,
                                       const float *last_xyz, const uint8_t *last_desc, const uint8_t *last_flags, const int32_t *last_octave,
                                       const float *last_angle, const float *Tcw_cur, const float *Tcw_last, int32_t *match_out) {
    CurFrame CurrentFrame;
    CurrentFrame.N = n_cur; CurrentFrame.mvKeysUn_src = cur_kps; CurrentFrame.un_xy = cur_un_xy; CurrentFrame.mvuRight = cur_uright;
    CurrentFrame.mDescriptors = cur_desc;
    CurrentFrame.fx = P->fx; CurrentFrame.fy = P->fy; CurrentFrame.cx = P->cx; CurrentFrame.cy = P->cy; CurrentFrame.mbf = P->bf;
    CurrentFrame.mb = P->bf / P->fx;                                                        // src/Frame.cc:150
    CurrentFrame.mnMinX = P->minX; CurrentFrame.mnMaxX = P->maxX; CurrentFrame.mnMinY = P->minY; CurrentFrame.mnMaxY = P->maxY;
    CurrentFrame.mfGridElementWidthInv = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(P->maxX - P->minX);    // :137-138
    CurrentFrame.mfGridElementHeightInv = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(P->maxY - P->minY);
    CurrentFrame.mvScaleFactors = P->scale_factors;
    for (int i = 0; i < n_cur; i++)   // AssignFeaturesToGrid (:155-168): insertion in keypoint order
        if (cur_grid_cell[i] >= 0) CurrentFrame.mGrid[cur_grid_cell[i] / FRAME_GRID_ROWS][cur_grid_cell[i] % FRAME_GRID_ROWS].push_back((size_t)i);
    std::vector<int> mvpMapPoints(n_cur, -1);
    const float th = P->th;

    int nmatches = 0;
    std::vector<int> rotHist[HISTO_LENGTH];
    const float factor = 1.0f / HISTO_LENGTH;
    const float *Rcw = Tcw_cur, *tcwp = Tcw_cur + 3;   // row r of R at Tcw[4 r .. 4 r + 2], t[r] at Tcw[4 r + 3]
    const float tcw[3] = {Tcw_cur[3], Tcw_cur[7], Tcw_cur[11]};
    (void)tcwp;
    float twc[3];
    gemm3(Rcw, 4, true, -1.0f, tcw, nullptr, twc);                     // twc = -Rcw.t() * tcw
    const float tlw[3] = {Tcw_last[3], Tcw_last[7], Tcw_last[11]};
    float tlc[3];
    gemm3(Tcw_last, 4, false, 1.0f, twc, tlw, tlc);                    // tlc = Rlw * twc + tlw
    const bool bForward = tlc[2] > CurrentFrame.mb;
    const bool bBackward = -tlc[2] > CurrentFrame.mb;

    for (int i = 0; i < n_last; i++) {
        if (!(last_flags[i] & 1)) continue;                            // pMP && !mvbOutlier[i]
        float x3Dc[3];
        gemm3(Rcw, 4, false, 1.0f, last_xyz + 3 * i, tcw, x3Dc);       // x3Dc = Rcw * x3Dw + tcw
        const float xc = x3Dc[0];
        const float yc = x3Dc[1];
        const float invzc = (float)(1.0 / x3Dc[2]);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (!(u >= CurrentFrame.mnMinX && u <= CurrentFrame.mnMaxX)) continue;   // (NaN: see header)
        if (!(v >= CurrentFrame.mnMinY && v <= CurrentFrame.mnMaxY)) continue;
        int nLastOctave = last_octave[i];
        float radius = th * CurrentFrame.mvScaleFactors[nLastOctave];
        std::vector<size_t> vIndices2;
        if (bForward) vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nLastOctave, -1);
        else if (bBackward) vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, 0, nLastOctave);
        else vIndices2 = CurrentFrame.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
        if (vIndices2.empty()) continue;
        const uint8_t *dMP = last_desc + 32 * (size_t)i;
        int bestDist = 256;
        int bestIdx2 = -1;
        for (std::vector<size_t>::const_iterator vit = vIndices2.begin(), vend = vIndices2.end(); vit != vend; vit++) {
            const size_t i2 = *vit;
            if (mvpMapPoints[i2] >= 0)
                if (last_flags[mvpMapPoints[i2]] & 2) continue;        // ->Observations() > 0
            if (CurrentFrame.mvuRight[i2] > 0) {
                const float ur = u - CurrentFrame.mbf * invzc;
                const float er = std::fabs(ur - CurrentFrame.mvuRight[i2]);
                if (er > radius) continue;
            }
            const int dist = DescriptorDistance(dMP, CurrentFrame.mDescriptors + 32 * i2);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = (int)i2; }
        }
        if (bestDist <= TH_HIGH) {
            mvpMapPoints[bestIdx2] = i;
            nmatches++;
            if (P->check_orientation) {
                float rot = last_angle[i] - cur_kps[bestIdx2].angle;
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)round(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                rotHist[bin].push_back(bestIdx2);
            }
        }
    }
    if (P->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
        for (int i = 0; i < HISTO_LENGTH; i++)
            if (i != ind1 && i != ind2 && i != ind3)
                for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
                    mvpMapPoints[rotHist[i][j]] = -1;
                    nmatches--;
                }
    }
    for (int i = 0; i < n_cur; i++) match_out[i] = mvpMapPoints[i];
    return nmatches;
}

MSLO_API int mslo_descriptor_distance(const uint8_t *a, const uint8_t *b) { return DescriptorDistance(a, b); }

}  // extern "C"
