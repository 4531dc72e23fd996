// msl_surfel.hip -- superpixel surfel fusion for gfx950 (MI355X): kernels + C ABI.
//
// Replaces SurfelFusion (reference src/SurfelFusion.cpp) and the slot refill / tail compaction of
// SurfelMapping::fuseMap (src/SurfelMapping.cpp:353-392).  Per keyframe:
//
//   k_seed_init                      one thread per 8x8 superpixel seed                  (:528-584)
//   3 x { k_assign                   one thread per pixel: argmin over <= 9 seeds        (:333-415)
//         [k_prop x R, k_commit_px]  raster-order `stable` semantics as a min-fixpoint   (App. B.7.1)
//         k_update_seeds             one wave per seed: ordered window gather, Huber mean (:428-515)
//         k_commit_seeds }           chunk-abort (`return`) semantics                    (App. B.7.2)
//   k_seed_plane                     one wave per seed: back-projection, pixel normals, Huber plane
//                                    fit with FP64 4x4 normal equations                  (:91-165, :597-773)
//   k_fuse                           one thread per live surfel, SoA map resident in HBM (:167-283)
//   k_new_surfels                    ordered emission of un-fused seeds                  (:285-331)
//   k_compact_*                      deleted-slot refill + tail compaction               (SurfelMapping.cpp:366-391)
//
// HBM-bound integer/float streaming; no MFMA.  Every float expression keeps the reference's
// evaluation order and float/double promotions; compiled with -ffp-contract=off.

#include "msl_common.h"

#include <algorithm>
#include <cmath>
#include <vector>

using namespace msl;

namespace {

constexpr int SP = 8;
constexpr int NCHUNK = 10;  // THREAD_NUM, include/SurfelFusion.h:34
constexpr double MAX_ANGLE_COS = 0.1, HUBER_RANGE = 0.4, BASELINE_D = 0.5, DISPARITY_ERROR = 4.0, MIN_TOLERATE_DIFF = 0.1;
constexpr unsigned T_INF = 0xFFFFFFFFu;
constexpr int PROP_ROUNDS = 6;
constexpr int MAP_GRID = 2048;  // grid-stride launches over the resident map

// Structure-of-arrays surfel map (device resident): 14 arrays of `cap` 4-byte elements.
struct MapSoA {
    float *px, *py, *pz, *nx, *ny, *nz, *size, *color;
    int *r, *g, *b;
    float *weight;
    int *updateTimes, *lastUpdate;
};

struct SfDev {
    int W, H, spW, spH, nseeds;
    float fx, fy, cx, cy, fuseFar, fuseNear;
    const uint8_t *gray; unsigned long long gstride, gbytes;
    const float *depth; unsigned long long dstride;   // floats
    const int32_t *member; unsigned long long mstride;  // ints
    msl_seed *seeds, *seedsTmp;
    int *index, *amap;
    unsigned *tmin;
    int *chunkAbort;      // [NCHUNK]
    int *changed;         // [PROP_ROUNDS+1]
    float pose[16], invPose[16];
    int ref;
    MapSoA map;
    unsigned long long cap;
    // device-side scalars: [0]=n_live, [1]=n_new, [2]=n_deleted, [3]=n_updated, [4]=n_before, [5]=err
    long long *ctr;
    msl_surfel *newSurfels;
    unsigned *blockSums;  // scan partials
    unsigned *delList;    // ascending deleted indices
    unsigned *srcOf;      // tail compaction: source position per low hole
};

__device__ __forceinline__ int seed_chunk(int seedI, int nseeds) {   // THREAD_NUM partition of :430-434
    const int step = nseeds / NCHUNK;
    if (step == 0) return NCHUNK - 1;
    const int c = seedI / step;
    return c > NCHUNK - 1 ? NCHUNK - 1 : c;
}
__device__ __forceinline__ uint8_t gray_at(const SfDev &P, int y, int x) { return P.gray[(size_t)y * P.gstride + x]; }
__device__ __forceinline__ float depth_at(const SfDev &P, int y, int x) { return P.depth[(size_t)y * P.dstride + x]; }
__device__ __forceinline__ float depth_flat(const SfDev &P, int idx) { return P.depth[(size_t)(idx / P.W) * P.dstride + (idx % P.W)]; }
__device__ __forceinline__ void vec3b(const SfDev &P, float row, float col, int &r, int &g, int &b) {
    const unsigned long long off = (unsigned long long)(int)row * P.gstride + 3ull * (unsigned long long)(int)col;
    r = off < P.gbytes ? P.gray[off] : 0;
    g = off + 1 < P.gbytes ? P.gray[off + 1] : 0;
    b = off + 2 < P.gbytes ? P.gray[off + 2] : 0;
}
__device__ __forceinline__ void back_project(const SfDev &P, float u, float v, float d, float &x, float &y, float &z) {
    x = (u - P.cx) / P.fx * d;   // src/SurfelFusion.cpp:80-85 (float expression, stored to double there)
    y = (v - P.cy) / P.fy * d;
    z = d;
}
__device__ __forceinline__ float get_weight(float d) { return (float)fmin(1.0 / (double)d / (double)d, 1.0); }

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seed_init(SfDev P) {
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    int imageX = spX * SP + SP / 2, imageY = spY * SP + SP / 2;
    imageX = imageX < (P.W - 1) ? imageX : (P.W - 1);
    imageY = imageY < (P.H - 1) ? imageY : (P.H - 1);
    msl_seed s;
    memset(&s, 0, sizeof(s));
    if (P.member[(size_t)(imageY / 2) * P.mstride + imageX / 2] != -1) { P.seeds[seedI] = s; return; }
    s.use = 1;
    s.x = (float)imageX; s.y = (float)imageY;
    vec3b(P, (float)imageY, (float)imageX, s.r, s.g, s.b);
    s.meanIntensity = gray_at(P, imageY, imageX);
    s.meanDepth = depth_at(P, imageY, imageX);
    if (s.meanDepth < 0.01) {
        int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
        int xe = xb + SP * 2, ye = yb + SP * 2;
        xb = xb > 0 ? xb : 0; yb = yb > 0 ? yb : 0;
        xe = xe < P.W - 1 ? xe : P.W - 1; ye = ye < P.H - 1 ? ye : P.H - 1;
        bool found = false;
        for (int j = yb; j < ye && !found; j++)
            for (int i = xb; i < xe; i++) {
                const float d = depth_at(P, j, i);
                if (d > 0.01) { s.meanDepth = d; found = true; break; }
            }
    }
    P.seeds[seedI] = s;
}

// ---------------------------------------------------------------------------------------------
// k_assign: a(p) = argmin seed of pixel p (:357-415 without the `stable` gate).  it == 0: every seed
// is unstable, so every free pixel is processed: write the index map directly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_assign(SfDev P, int it) {
    const int colI = blockIdx.x * 32 + (threadIdx.x & 31), rowI = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (it > 0 && threadIdx.x <= PROP_ROUNDS) P.changed[threadIdx.x] = threadIdx.x == 0 ? 1 : 0;
        if (threadIdx.x >= 64 && threadIdx.x < 64 + NCHUNK) P.chunkAbort[threadIdx.x - 64] = 0x7FFFFFFF;
    }
    if (colI >= P.W || rowI >= P.H) return;
    const int p = rowI * P.W + colI;
    if (P.member[(size_t)(rowI / 2) * P.mstride + colI / 2] != -1) {
        if (it == 0) P.index[p] = 0; else P.amap[p] = -2;
        return;
    }
    const float myIntensity = gray_at(P, rowI, colI);
    float myInvDepth = 0.0f;
    const float dpx = depth_at(P, rowI, colI);
    if (dpx > 0.01) myInvDepth = (float)(1.0 / (double)dpx);
    const int baseSpX = colI / SP, baseSpY = rowI / SP;
    float minDistDepth = 1e6f, minDistNodepth = 1e6f;
    int minSpIndexDepth = -1, minSpIndexNodepth = -1;
    bool allHasDepth = true;
    for (int checkI = -1; checkI <= 1; checkI++)
        for (int checkJ = -1; checkJ <= 1; checkJ++) {
            const int checkSpX = baseSpX + checkI, checkSpY = baseSpY + checkJ;
            const int distSpX = abs(checkSpX * SP + SP / 2 - colI), distSpY = abs(checkSpY * SP + SP / 2 - rowI);
            if (distSpX < SP && distSpY < SP && checkSpX >= 0 && checkSpX < P.spW && checkSpY >= 0 && checkSpY < P.spH) {
                const int spIndex = checkSpY * P.spW + checkSpX;
                const msl_seed *s = &P.seeds[spIndex];
                const float sx = s->x, sy = s->y, sI = s->meanIntensity, sD = s->meanDepth;
                // calculateCost (:333-355)
                float nodepthCost = 0;
                const float dist = (sx - colI) * (sx - colI) + (sy - rowI) * (sy - rowI);
                nodepthCost += dist / ((SP / 2) * (SP / 2));
                const float intensityDiff = sI - myIntensity;
                nodepthCost = (float)((double)nodepthCost + (double)(intensityDiff * intensityDiff) / 100.0);
                float depthCost = nodepthCost;
                bool has = false;
                if (sD > 0 && myInvDepth > 0) {
                    const float inverseDepthDiff = (float)(1.0 / (double)sD - (double)myInvDepth);
                    depthCost = (float)((double)depthCost + (double)(inverseDepthDiff * inverseDepthDiff) * 400.0);
                    has = true;
                }
                allHasDepth &= has;
                if (depthCost < minDistDepth) { minDistDepth = depthCost; minSpIndexDepth = spIndex; }
                if (nodepthCost < minDistNodepth) { minDistNodepth = nodepthCost; minSpIndexNodepth = spIndex; }
            }
        }
    const int pick = allHasDepth ? minSpIndexDepth : minSpIndexNodepth;
    if (it == 0) P.index[p] = pick >= 0 ? pick : 0;
    else P.amap[p] = pick;
}

// t(s) = raster position from which seed s counts as unstable: 0 if unstable at pass start, else
// 1 + the first processed pixel that picked it (min-fixpoint).
__global__ __launch_bounds__(256) void k_tmin_init(SfDev P) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s < P.nseeds) P.tmin[s] = P.seeds[s].stable ? T_INF : 0u;
}

__global__ __launch_bounds__(256) void k_prop(SfDev P, int round) {
    if (!P.changed[round]) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P.W * P.H) return;
    const int a = P.amap[p];
    if (a < 0) return;
    if (P.tmin[P.index[p]] <= (unsigned)p) {
        if (P.tmin[a] > (unsigned)p + 1u) {
            const unsigned old = atomicMin(&P.tmin[a], (unsigned)p + 1u);
            if (old > (unsigned)p + 1u) P.changed[round + 1] = 1;
        }
    }
}

__global__ __launch_bounds__(256) void k_commit_px(SfDev P) {
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p == 0 && P.changed[PROP_ROUNDS]) P.ctr[5] = 10;  // fixpoint not reached within PROP_ROUNDS
    if (p >= P.W * P.H) return;
    const int a = P.amap[p];
    if (a < 0) return;
    if (P.tmin[P.index[p]] <= (unsigned)p) P.index[p] = a;
}

// ---------------------------------------------------------------------------------------------
// k_update_seeds: one wave per seed (:428-515).  Integer-valued sums are exact in any order; the
// float depth sum and the Huber/Newton sums run in window raster order on lane 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_update_seeds(SfDev P, int it) {
    __shared__ float s_depth[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seedI = blockIdx.x * 4 + wave;
    if (seedI >= P.nseeds) return;
    const msl_seed S = P.seeds[seedI];
    bool stable = S.stable;
    if (it > 0) stable = P.tmin[seedI] == T_INF;  // cleared by any processed pixel that picked this seed
    // seedsTmp[]._pad: 0 = skipped (value = old seed with the post-pixel-pass stable flag), 1 = chunk abort, 2 = processed
    if (!S.use || stable) { if (lane == 0) { msl_seed t = S; t.stable = stable; t._pad = 0; P.seedsTmp[seedI] = t; } return; }
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
    int xe = xb + SP * 2, ye = yb + SP * 2;
    const int xb0 = xb, yb0 = yb;
    xb = xb > 0 ? xb : 0; yb = yb > 0 ? yb : 0;
    xe = xe < P.W - 1 ? xe : P.W - 1; ye = ye < P.H - 1 ? ye : P.H - 1;
    int sumX = 0, sumY = 0, sumI = 0, cnt = 0, ndepth = 0;
    float *dl = s_depth[wave];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;           // raster position inside the unclipped 16x16 window
        const int j = yb0 + (q >> 4), i = xb0 + (q & 15);
        bool own = false, hasd = false;
        float d = 0;
        if (j >= yb && j < ye && i >= xb && i < xe && P.index[j * P.W + i] == seedI) {
            own = true;
            sumX += i; sumY += j; sumI += gray_at(P, j, i); cnt++;
            d = depth_at(P, j, i);
            hasd = d > 0.1;
        }
        const unsigned long long m = __ballot(hasd);
        if (hasd) dl[ndepth + __popcll(m & ((1ull << lane) - 1ull))] = d;
        ndepth += __popcll(m);
        (void)own;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sumX += __shfl_xor(sumX, d, 64); sumY += __shfl_xor(sumY, d, 64);
        sumI += __shfl_xor(sumI, d, 64); cnt += __shfl_xor(cnt, d, 64);
    }
    if (lane != 0) return;
    msl_seed T = S;
    T._pad = 2;
    if (cnt == 0) {  // `return`: ends the chunk (:473-474)
        atomicMin(&P.chunkAbort[seed_chunk(seedI, P.nseeds)], seedI);
        T.stable = 0; T._pad = 1;
        P.seedsTmp[seedI] = T;
        return;
    }
    const float sumIntensityNum = (float)cnt;
    const float sumIntensity = (float)sumI / sumIntensityNum, mX = (float)sumX / sumIntensityNum, mY = (float)sumY / sumIntensityNum;
    const float preIntensity = S.meanIntensity, preX = S.x, preY = S.y;
    T.meanIntensity = sumIntensity; T.x = mX; T.y = mY;
    vec3b(P, mY, mX, T.r, T.g, T.b);
    const float updateDiff = fabsf(preIntensity - sumIntensity) + fabsf(preX - mX) + fabsf(preY - mY);
    T.stable = (updateDiff < 0.2) ? 1 : 0;
    if (ndepth > 0) {
        float sumDepth = 0.0f;
        for (int p = 0; p < ndepth; p++) sumDepth += dl[p];
        float meanDepth = sumDepth / (float)ndepth;
        for (int newtonI = 0; newtonI < 5; newtonI++) {
            float sumA = 0, sumB = 0;
            for (int p = 0; p < ndepth; p++) {
                const float residual = meanDepth - dl[p];
                if (residual < HUBER_RANGE && residual > -HUBER_RANGE) { sumA += 2 * residual; sumB += 2; }
                else sumA = (float)((double)sumA + (residual > 0 ? HUBER_RANGE : -1 * HUBER_RANGE));
            }
            const float deltaDepth = (float)((double)(-sumA) / ((double)sumB + 10.0));
            meanDepth = meanDepth + deltaDepth;
            if (deltaDepth < 0.01 && deltaDepth > -0.01) break;
        }
        T.meanDepth = meanDepth;
    } else {
        T.meanDepth = 0.0f;
    }
    P.seedsTmp[seedI] = T;
}

__global__ __launch_bounds__(256) void k_commit_seeds(SfDev P) {
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    const msl_seed T = P.seedsTmp[seedI];
    msl_seed out;
    if (T._pad == 0) out = T;                                                             // skipped
    else if (T._pad == 2 && seedI < P.chunkAbort[seed_chunk(seedI, P.nseeds)]) out = T;  // processed
    else { out = P.seeds[seedI]; out.stable = 0; }   // chunk already ended: values untouched, unstable after the pixel pass
    out._pad = 0;
    P.seeds[seedI] = out;
}

// ---------------------------------------------------------------------------------------------
// k_seed_plane: calculateNorms (:775-803) fused per seed.  Pixel positions and cross-product
// normals are recomputed from depth instead of materialising spaceMap (7.4 MB f64) / normMap.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void pixel_normal(const SfDev &P, int row, int col, float myX, float myY, float myZ, float &nX, float &nY,
                                             float &nZ) {
    nX = nY = nZ = 0.0f;
    if (row < 1 || row > P.H - 2 || col < 1 || col > P.W - 2) return;  // never written (:620-625)
    float rightX, rightY, rightZ, downX, downY, downZ;
    back_project(P, (float)(col + 1), (float)row, depth_at(P, row, col + 1), rightX, rightY, rightZ);
    back_project(P, (float)col, (float)(row + 1), depth_at(P, row + 1, col), downX, downY, downZ);
    if (myZ < 0.1 || rightZ < 0.1 || downZ < 0.1) return;
    rightX = rightX - myX; rightY = rightY - myY; rightZ = rightZ - myZ;
    downX = downX - myX; downY = downY - myY; downZ = downZ - myZ;
    float normX = rightY * downZ - rightZ * downY;
    float normY = rightZ * downX - rightX * downZ;
    float normZ = rightX * downY - rightY * downX;
    const float normLength = sqrtf(normX * normX + normY * normY + normZ * normZ);
    normX /= normLength; normY /= normLength; normZ /= normLength;
    const float viewAngle = (normX * myX + normY * myY + normZ * myZ) / sqrtf(myX * myX + myY * myY + myZ * myZ);
    if (viewAngle > -MAX_ANGLE_COS && viewAngle < MAX_ANGLE_COS) return;
    nX = normX; nY = normY; nZ = normZ;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// adjugate / determinant inverse of a 4x4 (column-major), same operation order as the oracle
template <typename T>
__host__ __device__ inline void inverse4(const T *m, T *inv) {
#define M_(r, c) m[(c) * 4 + (r)]
#define DET3(r0, r1, r2, c0, c1, c2)                                                                     \
    (M_(r0, c0) * (M_(r1, c1) * M_(r2, c2) - M_(r1, c2) * M_(r2, c1)) -                                 \
     M_(r0, c1) * (M_(r1, c0) * M_(r2, c2) - M_(r1, c2) * M_(r2, c0)) +                                 \
     M_(r0, c2) * (M_(r1, c0) * M_(r2, c1) - M_(r1, c1) * M_(r2, c0)))
    T cof[4][4];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            int rr[3], cc[3], k = 0;
            for (int i = 0; i < 4; i++) if (i != r) rr[k++] = i;
            k = 0;
            for (int i = 0; i < 4; i++) if (i != c) cc[k++] = i;
            const T d = DET3(rr[0], rr[1], rr[2], cc[0], cc[1], cc[2]);
            cof[r][c] = ((r + c) & 1) ? -d : d;
        }
    const T det = ((M_(0, 0) * cof[0][0] + M_(0, 1) * cof[0][1]) + M_(0, 2) * cof[0][2]) + M_(0, 3) * cof[0][3];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) inv[c * 4 + r] = cof[c][r] / det;
#undef DET3
#undef M_
}

__global__ __launch_bounds__(256) void k_seed_plane(SfDev P) {
    __shared__ float s_d[4][256];
    __shared__ float s_n[4][3][256];
    __shared__ float s_p[4][3][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seedI = blockIdx.x * 4 + wave;
    if (seedI >= P.nseeds) return;
    msl_seed S = P.seeds[seedI];
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    const int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
    const int total = P.W * P.H;
    float *dl = s_d[wave];
    int nvalid = 0;
    float maxDist = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int q = lane + 64 * k;
        const int j = yb + (q >> 4), i = xb + (q & 15);
        const int pixelIndex = j * P.W + i;
        bool valid = false;
        float myDepth = 0, nX = 0, nY = 0, nZ = 0, pX = 0, pY = 0, pZ = 0;
        if (pixelIndex >= 0 && pixelIndex < total && P.index[pixelIndex] == seedI) {
            const float xDiff = i - S.x, yDiff = j - S.y;
            const float dist = xDiff * xDiff + yDiff * yDiff;
            if (dist > maxDist) maxDist = dist;
            myDepth = depth_flat(P, pixelIndex);
            if (myDepth > 0.05) {
                valid = true;
                const int row = pixelIndex / P.W, col = pixelIndex % P.W;   // wrapped pixel (App. B.6)
                back_project(P, (float)col, (float)row, myDepth, pX, pY, pZ);
                pixel_normal(P, row, col, pX, pY, pZ, nX, nY, nZ);
            }
        }
        const unsigned long long m = __ballot(valid);
        if (valid) {
            const int o = nvalid + __popcll(m & ((1ull << lane) - 1ull));
            dl[o] = myDepth;
            s_n[wave][0][o] = nX; s_n[wave][1][o] = nY; s_n[wave][2][o] = nZ;
            s_p[wave][0][o] = pX; s_p[wave][1][o] = pY; s_p[wave][2][o] = pZ;
        }
        nvalid += __popcll(m);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) maxDist = fmaxf(maxDist, __shfl_xor(maxDist, d, 64));
    if (nvalid < 16) return;
    float meanDepth = S.meanDepth;
    // inliers, kept in order: compact positions in place (inlier list is a subsequence)
    int ninl = 0;
    float normX = 0.0f, normY = 0.0f, normZ = 0.0f, normB = 0.0f;
    {
        // ordered compaction of inliers into the front of s_p / s_n
        int base = 0;
        for (int k = 0; k < 4; k++) {
            const int o = lane + 64 * k;
            bool inl = false;
            float a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
            if (o < nvalid) {
                const float residual = meanDepth - dl[o];
                inl = residual < HUBER_RANGE && residual > -HUBER_RANGE;
                a0 = s_p[wave][0][o]; a1 = s_p[wave][1][o]; a2 = s_p[wave][2][o];
                b0 = s_n[wave][0][o]; b1 = s_n[wave][1][o]; b2 = s_n[wave][2][o];
            }
            const unsigned long long m = __ballot(inl);
            // all lanes have read slot o (>= base + rank) before anyone writes: reads of this round precede writes
            if (inl) {
                const int w = base + __popcll(m & ((1ull << lane) - 1ull));
                s_p[wave][0][w] = a0; s_p[wave][1][w] = a1; s_p[wave][2][w] = a2;
                s_n[wave][0][w] = b0; s_n[wave][1][w] = b1; s_n[wave][2][w] = b2;
            }
            base += __popcll(m);
        }
        ninl = base;
    }
    if ((float)ninl / (float)nvalid < 0.8) return;
    float sumX = 0.0f, sumY = 0.0f, sumZ = 0.0f;
    if (lane == 0) {
        for (int p = 0; p < ninl; p++) { normX += s_n[wave][0][p]; normY += s_n[wave][1][p]; normZ += s_n[wave][2][p]; }
        const float normLength = sqrtf(normX * normX + normY * normY + normZ * normZ);
        normX = normX / normLength; normY = normY / normLength; normZ = normZ / normLength;
        for (int p = 0; p < ninl; p++) { sumX += s_p[wave][0][p]; sumY += s_p[wave][1][p]; sumZ += s_p[wave][2][p]; }
        sumX /= ninl; sumY /= ninl; sumZ /= ninl;
    }
    normX = __shfl(normX, 0, 64); normY = __shfl(normY, 0, 64); normZ = __shfl(normZ, 0, 64);
    sumX = __shfl(sumX, 0, 64); sumY = __shfl(sumY, 0, 64); sumZ = __shfl(sumZ, 0, 64);
    // getHuberNorm (:91-165): centred points in registers, 4 per lane
    float cx_[4], cy_[4], cz_[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int o = lane + 64 * k;
        if (o < ninl) { cx_[k] = s_p[wave][0][o] - sumX; cy_[k] = s_p[wave][1][o] - sumY; cz_[k] = s_p[wave][2][o] - sumZ; }
        else { cx_[k] = cy_[k] = cz_[k] = 0; }
    }
    float nx = normX, ny = normY, nz = normZ, nb = 0.0f;
    for (int gnI = 0; gnI < 5; gnI++) {
        double J0 = 0, J1 = 0, J2 = 0, J3 = 0, H00 = 0, H01 = 0, H02 = 0, H03 = 0, H11 = 0, H12 = 0, H13 = 0, H22 = 0, H23 = 0, H33 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (lane + 64 * k >= ninl) continue;
            const float px = cx_[k], py = cy_[k], pz = cz_[k];
            const float residual = px * nx + py * ny + pz * nz + nb;
            if (residual < HUBER_RANGE && residual > -1 * HUBER_RANGE) {
                J0 += 2 * residual * px; J1 += 2 * residual * py; J2 += 2 * residual * pz; J3 += 2 * residual;
                H00 += 2 * px * px; H01 += 2 * px * py; H02 += 2 * px * pz; H03 += 2 * px;
                H11 += 2 * py * py; H12 += 2 * py * pz; H13 += 2 * py;
                H22 += 2 * pz * pz; H23 += 2 * pz; H33 += 2;
            } else if (residual >= HUBER_RANGE) {
                J0 += HUBER_RANGE * px; J1 += HUBER_RANGE * py; J2 += HUBER_RANGE * pz; J3 += HUBER_RANGE;
            } else if (residual <= -1 * HUBER_RANGE) {
                J0 += -1 * HUBER_RANGE * px; J1 += -1 * HUBER_RANGE * py; J2 += -1 * HUBER_RANGE * pz; J3 += -1 * HUBER_RANGE;
            }
        }
        J0 = wave_sum_d(J0); J1 = wave_sum_d(J1); J2 = wave_sum_d(J2); J3 = wave_sum_d(J3);
        H00 = wave_sum_d(H00); H01 = wave_sum_d(H01); H02 = wave_sum_d(H02); H03 = wave_sum_d(H03);
        H11 = wave_sum_d(H11); H12 = wave_sum_d(H12); H13 = wave_sum_d(H13);
        H22 = wave_sum_d(H22); H23 = wave_sum_d(H23); H33 = wave_sum_d(H33);
        double hs[16] = {H00 + 5, H01, H02, H03, H01, H11 + 5, H12, H13, H02, H12, H22 + 5, H23, H03, H13, H23, H33 + 5};
        double inv[16];
        inverse4<double>(hs, inv);
        const double jac[4] = {J0, J1, J2, J3};
        double upd[4];
#pragma unroll
        for (int r = 0; r < 4; r++) upd[r] = ((inv[0 * 4 + r] * jac[0] + inv[1 * 4 + r] * jac[1]) + inv[2 * 4 + r] * jac[2]) + inv[3 * 4 + r] * jac[3];
        nx = (float)((double)nx - upd[0]); ny = (float)((double)ny - upd[1]); nz = (float)((double)nz - upd[2]); nb = (float)((double)nb - upd[3]);
    }
    if (lane != 0) return;
    nb = nb - (nx * sumX + ny * sumY + nz * sumZ);
    {
        const float normLength = sqrtf(nx * nx + ny * ny + nz * nz);
        nx /= normLength; ny /= normLength; nz /= normLength; nb /= normLength;
    }
    normX = nx; normY = ny; normZ = nz; normB = nb;
    float ax, ay, az;
    back_project(P, S.x, S.y, meanDepth, ax, ay, az);
    double avgX = ax, avgY = ay, avgZ = az;
    {
        const float k = (float)(-1 * (avgX * (double)normX + avgY * (double)normY + avgZ * (double)normZ) - (double)normB);
        avgX += (double)(k * normX); avgY += (double)(k * normY); avgZ += (double)(k * normZ);
        meanDepth = (float)avgZ;
    }
    float viewCos = (float)(-1.0 * ((double)normX * avgX + (double)normY * avgY + (double)normZ * avgZ) / sqrt(avgX * avgX + avgY * avgY + avgZ * avgZ));
    if (viewCos < 0) { viewCos = -viewCos; normX = -normX; normY = -normY; normZ = -normZ; }
    S.normX = normX; S.normY = normY; S.normZ = normZ;
    S.posX = (float)avgX; S.posY = (float)avgY; S.posZ = (float)avgZ;
    S.meanDepth = meanDepth; S.viewCos = viewCos; S.size = sqrtf(maxDist);
    P.seeds[seedI] = S;
}

// ---------------------------------------------------------------------------------------------
// k_fuse (:167-283): one thread per live surfel, grid-stride over the device-side live count.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void mul4(const float *m, float v0, float v1, float v2, float v3, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2) + m[12 + r] * v3;
}
__device__ __forceinline__ void mul3(const float *m, float v0, float v1, float v2, float out[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2;
}

__global__ __launch_bounds__(256) void k_fuse(SfDev P) {
    __shared__ unsigned s_cnt[2];
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const long long n = P.ctr[0];
    const MapSoA &M = P.map;
    unsigned ndel = 0, nupd = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int updateTimes = M.updateTimes[i];
        const int lastUpdate = M.lastUpdate[i];
        if (P.ref - lastUpdate > 5 && updateTimes < 5) { if (updateTimes != 0) M.updateTimes[i] = 0; ndel++; continue; }
        if (updateTimes == 0) { ndel++; continue; }
        const float Lpx = M.px[i], Lpy = M.py[i], Lpz = M.pz[i];
        float pc[4];
        mul4(P.invPose, Lpx, Lpy, Lpz, 1.0f, pc);
        if (pc[2] < P.fuseNear || pc[2] > P.fuseFar) continue;
        float nc[3];
        mul3(P.invPose, M.nx[i], M.ny[i], M.nz[i], nc);
        const float projectU = pc[0] * P.fx / pc[2] + P.cx, projectV = pc[1] * P.fy / pc[2] + P.cy;  // :75-78
        const int pUInt = (int)((double)projectU + 0.5), pVInt = (int)((double)projectV + 0.5);
        if (pUInt < 1 || pUInt > P.W - 2 || pVInt < 1 || pVInt > P.H - 2) continue;
        if ((double)pc[2] < (double)depth_at(P, pVInt, pUInt) - 1.0) { M.updateTimes[i] = 0; ndel++; continue; }
        const int spIndex = P.index[pVInt * P.W + pUInt];
        const msl_seed S = P.seeds[spIndex];
        if (S.normX == 0 && S.normY == 0 && S.normZ == 0) continue;
        if (S.viewCos < MAX_ANGLE_COS) continue;
        const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
        float tolerateDiff = (float)((double)(pc[2] * pc[2]) / (BASELINE_D * (double)cameraF) * DISPARITY_ERROR);
        tolerateDiff = tolerateDiff < MIN_TOLERATE_DIFF ? (float)MIN_TOLERATE_DIFF : tolerateDiff;
        if (pc[2] < S.meanDepth - tolerateDiff) continue;
        if (pc[2] > S.meanDepth + tolerateDiff) continue;
        const float normDiffCos = nc[0] * S.normX + nc[1] * S.normY + nc[2] * S.normZ;
        if (normDiffCos < MAX_ANGLE_COS) { M.updateTimes[i] = 0; ndel++; continue; }
        const float oldWeight = M.weight[i];
        const float newWeight = get_weight(S.meanDepth);
        const float sumWeight = oldWeight + newWeight;
        float spPW[4];
        mul4(P.pose, S.posX, S.posY, S.posZ, 1.0f, spPW);
        const float fusedPx = (Lpx * oldWeight + newWeight * spPW[0]) / sumWeight;
        const float fusedPy = (Lpy * oldWeight + newWeight * spPW[1]) / sumWeight;
        const float fusedPz = (Lpz * oldWeight + newWeight * spPW[2]) / sumWeight;
        float fusedNx = nc[0] * oldWeight + newWeight * S.normX;
        float fusedNy = nc[1] * oldWeight + newWeight * S.normY;
        float fusedNz = nc[2] * oldWeight + newWeight * S.normZ;
        const double newNormLength = (double)sqrtf(fusedNx * fusedNx + fusedNy * fusedNy + fusedNz * fusedNz);
        fusedNx = (float)((double)fusedNx / newNormLength); fusedNy = (float)((double)fusedNy / newNormLength);
        fusedNz = (float)((double)fusedNz / newNormLength);
        float newNormW[3];
        mul3(P.pose, fusedNx, fusedNy, fusedNz, newNormW);
        M.px[i] = fusedPx; M.py[i] = fusedPy; M.pz[i] = fusedPz;
        M.r[i] = S.r; M.g[i] = S.g; M.b[i] = S.b;
        M.nx[i] = newNormW[0]; M.ny[i] = newNormW[1]; M.nz[i] = newNormW[2];
        M.weight[i] = sumWeight;
        M.color[i] = S.meanIntensity;
        const float newSize = S.size * fabsf(S.meanDepth / (cameraF * S.viewCos));
        if (newSize < M.size[i]) M.size[i] = newSize;
        M.lastUpdate[i] = P.ref;
        M.updateTimes[i] = updateTimes + 1;
        P.seeds[spIndex].fused = 1;
        nupd++;
    }
    if (ndel) atomicAdd(&s_cnt[0], ndel);
    if (nupd) atomicAdd(&s_cnt[1], nupd);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_cnt[0]) atomicAdd((unsigned long long *)&P.ctr[2], (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd((unsigned long long *)&P.ctr[3], (unsigned long long)s_cnt[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// k_new_surfels (:285-331): single workgroup, seeds in index order, ordered emission.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_new_surfels(SfDev P) {
    __shared__ unsigned s_wave[17];
    unsigned base = 0;
    for (int s0 = 0; s0 < P.nseeds; s0 += 1024) {
        const int i = s0 + threadIdx.x;
        bool emit = false;
        msl_seed S;
        if (i < P.nseeds) {
            S = P.seeds[i];
            emit = !(S.meanDepth == 0) && !S.fused && !(S.viewCos < MAX_ANGLE_COS) && !(S.normX == 0 && S.normY == 0 && S.normZ == 0);
        }
        unsigned tot;
        const unsigned pos = base + block_excl_scan(emit ? 1u : 0u, s_wave, &tot);
        if (emit) {
            float pw[4], nw[3];
            mul4(P.pose, S.posX, S.posY, S.posZ, 1.0f, pw);
            mul3(P.pose, S.normX, S.normY, S.normZ, nw);
            msl_surfel e;
            e.px = pw[0]; e.py = pw[1]; e.pz = pw[2];
            e.r = S.r; e.g = S.g; e.b = S.b;
            e.nx = nw[0]; e.ny = nw[1]; e.nz = nw[2];
            const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
            e.size = S.size * fabsf(S.meanDepth / (cameraF * S.viewCos));
            e.color = S.meanIntensity;
            e.weight = get_weight(S.meanDepth);
            e.updateTimes = 1;
            e.lastUpdate = P.ref;
            P.newSurfels[pos] = e;
        }
        base += tot;
    }
    if (threadIdx.x == 0) P.ctr[1] = base;
}

// ---------------------------------------------------------------------------------------------
// Resident-map compaction (SurfelMapping.cpp:366-391) with prefix sums:
//   deleted slots ascending d_0 < ... < d_{D-1}; new surfel k -> d_{D-1-k} while any remain, else
//   appended; leftover holes d_0..d_{D-K-1}: the live elements of the tail [n-(D-K), n) move, in
//   ascending order, into the holes below n-(D-K) in ascending order.
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 4096;  // per workgroup of 1024 threads

__global__ __launch_bounds__(1024) void k_del_count(SfDev P) {
    __shared__ unsigned s_c;
    const long long n = P.ctr[0];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        if (threadIdx.x == 0) s_c = 0;
        __syncthreads();
        unsigned c = 0;
        for (int k = 0; k < 4; k++) {
            const long long i = b * SCAN_ITEMS + k * 1024 + threadIdx.x;
            if (i < n && P.map.updateTimes[i] == 0) c++;
        }
        if (c) atomicAdd(&s_c, c);
        __syncthreads();
        if (threadIdx.x == 0) P.blockSums[b] = s_c;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void k_scan_partials(SfDev P) {
    __shared__ unsigned s_wave[17];
    const long long n = P.ctr[0];
    const int nblk = (int)((n + SCAN_ITEMS - 1) / SCAN_ITEMS);
    unsigned carry = 0;
    for (int b0 = 0; b0 < nblk; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        const unsigned v = b < nblk ? P.blockSums[b] : 0;
        unsigned tot;
        const unsigned ex = carry + block_excl_scan(v, s_wave, &tot);
        if (b < nblk) P.blockSums[b] = ex;
        carry += tot;
    }
    if (threadIdx.x == 0) {
        const long long D = carry, K = P.ctr[1];
        P.ctr[4] = n;                 // n before
        P.ctr[2] = D;                 // deleted (authoritative count)
        const long long nAfter = D >= K ? n - (D - K) : n + (K - D);
        if ((unsigned long long)nAfter > P.cap) P.ctr[5] = 20;  // capacity exceeded
        P.ctr[6] = nAfter;
    }
}

__global__ __launch_bounds__(1024) void k_del_list(SfDev P) {
    __shared__ unsigned s_wave[17];
    const long long n = P.ctr[4];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        unsigned base = P.blockSums[b];
        for (int k = 0; k < 4; k++) {
            const long long i = b * SCAN_ITEMS + k * 1024 + threadIdx.x;
            const unsigned f = (i < n && P.map.updateTimes[i] == 0) ? 1u : 0u;
            unsigned tot;
            const unsigned pos = base + block_excl_scan(f, s_wave, &tot);
            if (f) P.delList[pos] = (unsigned)i;
            base += tot;
        }
    }
}

__device__ __forceinline__ void store_surfel(const MapSoA &M, long long i, const msl_surfel &e) {
    M.px[i] = e.px; M.py[i] = e.py; M.pz[i] = e.pz; M.nx[i] = e.nx; M.ny[i] = e.ny; M.nz[i] = e.nz;
    M.size[i] = e.size; M.color[i] = e.color; M.r[i] = e.r; M.g[i] = e.g; M.b[i] = e.b; M.weight[i] = e.weight;
    M.updateTimes[i] = e.updateTimes; M.lastUpdate[i] = e.lastUpdate;
}
__device__ __forceinline__ void move_surfel(const MapSoA &M, long long dst, long long src) {
    M.px[dst] = M.px[src]; M.py[dst] = M.py[src]; M.pz[dst] = M.pz[src]; M.nx[dst] = M.nx[src]; M.ny[dst] = M.ny[src];
    M.nz[dst] = M.nz[src]; M.size[dst] = M.size[src]; M.color[dst] = M.color[src]; M.r[dst] = M.r[src]; M.g[dst] = M.g[src];
    M.b[dst] = M.b[src]; M.weight[dst] = M.weight[src]; M.updateTimes[dst] = M.updateTimes[src]; M.lastUpdate[dst] = M.lastUpdate[src];
}

// place the new surfels: k < min(K, D) -> slot d_{D-1-k}; the rest appended after n
__global__ __launch_bounds__(256) void k_place_new(SfDev P) {
    if (P.ctr[5] == 20) return;
    const long long n = P.ctr[4], D = P.ctr[2], K = P.ctr[1];
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= K) return;
    const long long dst = k < D ? (long long)P.delList[D - 1 - k] : n + (k - D);
    store_surfel(P.map, dst, P.newSurfels[k]);
}

// Tail compaction when D > K (the `while (deletedIndex.size() > 0)` loop, SurfelMapping.cpp:386-390).
// Step i (i = 1..R) of the literal loop moves the element at position n-i into the i-th largest leftover
// hole.  A hole inside the tail [nFinal, n) only relays: what lands there is moved again later.  So the
// a-th smallest leftover hole (all < nFinal) finally receives resolve(nFinal + a), where
// resolve(p) = p if p is live, else resolve(n - rank_desc(p)) -- a short upward chain.
constexpr int TAIL_MAX_HOPS = 64;

__global__ __launch_bounds__(256) void k_tail_resolve(SfDev P) {
    const long long n = P.ctr[4], D = P.ctr[2], K = P.ctr[1];
    if (P.ctr[5] == 20 || D <= K) return;
    const long long R = D - K, nFinal = n - R;
    auto lower = [&](long long x) -> long long {   // first index in delList[0..R) with value >= x
        long long lo = 0, hi = R;
        while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)P.delList[mid] < x) lo = mid + 1; else hi = mid; }
        return lo;
    };
    const long long cntLow = lower(nFinal);
    for (long long a = (long long)blockIdx.x * 256 + threadIdx.x; a < cntLow; a += (long long)gridDim.x * 256) {
        long long p = nFinal + a;
        int hop = 0;
        for (; hop < TAIL_MAX_HOPS; hop++) {
            const long long lb = lower(p);
            if (lb < R && (long long)P.delList[lb] == p) p = n - (R - lb);   // relay hole: follow to where its content came from
            else break;
        }
        if (hop == TAIL_MAX_HOPS) P.ctr[7] = -1;   // pathological chain: fall back to the literal loop
        P.srcOf[a] = (unsigned)p;
    }
}

__global__ __launch_bounds__(256) void k_tail_move(SfDev P) {
    const long long n = P.ctr[4], D = P.ctr[2], K = P.ctr[1];
    if (P.ctr[5] == 20) return;
    if (D > K) {
        const long long R = D - K, nFinal = n - R;
        if (P.ctr[7] == -1) {
            // literal back-to-front loop, one thread (only for pathological delete patterns)
            if (blockIdx.x == 0 && threadIdx.x == 0)
                for (long long i = 1; i <= R; i++) {
                    const long long hole = P.delList[R - i], src = n - i;
                    if (src != hole) move_surfel(P.map, hole, src);
                }
        } else {
            long long lo = 0, hi = R;
            while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)P.delList[mid] < nFinal) lo = mid + 1; else hi = mid; }
            const long long cntLow = lo;
            for (long long a = (long long)blockIdx.x * 256 + threadIdx.x; a < cntLow; a += (long long)gridDim.x * 256)
                move_surfel(P.map, (long long)P.delList[a], (long long)P.srcOf[a]);
        }
    }
}

__global__ void k_end_frame(long long *ctr) {
    if (threadIdx.x == 0 && ctr[5] != 20) { ctr[0] = ctr[6]; if (ctr[7] == -1) ctr[7] = 0; }
}

// AoS <-> SoA conversion for upload / download / host-vector mode
__global__ __launch_bounds__(256) void k_aos_to_soa(MapSoA M, const msl_surfel *src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_surfel(M, i, src[i]);
}
__global__ __launch_bounds__(256) void k_soa_to_aos(MapSoA M, msl_surfel *dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    msl_surfel e;
    e.px = M.px[i]; e.py = M.py[i]; e.pz = M.pz[i]; e.nx = M.nx[i]; e.ny = M.ny[i]; e.nz = M.nz[i];
    e.size = M.size[i]; e.color = M.color[i]; e.r = M.r[i]; e.g = M.g[i]; e.b = M.b[i]; e.weight = M.weight[i];
    e.updateTimes = M.updateTimes[i]; e.lastUpdate = M.lastUpdate[i];
    dst[i] = e;
}
__global__ void k_set_ctr(long long *ctr, long long n) {
    if (threadIdx.x == 0) { ctr[0] = n; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; ctr[4] = n; ctr[6] = n; }
}
__global__ void k_begin_frame(long long *ctr) {
    if (threadIdx.x == 0) { ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; ctr[4] = ctr[0]; }
}

enum { SK_SEED_INIT = 0, SK_ASSIGN, SK_PROP, SK_COMMIT_PX, SK_UPDATE_SEEDS, SK_COMMIT_SEEDS, SK_SEED_PLANE, SK_FUSE, SK_NEW, SK_COMPACT,
       SK_CONVERT, SK_COPY };
const char *kSfNames[MSL_SF_NKERNELS] = {"k_seed_init", "k_assign", "k_prop", "k_commit_px", "k_update_seeds", "k_commit_seeds",
                                         "k_seed_plane", "k_fuse", "k_new_surfels", "k_compact", "k_convert", "copy"};

}  // namespace

struct msl_sf {
    int device = 0;
    SfDev dev{};
    hipStream_t stream = nullptr; bool ownStream = true;
    // owned device buffers
    uint8_t *d_gray = nullptr; float *d_depth = nullptr; int32_t *d_member = nullptr;
    size_t grayCap = 0, depthCap = 0, memberCap = 0;
    msl_seed *d_seeds = nullptr, *d_seedsTmp = nullptr;
    int *d_index = nullptr, *d_amap = nullptr; unsigned *d_tmin = nullptr; int *d_chunkAbort = nullptr, *d_changed = nullptr;
    long long *d_ctr = nullptr; long long *h_ctr = nullptr;  // pinned mirror
    msl_surfel *d_new = nullptr;
    float *d_mapStore = nullptr; size_t mapCap = 0;
    unsigned *d_blockSums = nullptr, *d_delList = nullptr, *d_srcOf = nullptr;
    msl_surfel *d_aos = nullptr; size_t aosCap = 0;
    KernelProfiler prof;
};

namespace {

void set_map_ptrs(msl_sf *h) {
    float *b = h->d_mapStore; const size_t c = h->mapCap;
    MapSoA &M = h->dev.map;
    M.px = b; M.py = b + c; M.pz = b + 2 * c; M.nx = b + 3 * c; M.ny = b + 4 * c; M.nz = b + 5 * c; M.size = b + 6 * c; M.color = b + 7 * c;
    M.r = (int *)(b + 8 * c); M.g = (int *)(b + 9 * c); M.b = (int *)(b + 10 * c); M.weight = b + 11 * c;
    M.updateTimes = (int *)(b + 12 * c); M.lastUpdate = (int *)(b + 13 * c);
    h->dev.cap = c;
    h->dev.blockSums = h->d_blockSums; h->dev.delList = h->d_delList; h->dev.srcOf = h->d_srcOf;
}

// (Re)allocate the resident map for `cap` surfels, preserving the first `keep` entries.
int map_realloc(msl_sf *h, size_t cap, size_t keep) {
    cap = (cap + 1023) & ~(size_t)1023;
    float *nstore = nullptr; unsigned *nbs = nullptr, *ndl = nullptr, *nso = nullptr;
    MSL_HIP_TRY(hipMalloc(&nstore, sizeof(float) * 14 * cap));
    MSL_HIP_TRY(hipMalloc(&nbs, sizeof(unsigned) * (cap / SCAN_ITEMS + 2)));
    MSL_HIP_TRY(hipMalloc(&ndl, sizeof(unsigned) * cap));
    MSL_HIP_TRY(hipMalloc(&nso, sizeof(unsigned) * cap));
    if (keep && h->d_mapStore) {
        MSL_HIP_TRY(hipStreamSynchronize(h->stream));
        for (int a = 0; a < 14; a++)
            MSL_HIP_TRY(hipMemcpy(nstore + (size_t)a * cap, h->d_mapStore + (size_t)a * h->mapCap, sizeof(float) * keep, hipMemcpyDeviceToDevice));
    }
    if (h->d_mapStore) { (void)hipFree(h->d_mapStore); (void)hipFree(h->d_blockSums); (void)hipFree(h->d_delList); (void)hipFree(h->d_srcOf); }
    h->d_mapStore = nstore; h->d_blockSums = nbs; h->d_delList = ndl; h->d_srcOf = nso; h->mapCap = cap;
    set_map_ptrs(h);
    return MSL_OK;
}

int read_ctr(msl_sf *h) {
    MSL_HIP_TRY(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(long long) * 8, hipMemcpyDeviceToHost, h->stream));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    return MSL_OK;
}

int check_err(msl_sf *h) {
    if (h->h_ctr[5]) {
        const long long e = h->h_ctr[5];
        (void)hipMemsetAsync(h->d_ctr + 5, 0, sizeof(long long), h->stream);
        if (e == 20) set_error("resident surfel map capacity exceeded (reserve more with msl_sf_map_reserve)");
        else set_error("surfel pipeline device-side bound exceeded (code %lld)", e);
        return MSL_ERR_OVERFLOW;
    }
    return MSL_OK;
}

// Stage the three images (host -> owned device buffers) or adopt device pointers.
int set_images(msl_sf *h, const uint8_t *gray, size_t gs, const float *depth, size_t ds, const int32_t *member, size_t ms, msl_mem mem) {
    SfDev &D = h->dev;
    const int W = D.W, H = D.H;
    if (!gray || !depth || !member || gs < (size_t)W || ds < (size_t)W * 4 || ms < (size_t)(W / 2) * 4 || (ds & 3) || (ms & 3)) {
        set_error("msl_sf: bad image pointers or strides");
        return MSL_ERR_INVALID;
    }
    D.gstride = gs; D.gbytes = gs * (size_t)H; D.dstride = ds / 4; D.mstride = ms / 4;
    if (mem == MSL_MEM_DEVICE) { D.gray = gray; D.depth = depth; D.member = member; return MSL_OK; }
    const size_t gb = gs * H, db = ds * H, mb = ms * (H / 2);
    if (gb > h->grayCap) { if (h->d_gray) (void)hipFree(h->d_gray); MSL_HIP_TRY(hipMalloc(&h->d_gray, gb)); h->grayCap = gb; }
    if (db > h->depthCap) { if (h->d_depth) (void)hipFree(h->d_depth); MSL_HIP_TRY(hipMalloc(&h->d_depth, db)); h->depthCap = db; }
    if (mb > h->memberCap) { if (h->d_member) (void)hipFree(h->d_member); MSL_HIP_TRY(hipMalloc(&h->d_member, mb)); h->memberCap = mb; }
    h->prof.begin(SK_COPY, h->stream);
    MSL_HIP_TRY(hipMemcpyAsync(h->d_gray, gray, gb, hipMemcpyHostToDevice, h->stream));
    MSL_HIP_TRY(hipMemcpyAsync(h->d_depth, depth, db, hipMemcpyHostToDevice, h->stream));
    MSL_HIP_TRY(hipMemcpyAsync(h->d_member, member, mb, hipMemcpyHostToDevice, h->stream));
    h->prof.end(h->stream);
    D.gray = h->d_gray; D.depth = h->d_depth; D.member = h->d_member;
    return MSL_OK;
}

#define LAUNCH(kid, kern, grid, block, ...)                                     \
    do {                                                                        \
        h->prof.begin(kid, s);                                                  \
        hipLaunchKernelGGL(kern, grid, block, 0, s, __VA_ARGS__);               \
        h->prof.end(s);                                                         \
    } while (0)

// generateSuperPixels + fuse + initializeSurfels on the resident SoA map (ctr[0] live surfels)
int launch_fusion(msl_sf *h, int ref, const float pose[16], bool compact) {
    SfDev &D = h->dev;
    hipStream_t s = h->stream;
    memcpy(D.pose, pose, sizeof(float) * 16);
    inverse4<float>(D.pose, D.invPose);   // pose.inverse() (:59), adjugate/determinant in float
    D.ref = ref;
    const SfDev P = D;
    const int npx = P.W * P.H;
    const dim3 pxGrid((P.W + 31) / 32, (P.H + 7) / 8);
    hipLaunchKernelGGL(k_begin_frame, dim3(1), dim3(64), 0, s, P.ctr);
    LAUNCH(SK_SEED_INIT, k_seed_init, dim3((P.nseeds + 255) / 256), dim3(256), P);
    for (int it = 0; it < 3; it++) {
        LAUNCH(SK_ASSIGN, k_assign, pxGrid, dim3(256), P, it);
        if (it > 0) {
            LAUNCH(SK_PROP, k_tmin_init, dim3((P.nseeds + 255) / 256), dim3(256), P);
            for (int r = 0; r < PROP_ROUNDS; r++) LAUNCH(SK_PROP, k_prop, dim3((npx + 255) / 256), dim3(256), P, r);
            LAUNCH(SK_COMMIT_PX, k_commit_px, dim3((npx + 255) / 256), dim3(256), P);
        }
        LAUNCH(SK_UPDATE_SEEDS, k_update_seeds, dim3((P.nseeds + 3) / 4), dim3(256), P, it);
        LAUNCH(SK_COMMIT_SEEDS, k_commit_seeds, dim3((P.nseeds + 255) / 256), dim3(256), P);
    }
    LAUNCH(SK_SEED_PLANE, k_seed_plane, dim3((P.nseeds + 3) / 4), dim3(256), P);
    LAUNCH(SK_FUSE, k_fuse, dim3(MAP_GRID), dim3(256), P);
    LAUNCH(SK_NEW, k_new_surfels, dim3(1), dim3(1024), P);
    if (compact) {
        h->prof.begin(SK_COMPACT, s);
        hipLaunchKernelGGL(k_del_count, dim3(512), dim3(1024), 0, s, P);
        hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, P);
        hipLaunchKernelGGL(k_del_list, dim3(512), dim3(1024), 0, s, P);
        hipLaunchKernelGGL(k_place_new, dim3((P.nseeds + 255) / 256), dim3(256), 0, s, P);
        hipLaunchKernelGGL(k_tail_resolve, dim3(256), dim3(256), 0, s, P);
        hipLaunchKernelGGL(k_tail_move, dim3(256), dim3(256), 0, s, P);
        hipLaunchKernelGGL(k_end_frame, dim3(1), dim3(64), 0, s, P.ctr);
        h->prof.end(s);
    }
    MSL_HIP_TRY(hipGetLastError());
    return MSL_OK;
}

}  // namespace

extern "C" {

msl_sf *msl_sf_create(int width, int height, float fx, float fy, float cx, float cy, float fuseFar, float fuseNear, int device) {
    if (width < 16 || height < 16 || (width % SP) || (height % SP) || fx == 0 || fy == 0) {
        set_error("msl_sf_create: width/height must be multiples of 8 (>= 16) and fx, fy non-zero");
        return nullptr;
    }
    if (bind_device(device) != MSL_OK) return nullptr;
    msl_sf *h = new msl_sf;
    h->device = device;
    SfDev &D = h->dev;
    D.W = width; D.H = height; D.spW = width / SP; D.spH = height / SP; D.nseeds = D.spW * D.spH;
    D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy; D.fuseFar = fuseFar; D.fuseNear = fuseNear;
    const size_t npx = (size_t)width * height;
    bool ok = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipMalloc(&h->d_seeds, sizeof(msl_seed) * D.nseeds) == hipSuccess;
    ok = ok && hipMalloc(&h->d_seedsTmp, sizeof(msl_seed) * D.nseeds) == hipSuccess;
    ok = ok && hipMalloc(&h->d_index, sizeof(int) * npx) == hipSuccess;
    ok = ok && hipMalloc(&h->d_amap, sizeof(int) * npx) == hipSuccess;
    ok = ok && hipMalloc(&h->d_tmin, sizeof(unsigned) * D.nseeds) == hipSuccess;
    ok = ok && hipMalloc(&h->d_chunkAbort, sizeof(int) * 16) == hipSuccess;
    ok = ok && hipMalloc(&h->d_changed, sizeof(int) * (PROP_ROUNDS + 2)) == hipSuccess;
    ok = ok && hipMalloc(&h->d_ctr, sizeof(long long) * 8) == hipSuccess;
    ok = ok && hipMemset(h->d_ctr, 0, sizeof(long long) * 8) == hipSuccess;
    ok = ok && hipMemset(h->d_seeds, 0, sizeof(msl_seed) * D.nseeds) == hipSuccess;
    ok = ok && hipMemset(h->d_index, 0, sizeof(int) * npx) == hipSuccess;
    ok = ok && hipHostMalloc(&h->h_ctr, sizeof(long long) * 8) == hipSuccess;
    ok = ok && hipMalloc(&h->d_new, sizeof(msl_surfel) * D.nseeds) == hipSuccess;
    if (!ok) { set_error("msl_sf_create: HIP allocation failed"); msl_sf_destroy(h); return nullptr; }
    memset(h->h_ctr, 0, sizeof(long long) * 8);
    D.seeds = h->d_seeds; D.seedsTmp = h->d_seedsTmp; D.index = h->d_index; D.amap = h->d_amap; D.tmin = h->d_tmin;
    D.chunkAbort = h->d_chunkAbort; D.changed = h->d_changed; D.ctr = h->d_ctr; D.newSurfels = h->d_new;
    h->prof.nk = MSL_SF_NKERNELS;
    if (map_realloc(h, 1 << 16, 0) != MSL_OK) { msl_sf_destroy(h); return nullptr; }
    return h;
}

void msl_sf_destroy(msl_sf *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->prof.destroy();
    auto F = [](auto *p) { if (p) (void)hipFree(p); };
    F(h->d_gray); F(h->d_depth); F(h->d_member); F(h->d_seeds); F(h->d_seedsTmp); F(h->d_index); F(h->d_amap); F(h->d_tmin);
    F(h->d_chunkAbort); F(h->d_changed); F(h->d_ctr); F(h->d_new); F(h->d_mapStore); F(h->d_blockSums); F(h->d_delList); F(h->d_srcOf); F(h->d_aos);
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
    if (h->stream && h->ownStream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int msl_sf_set_stream(msl_sf *h, void *hip_stream) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->ownStream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream; h->ownStream = false;
    return MSL_OK;
}

int msl_sf_sync(msl_sf *h) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    return check_err(h);
}

int msl_sf_map_reserve(msl_sf *h, size_t capacity) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    if (capacity <= h->mapCap) return MSL_OK;
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    return map_realloc(h, capacity, (size_t)h->h_ctr[0]);
}

static int ensure_aos(msl_sf *h, size_t n) {
    if (n > h->aosCap) {
        if (h->d_aos) (void)hipFree(h->d_aos);
        h->d_aos = nullptr; h->aosCap = 0;
        MSL_HIP_TRY(hipMalloc(&h->d_aos, sizeof(msl_surfel) * n));
        h->aosCap = n;
    }
    return MSL_OK;
}

int msl_sf_map_upload(msl_sf *h, const msl_surfel *host, size_t n) {
    if (!h || (n && !host)) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    if (n + (size_t)h->dev.nseeds > h->mapCap) {
        MSL_HIP_TRY(hipStreamSynchronize(h->stream));
        int rc = map_realloc(h, n + n / 4 + 4 * (size_t)h->dev.nseeds, 0);
        if (rc != MSL_OK) return rc;
    }
    hipStream_t s = h->stream;
    if (n) {
        int rc = ensure_aos(h, n);
        if (rc != MSL_OK) return rc;
        MSL_HIP_TRY(hipMemcpyAsync(h->d_aos, host, sizeof(msl_surfel) * n, hipMemcpyHostToDevice, s));
        LAUNCH(SK_CONVERT, k_aos_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n);
    }
    hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(64), 0, s, h->d_ctr, (long long)n);
    MSL_HIP_TRY(hipStreamSynchronize(s));
    return MSL_OK;
}

int msl_sf_map_size(msl_sf *h, size_t *n_out) {
    if (!h || !n_out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    *n_out = (size_t)h->h_ctr[0];
    return check_err(h);
}

int msl_sf_map_download(msl_sf *h, msl_surfel *host, size_t cap, size_t *n_out) {
    if (!h || !n_out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    *n_out = n;
    if (n > cap || (n && !host)) { set_error("msl_sf_map_download: capacity %zu < map size %zu", cap, n); return MSL_ERR_CAPACITY; }
    if (n) {
        rc = ensure_aos(h, n);
        if (rc != MSL_OK) return rc;
        hipStream_t s = h->stream;
        LAUNCH(SK_CONVERT, k_soa_to_aos, dim3((unsigned)((n + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n);
        MSL_HIP_TRY(hipMemcpyAsync(host, h->d_aos, sizeof(msl_surfel) * n, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipStreamSynchronize(s));
    }
    return check_err(h);
}

int msl_sf_fuse_resident(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth,
                         size_t depth_stride, const int32_t *member, size_t member_stride, msl_mem img_mem,
                         const float pose_colmajor[16]) {
    if (!h || !pose_colmajor) { set_error("msl_sf_fuse_resident: invalid argument"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = set_images(h, gray, gray_stride, depth, depth_stride, member, member_stride, img_mem);
    if (rc != MSL_OK) return rc;
    return launch_fusion(h, referenceFrameIndex, pose_colmajor, true);
}

int msl_sf_last_counters(msl_sf *h, int64_t counters[5]) {
    if (!h || !counters) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    counters[0] = h->h_ctr[4]; counters[1] = h->h_ctr[1]; counters[2] = h->h_ctr[2]; counters[3] = h->h_ctr[3]; counters[4] = h->h_ctr[0];
    return check_err(h);
}

int msl_sf_fuse(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth, size_t depth_stride,
                const int32_t *member, size_t member_stride, const float pose_colmajor[16], msl_surfel *local, size_t n_local,
                msl_surfel *new_out, size_t new_cap, size_t *n_new) {
    if (!h || !pose_colmajor || !n_new || (n_local && !local)) { set_error("msl_sf_fuse: invalid argument"); return MSL_ERR_INVALID; }
    if (new_cap < (size_t)h->dev.nseeds || !new_out) { set_error("msl_sf_fuse: new_cap must be >= (w/8)*(h/8) = %d", h->dev.nseeds); return MSL_ERR_CAPACITY; }
    int rc = msl_sf_map_upload(h, local, n_local);   // the caller's vector is the map for this call
    if (rc != MSL_OK) return rc;
    rc = set_images(h, gray, gray_stride, depth, depth_stride, member, member_stride, MSL_MEM_HOST);
    if (rc != MSL_OK) return rc;
    rc = launch_fusion(h, referenceFrameIndex, pose_colmajor, false);
    if (rc != MSL_OK) return rc;
    rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t K = (size_t)h->h_ctr[1];
    *n_new = K;
    hipStream_t s = h->stream;
    if (n_local) {
        LAUNCH(SK_CONVERT, k_soa_to_aos, dim3((unsigned)((n_local + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n_local);
        MSL_HIP_TRY(hipMemcpyAsync(local, h->d_aos, sizeof(msl_surfel) * n_local, hipMemcpyDeviceToHost, s));
    }
    if (K) MSL_HIP_TRY(hipMemcpyAsync(new_out, h->d_new, sizeof(msl_surfel) * K, hipMemcpyDeviceToHost, s));
    MSL_HIP_TRY(hipStreamSynchronize(s));
    return MSL_OK;
}

int msl_sf_debug_seeds(msl_sf *h, msl_seed *out) {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    MSL_HIP_TRY(hipMemcpy(out, h->d_seeds, sizeof(msl_seed) * h->dev.nseeds, hipMemcpyDeviceToHost));
    return MSL_OK;
}
int msl_sf_debug_index(msl_sf *h, int32_t *out) {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    MSL_HIP_TRY(hipMemcpy(out, h->d_index, sizeof(int) * (size_t)h->dev.W * h->dev.H, hipMemcpyDeviceToHost));
    return MSL_OK;
}

int msl_sf_profile_enable(msl_sf *h, int on) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    h->prof.set_mode(on);
    return MSL_OK;
}
int msl_sf_profile_read(msl_sf *h, float *ms, int32_t *launches) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    for (int i = 0; i < MSL_SF_NKERNELS; i++) { if (ms) ms[i] = h->prof.ms[i]; if (launches) launches[i] = h->prof.launches[i]; }
    return MSL_OK;
}
const char *msl_sf_kernel_name(int k) { return (k >= 0 && k < MSL_SF_NKERNELS) ? kSfNames[k] : ""; }

}  // extern "C"
