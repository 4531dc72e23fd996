// msl_peac.hip -- the PEAC plane extractor, producer of SurfelFusion's plane-membership image.
//
// Replaces PlaneDetection::readDepthImage + runPlaneDetection (reference src/PlaneExtractor.cpp:44-81), i.e.
// ahc::PlaneFitter<ImagePointCloud>::run (include/peac/AHCPlaneFitter.hpp:218-262) with the reference's default parameters.
//
//   GPU (frame-batched, FP64; the same operations in the same order as the host arithmetic of the reference, ties among exactly equal merge costs
//   resolved by node creation order where the reference depends on heap addresses -- DESIGN.md section 3):
//     k_peac_cloud  organised half-resolution cloud (src/PlaneExtractor.cpp:60-74), only when the caller asks for it
//     k_peac_fit    ONE WAVE PER WINDOW: the lanes evaluate the window's points in parallel -- missing data, depth discontinuity
//                   towards the right / lower neighbour (include/peac/AHCPlaneSeg.hpp:237-285, :41-43), the nine products of
//                   Stats::push (:81-92) -- and stage the products in LDS; nine lanes then add one statistic each in window
//                   raster order (the reference's summation order, so the FP64 sums are the reference's bit for bit); one lane
//                   runs the PCA plane fit (Stats::compute, :148-183) with the 3x3 symmetric eigen-solve of
//                   include/peac/eig33sym.hpp:71-75 (Eigen::SelfAdjointEigenSolver, restated below).
//     k_peac_cluster ONE WAVE PER FRAME: the agglomerative clustering (AHCPlaneFitter.hpp:939-1143).  A sequential chain of pops, but every pop fits a
//                   plane for each neighbour of the popped node (one per lane); binary heap and disjoint set in LDS, neighbour sets as a bit matrix.
//   Host: graph initialisation (AHCPlaneFitter.hpp:756-928; needs cos()), then -- after the device clustering -- block erosion + seeds (:490-596), the
//   FIFO region growing (:422-471), final merge and relabelling (:296-372): order-dependent pixel work, one frame per worker thread at a time, index based
//   (node pool + sorted adjacency vectors instead of shared_ptr / std::set<PlaneSeg*>).  The host also keeps the whole clustering (cluster()): it is the
//   path of small calls -- a single frame, the reference's call pattern, in ~2 ms: the candidate merges of a pop are fitted 16 at a time in SIMD lanes
//   (plane_mse_lanes) -- of frames whose node data does not fit the LDS, of MSL_PEAC_CLUSTER=host, and of msl_peac_membership_from_blocks (no device).
//
// The membership image keeps every quirk a consumer can observe: rid2plid[] default-inserts plane 0 for an
// unknown set id, pixels whose plane was eroded keep their old id, rejected pixels keep their visit counters -2..-6.
#include "msl_common.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <map>
#include <mutex>
#include <queue>
#include <thread>
#include <exception>
#include <stdexcept>
#include <functional>
#include <condition_variable>
#include <atomic>
#include <memory>
#include <sched.h>
#include <cstring>
#include <cstdio>
#include <vector>

namespace {
using namespace msl;

struct PeacDev {
    const uint16_t *depth;
    size_t strideBytes, frameStrideBytes;
    int width, height, cw, ch;             // full-resolution image, half-resolution cloud
    float fx, fy, cx, cy, factor;
    int winW, winH, Nw, Nh, loose;
    double alpha, tol;
    double *cloud;                         // [frames][ch * cw][3] or nullptr
    msl_peac_block *blocks;                // [frames][Nh * Nw]
};

// ---- arithmetic shared by the device kernel and the host clustering (same expressions, IEEE double, no FMA contraction) ----
__host__ __device__ inline double hypot_pos(double x, double y) {   // Eigen::numext::hypot
    const double ax = fabs(x), ay = fabs(y);
    double p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0) return 0;
    return p * sqrt(1.0 + qp * qp);
}

// Eigen::SelfAdjointEigenSolver<Matrix3d>::compute as LA::eig33sym uses it: s[0] <= s[1] <= s[2], V[:][i] the eigenvector of s[i].
// (lower triangle scaled by its largest coefficient, closed-form 3x3 Householder tridiagonalisation, implicit symmetric QR with
// Wilkinson shift and the 2-epsilon deflation test, eigenvalues sorted increasingly with their vectors)
// VECTORS = false leaves out the accumulation of the rotations (q never feeds back into the diagonal / sub-diagonal updates, so the eigenvalues are the
// same bits either way): the clustering only needs the smallest eigenvalue of every candidate merge and the vectors of the one it accepts.
// The diagonal, the sub-diagonal and the rotation matrix are named scalars and every "array" access with a run-time index is a select: the device
// kernels keep them in registers (round 5: the indexed local arrays of rounds 1-4 lived in 80 bytes of scratch memory per lane).  Same operations in
// the same order as before, so the same bits (tests/test_peac_host.py compares against the oracle and the SIMD-lane form).
template <bool VECTORS>
__host__ __device__ inline void eig33sym_t(const double K[3][3], double s[3], double V[3][3]) {
    double a00 = K[0][0], a10 = K[1][0], a11 = K[1][1], a20 = K[2][0], a21 = K[2][1], a22 = K[2][2];
    double scale = fmax(fmax(fmax(fabs(a00), fabs(a10)), fmax(fabs(a11), fabs(a20))), fmax(fabs(a21), fabs(a22)));
    if (scale == 0) scale = 1;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    double d0 = a00, d1, d2, e0, e1;
    double q00 = 1, q01 = 0, q02 = 0, q10 = 0, q11 = 1, q12 = 0, q20 = 0, q21 = 0, q22 = 1;
    const double tiny = 2.2250738585072014e-308;   // std::numeric_limits<double>::min()
    const double v1norm2 = a20 * a20;
    if (v1norm2 <= tiny) {
        d1 = a11; d2 = a22; e0 = a10; e1 = a21;
    } else {
        const double beta = sqrt(a10 * a10 + v1norm2);
        const double invBeta = 1.0 / beta;
        const double m01 = a10 * invBeta, m02 = a20 * invBeta;
        const double qq = 2.0 * m01 * a21 + m02 * (a22 - a11);
        d1 = a11 + m02 * qq; d2 = a22 - m02 * qq;
        e0 = beta; e1 = a21 - m01 * qq;
        if (VECTORS) { q11 = m01; q12 = m02; q21 = m02; q22 = -m01; }
    }
    // dg[i] = (d0, d1, d2)[i], sb[i] = (e0, e1)[i]
    auto DG = [&](int i) -> double { return i == 0 ? d0 : (i == 1 ? d1 : d2); };
    auto SB = [&](int i) -> double { return i == 0 ? e0 : e1; };
    auto setDG = [&](int i, double v) { if (i == 0) d0 = v; else if (i == 1) d1 = v; else d2 = v; };
    auto setSB = [&](int i, double v) { if (i == 0) e0 = v; else e1 = v; };
    int end = 2, start = 0, iter = 0;
    const double precision = 2.0 * 2.220446049250313e-16;
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (fabs(SB(i)) <= (fabs(DG(i)) + fabs(DG(i + 1))) * precision || fabs(SB(i)) <= tiny) setSB(i, 0);
        while (end > 0 && SB(end - 1) == 0.0) end--;
        if (end <= 0) break;
        if (++iter > 30 * 3) break;
        start = end - 1;
        while (start > 0 && SB(start - 1) != 0) start--;
        const double td = (DG(end - 1) - DG(end)) * 0.5, e = SB(end - 1);
        double mu = DG(end);
        if (td == 0.0) mu -= fabs(e);
        else if (e != 0.0) {
            const double e2 = e * e, h = hypot_pos(td, e);
            if (e2 == 0.0) mu -= e / ((td + (td > 0.0 ? h : -h)) / e);
            else mu -= e2 / (td + (td > 0.0 ? h : -h));
        }
        double x = DG(start) - mu, z = SB(start);
        for (int k = start; k < end && z != 0.0; ++k) {
            double c, sn;   // Givens rotation that annihilates z against x
            if (x == 0.0) { c = 0.0; sn = z < 0.0 ? 1.0 : -1.0; }
            else if (fabs(x) > fabs(z)) { const double t = z / x; double u = sqrt(1.0 + t * t); if (x < 0.0) u = -u; c = 1.0 / u; sn = -t * c; }
            else { const double t = x / z; double u = sqrt(1.0 + t * t); if (z < 0.0) u = -u; sn = -1.0 / u; c = -t * sn; }
            const double dk = DG(k), dk1 = DG(k + 1), sk = SB(k);
            const double sdk = sn * dk + c * sk;
            const double dkp1 = sn * sk + c * dk1;
            setDG(k, c * (c * dk - sn * sk) - sn * (c * sk - sn * dk1));
            setDG(k + 1, sn * sdk + c * dkp1);
            const double skNew = c * sdk - sn * dkp1;
            setSB(k, skNew);
            if (k > start) setSB(k - 1, c * SB(k - 1) - sn * z);
            x = skNew;
            if (k < end - 1) { const double s1 = SB(k + 1); z = -sn * s1; setSB(k + 1, c * s1); }
            if (VECTORS) {   // columns k, k + 1 of q (k is 0 or 1)
                if (k == 0) {
                    const double x0 = q00, y0 = q01, x1 = q10, y1 = q11, x2 = q20, y2 = q21;
                    q00 = c * x0 - sn * y0; q01 = sn * x0 + c * y0;
                    q10 = c * x1 - sn * y1; q11 = sn * x1 + c * y1;
                    q20 = c * x2 - sn * y2; q21 = sn * x2 + c * y2;
                } else {
                    const double x0 = q01, y0 = q02, x1 = q11, y1 = q12, x2 = q21, y2 = q22;
                    q01 = c * x0 - sn * y0; q02 = sn * x0 + c * y0;
                    q11 = c * x1 - sn * y1; q12 = sn * x1 + c * y1;
                    q21 = c * x2 - sn * y2; q22 = sn * x2 + c * y2;
                }
            }
        }
    }
    // selection sort, columns follow: i = 0 picks the smallest of (d0, d1, d2), i = 1 the smaller of the remaining two
    {
        int k = 0;
        if (d1 < d0) k = 1;
        if (d2 < (k == 0 ? d0 : d1)) k = 2;
        if (k == 1) {
            const double t = d0; d0 = d1; d1 = t;
            if (VECTORS) { double u = q00; q00 = q01; q01 = u; u = q10; q10 = q11; q11 = u; u = q20; q20 = q21; q21 = u; }
        } else if (k == 2) {
            const double t = d0; d0 = d2; d2 = t;
            if (VECTORS) { double u = q00; q00 = q02; q02 = u; u = q10; q10 = q12; q12 = u; u = q20; q20 = q22; q22 = u; }
        }
        if (d2 < d1) {
            const double t = d1; d1 = d2; d2 = t;
            if (VECTORS) { double u = q01; q01 = q02; q02 = u; u = q11; q11 = q12; q12 = u; u = q21; q21 = q22; q22 = u; }
        }
    }
    s[0] = d0 * scale; s[1] = d1 * scale; s[2] = d2 * scale;
    if (VECTORS) { V[0][0] = q00; V[0][1] = q01; V[0][2] = q02; V[1][0] = q10; V[1][1] = q11; V[1][2] = q12; V[2][0] = q20; V[2][1] = q21; V[2][2] = q22; }
}
__host__ __device__ inline void eig33sym(const double K[3][3], double s[3], double V[3][3]) { eig33sym_t<true>(K, s, V); }

// ahc::PlaneSeg::Stats::compute (AHCPlaneSeg.hpp:148-183)
__host__ __device__ inline void plane_fit(const msl_peac_stats &st, double center[3], double normal[3], double &mse, double &curvature) {
    const double sc = ((double)1.0) / st.N;
    center[0] = st.sx * sc; center[1] = st.sy * sc; center[2] = st.sz * sc;
    double K[3][3] = {{st.sxx - st.sx * st.sx * sc, st.sxy - st.sx * st.sy * sc, st.sxz - st.sx * st.sz * sc},
                      {0, st.syy - st.sy * st.sy * sc, st.syz - st.sy * st.sz * sc},
                      {0, 0, st.szz - st.sz * st.sz * sc}};
    K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
    double sv[3], V[3][3];
    eig33sym(K, sv, V);
    const double sgn = (V[0][0] * center[0] + V[1][0] * center[1] + V[2][0] * center[2] <= 0) ? 1.0 : -1.0;   // normal towards the camera
    normal[0] = sgn > 0 ? V[0][0] : -V[0][0]; normal[1] = sgn > 0 ? V[1][0] : -V[1][0]; normal[2] = sgn > 0 ? V[2][0] : -V[2][0];
    mse = sv[0] * sc;
    curvature = sv[0] / (sv[0] + sv[1] + sv[2]);
}

// the MSE plane_fit would report, without centre / normal / curvature (the same K, the same eigenvalue bits)
__host__ __device__ inline double plane_mse(const msl_peac_stats &st) {
    const double sc = ((double)1.0) / st.N;
    double K[3][3] = {{st.sxx - st.sx * st.sx * sc, st.sxy - st.sx * st.sy * sc, st.sxz - st.sx * st.sz * sc},
                      {0, st.syy - st.sy * st.sy * sc, st.syz - st.sy * st.sz * sc},
                      {0, 0, st.szz - st.sz * st.sz * sc}};
    K[1][0] = K[0][1]; K[2][0] = K[0][2]; K[2][1] = K[1][2];
    double sv[3];
    eig33sym_t<false>(K, sv, nullptr);
    return sv[0] * sc;
}

// z of cloud vertex (row, col): (double)depth(2 row, 2 col) * depthMapFactor (src/PlaneExtractor.cpp:64)
__host__ __device__ inline double vertex_z(const uint16_t *img, size_t strideBytes, float factor, int row, int col) {
    const uint16_t d = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(img) + (size_t)(2 * row) * strideBytes + 2 * (size_t)(2 * col));
    return (double)d * factor;
}
__host__ __device__ inline void vertex_xy(float fx, float fy, float cx, float cy, int row, int col, double z, double &x, double &y) {
    x = ((double)(2 * col) - cx) * z / fx;   // :69
    y = ((double)(2 * row) - cy) * z / fy;   // :70
}

__global__ __launch_bounds__(256) void k_peac_cloud(PeacDev P) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.cw * P.ch) return;
    const int row = i / P.cw, col = i - row * P.cw;
    const uint16_t *img = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(P.depth) + (size_t)frame * P.frameStrideBytes);
    const double z = vertex_z(img, P.strideBytes, P.factor, row, col);
    double x, y;
    vertex_xy(P.fx, P.fy, P.cx, P.cy, row, col, z, x, y);
    double *o = P.cloud + ((size_t)frame * P.cw * P.ch + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

// Raw depth of the cloud's vertices (even rows / columns) packed to [frames][ch][cw]: what the host-side region growing reads --
// a quarter of the image, so device-resident input costs one small copy instead of the whole frame.
__global__ __launch_bounds__(256) void k_peac_half(PeacDev P, uint16_t *out) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.cw * P.ch) return;
    const int row = i / P.cw, col = i - row * P.cw;
    const uint8_t *img = reinterpret_cast<const uint8_t *>(P.depth) + (size_t)frame * P.frameStrideBytes;
    out[(size_t)frame * P.cw * P.ch + i] = *reinterpret_cast<const uint16_t *>(img + (size_t)(2 * row) * P.strideBytes + 2 * (size_t)(2 * col));
}

// One wave per window.  LDS: nine arrays of winW * winH products (a missing point contributes +0.0, which leaves every partial
// sum unchanged, so the additions that matter happen in the reference's raster order).
__global__ __launch_bounds__(64) void k_peac_fit(PeacDev P) {
    extern __shared__ double s_term[];   // [9][win]
    const int frame = blockIdx.y, b = blockIdx.x, lane = threadIdx.x;
    const uint16_t *img = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(P.depth) + (size_t)frame * P.frameStrideBytes);
    const int seedRow = (b / P.Nw) * P.winH, seedCol = (b % P.Nw) * P.winW, win = P.winW * P.winH;
    int nMissing = 0, nPushed = 0;
    bool broken = false;   // a valid point with a depth discontinuity towards its right / lower neighbour
    for (int p = lane; p < win; p += 64) {
        const int i = seedRow + p / P.winW, j = seedCol + p % P.winW;
        const double z = vertex_z(img, P.strideBytes, P.factor, i, j);
        double x = 0, y = 0;
        const bool has = z != 0;   // ImagePointCloud::get (include/PlaneExtractor.h:47-55): a 16-bit depth times a float is never NaN
        if (has) {
            vertex_xy(P.fx, P.fy, P.cx, P.cy, i, j, z, x, y);
            if (j + 1 < P.cw) { const double zn = vertex_z(img, P.strideBytes, P.factor, i, j + 1); if (zn != 0 && fabs(z - zn) > P.alpha * fabs(z) + P.tol) broken = true; }
            if (i + 1 < P.ch) { const double zn = vertex_z(img, P.strideBytes, P.factor, i + 1, j); if (zn != 0 && fabs(z - zn) > P.alpha * fabs(z) + P.tol) broken = true; }
            nPushed++;
        } else {
            nMissing++;
        }
        const double zz = has ? z : 0.0;
        s_term[0 * win + p] = x; s_term[1 * win + p] = y; s_term[2 * win + p] = zz;
        s_term[3 * win + p] = x * x; s_term[4 * win + p] = y * y; s_term[5 * win + p] = zz * zz;
        s_term[6 * win + p] = x * y; s_term[7 * win + p] = y * zz; s_term[8 * win + p] = x * zz;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { nMissing += __shfl_xor(nMissing, d, 64); nPushed += __shfl_xor(nPushed, d, 64); }
    // INIT_STRICT: any missing point rejects the window; INIT_LOOSE: the nanCntTh-th missing point does (AHCPlaneSeg.hpp:245-257)
    const bool valid = !__any(broken) && (P.loose ? nMissing < win / 2 : nMissing == 0);
    __builtin_amdgcn_wave_barrier();
    double sum = 0.0;
    if (valid && lane < 9) {
        const double *t = s_term + lane * win;
        for (int p = 0; p < win; p++) sum += t[p];   // Stats::push order (:81-92)
    }
    msl_peac_stats S;
    S.sx = __shfl(sum, 0, 64); S.sy = __shfl(sum, 1, 64); S.sz = __shfl(sum, 2, 64);
    S.sxx = __shfl(sum, 3, 64); S.syy = __shfl(sum, 4, 64); S.szz = __shfl(sum, 5, 64);
    S.sxy = __shfl(sum, 6, 64); S.syz = __shfl(sum, 7, 64); S.sxz = __shfl(sum, 8, 64);
    S.N = valid ? nPushed : 0; S.nouse = valid ? 0 : 1;
    if (lane != 0) return;
    msl_peac_block B;
    B.stats = S;
    B.center[0] = B.center[1] = B.center[2] = B.normal[0] = B.normal[1] = B.normal[2] = 0;
    if (S.N < 4) B.mse = B.curvature = __longlong_as_double(0x7FF8000000000000ll);   // quiet NaN (:279-280)
    else plane_fit(S, B.center, B.normal, B.mse, B.curvature);
    P.blocks[(size_t)frame * P.Nw * P.Nh + b] = B;
}

// ---- agglomerative clustering on the device: one wave per frame ----------------------------------------------------------------------------
// ahCluster (AHCPlaneFitter.hpp:939-1143) is a sequential chain of pops, but every pop evaluates ALL neighbours of the popped node (a plane fit
// each: ~30 000 3x3 eigen-solves per 640x480 frame) -- that part is data parallel, and frames are independent.  One wave per frame: lane 0 keeps the
// reference's binary heap (libstdc++ push_heap / pop_heap spelled out, so the pop order is the reference's even among equal keys) and the disjoint
// set in LDS; the neighbours of the popped node are fitted one per lane with the same __host__ __device__ code the host uses (so the same bits),
// the reference's first-minimum rule picks the merge, and the adjacency -- a bit matrix, whose ascending bit order is the reference's ordered
// neighbour set -- is updated by all lanes.  The host supplies the initial heap and edges (graph initialisation needs cos()) and continues with the
// extracted planes (erosion, FIFO region growing: order dependent pixel work that stays on the host).
struct PlaneOut { double st[9], center[3], normal[3], mse; int32_t id, N, rid, _pad; };
struct ClusterDev {
    int nB, maxN, words, minSupport, maxStep, maxE, maxPl;
    double depthSigma, stdTolMerge, simMerge;
    const msl_peac_block *blocks;   // [frames][nB]
    unsigned *rows;                 // [frames][maxN][words] adjacency bit matrix
    double *gst;                    // [frames][maxN][9] statistics of every node
    double *gcxy;                   // [frames][maxN][2] centre x, y (output only)
    const int *heap0, *heapCount;   // [frames][nB], [frames] initial heap
    const int *edges, *edgeCount;   // [frames][maxE][2], [frames] initial edges
    int *nPlanes; PlaneOut *planes; // [frames] (-1: more than maxPl), [frames][maxPl] extracted planes in extraction order
    int *parent, *setSize;          // [frames][nB] disjoint set after the clustering
};
__device__ __forceinline__ unsigned ld_ag(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_ag(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_agd(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agd(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Minimum of a double over the 64 lanes (DPP row shifts and row broadcasts, as msl::wave_incl_scan; lanes without a source see +inf); every lane gets it.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_min_step(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROWMASK, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)0x7FF00000, (int)(unsigned)(u >> 32), CTRL, ROWMASK, 0xF, false);
    return fmin(v, __longlong_as_double(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_min_d(double v) {
    v = dpp_min_step<0x111, 0xF>(v); v = dpp_min_step<0x112, 0xF>(v); v = dpp_min_step<0x114, 0xF>(v); v = dpp_min_step<0x118, 0xF>(v);
    v = dpp_min_step<0x142, 0xA>(v);   // row_bcast:15 -> rows 1, 3
    v = dpp_min_step<0x143, 0xC>(v);   // row_bcast:31 -> rows 2, 3
    return __shfl(v, 63, 64);
}

__global__ __launch_bounds__(64) void k_peac_cluster(ClusterDev C) {
    extern __shared__ double s_mem[];
    const int f = blockIdx.x, lane = threadIdx.x, nB = C.nB, maxN = C.maxN, words = C.words;
    double *s_mse = s_mem, *s_nx = s_mse + maxN, *s_ny = s_nx + maxN, *s_nz = s_ny + maxN, *s_cz = s_nz + maxN;
    double *s_hkey = s_cz + maxN;   // MSE of the node in heap slot i (saves the dependent s_mse[s_heap[i]] read on every comparison)
    int *s_N = reinterpret_cast<int *>(s_hkey + maxN), *s_rid = s_N + maxN, *s_heap = s_rid + maxN, *s_cand = s_heap + maxN;
    int *s_parent = s_cand + maxN, *s_size = s_parent + nB;
    uint8_t *s_nouse = reinterpret_cast<uint8_t *>(s_size + nB);
    unsigned *rows = C.rows + (size_t)f * maxN * words;
    double *gst = C.gst + (size_t)f * maxN * 9, *gcxy = C.gcxy + (size_t)f * maxN * 2;
    const msl_peac_block *blocks = C.blocks + (size_t)f * nB;
    PlaneOut *planes = C.planes + (size_t)f * C.maxPl;
    for (int b = lane; b < nB; b += 64) {
        const msl_peac_block &B = blocks[b];
        const bool nouse = B.stats.nouse != 0;
        s_mse[b] = B.mse; s_nx[b] = B.normal[0]; s_ny[b] = B.normal[1]; s_nz[b] = B.normal[2]; s_cz[b] = B.center[2];
        s_nouse[b] = nouse ? 1 : 0; s_N[b] = nouse ? 0 : B.stats.N; s_rid[b] = b; s_parent[b] = b; s_size[b] = 1;
        double *g = gst + (size_t)b * 9;
        g[0] = B.stats.sx; g[1] = B.stats.sy; g[2] = B.stats.sz; g[3] = B.stats.sxx; g[4] = B.stats.syy; g[5] = B.stats.szz; g[6] = B.stats.sxy; g[7] = B.stats.syz; g[8] = B.stats.sxz;
        gcxy[2 * b] = B.center[0]; gcxy[2 * b + 1] = B.center[1];
    }
    int hcount = C.heapCount[f];
    for (int i = lane; i < hcount; i += 64) { const int id = C.heap0[(size_t)f * nB + i]; s_heap[i] = id; s_hkey[i] = blocks[id].mse; }
    {
        const int ne = C.edgeCount[f];
        const int *E = C.edges + (size_t)f * C.maxE * 2;
        for (int e = lane; e < ne; e += 64) {
            const int a = E[2 * e], b = E[2 * e + 1];
            atomicOr(&rows[(size_t)a * words + (b >> 5)], 1u << (b & 31));
            atomicOr(&rows[(size_t)b * words + (a >> 5)], 1u << (a & 31));
        }
    }
    __threadfence();   // (once: the plain initialisation stores above become visible to the agent-scope accesses below)
    __syncthreads();
    int nNodes = nB, nPl = 0, step = 0;
    // the set bits of `words` words held one per lane (chunks of 64 words), ascending, into s_cand; returns their number
    auto list_bits = [&](auto word_of) -> int {
        int base = 0;
        for (int w0 = 0; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            unsigned bits = w < words ? word_of(w) : 0u;
            const unsigned cnt = (unsigned)__popc(bits);
            const unsigned incl = wave_incl_scan(cnt);
            int o = base + (int)(incl - cnt);
            while (bits) { const int b = __ffs((int)bits) - 1; bits &= bits - 1; s_cand[o++] = w * 32 + b; }
            base += __shfl((int)incl, 63, 64);
        }
        __syncthreads();
        return base;
    };
    // PlaneSegMinMSECmp(a, b) = mse[b] < mse[a]; comp(parent, value) in __push_heap reads "value's key < parent's key"
    auto heap_push = [&](int id, double key) {   // std::push_heap (libstdc++ __push_heap), lane 0
        int hole = hcount++;
        int parent = (hole - 1) / 2;
        while (hole > 0 && key < s_hkey[parent]) { s_heap[hole] = s_heap[parent]; s_hkey[hole] = s_hkey[parent]; hole = parent; parent = (hole - 1) / 2; }
        s_heap[hole] = id; s_hkey[hole] = key;
    };
    auto heap_pop = [&]() -> int {   // top, then std::pop_heap (libstdc++ __pop_heap / __adjust_heap) + pop_back, lane 0
        const int top = s_heap[0];
        const int len = --hcount;
        if (len > 0) {
            const int value = s_heap[len];
            const double vkey = s_hkey[len];
            int hole = 0, child = 0;
            while (child < (len - 1) / 2) {
                child = 2 * (child + 1);
                if (s_hkey[child - 1] < s_hkey[child]) child--;   // comp(first[child], first[child - 1])
                s_heap[hole] = s_heap[child]; s_hkey[hole] = s_hkey[child]; hole = child;
            }
            if ((len & 1) == 0 && child == (len - 2) / 2) { child = 2 * (child + 1); s_heap[hole] = s_heap[child - 1]; s_hkey[hole] = s_hkey[child - 1]; hole = child - 1; }
            int parent = (hole - 1) / 2;
            while (hole > 0 && vkey < s_hkey[parent]) { s_heap[hole] = s_heap[parent]; s_hkey[hole] = s_hkey[parent]; hole = parent; parent = (hole - 1) / 2; }
            s_heap[hole] = value; s_hkey[hole] = vkey;
        }
        return top;
    };
    auto ds_find = [&](int x) { while (s_parent[x] != x) { s_parent[x] = s_parent[s_parent[x]]; x = s_parent[x]; } return x; };
    auto emit_plane = [&](int p) {   // extractedPlanes.push_back
        if (nPl >= C.maxPl) { nPl++; return; }
        PlaneOut &O = planes[nPl];
        if (lane < 9) O.st[lane] = ld_agd(&gst[(size_t)p * 9 + lane]);
        if (lane == 9) { O.center[0] = ld_agd(&gcxy[2 * p]); O.center[1] = ld_agd(&gcxy[2 * p + 1]); O.center[2] = s_cz[p]; }
        if (lane == 10) { O.normal[0] = s_nx[p]; O.normal[1] = s_ny[p]; O.normal[2] = s_nz[p]; O.mse = s_mse[p]; }
        if (lane == 11) { O.id = p; O.N = s_N[p]; O.rid = s_rid[p]; O._pad = 0; }
        nPl++;
    };
    // Rows, statistics and centres are read and written at agent scope (L2) only, so program order within the wave plus completion of
    // the outstanding stores / atomics is all the ordering the next step needs -- no cache write-back or invalidation.
    auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    auto isolate = [&](int p, int nc) {   // the neighbours of p are s_cand[0..nc)
        const unsigned clr = ~(1u << (p & 31));
        for (int i = lane; i < nc; i += 64) atomicAnd(&rows[(size_t)s_cand[i] * words + (p >> 5)], clr);
        for (int w = lane; w < words; w += 64) st_ag(&rows[(size_t)p * words + w], 0u);
        drain();
    };
    while (hcount > 0 && step <= C.maxStep) {
        int p = 0;
        if (lane == 0) p = heap_pop();
        p = __shfl(p, 0, 64);
        hcount = __shfl(hcount, 0, 64);
        __syncthreads();
        if (s_nouse[p]) continue;
        const int nc = list_bits([&](int w) { return ld_ag(&rows[(size_t)p * words + w]); });
        // every candidate merge fitted by one lane; the reference's rule (first strict minimum of the MSE, ascending neighbour order) picks one
        bool have = false;
        double bMse = 0, bC0 = 0, bC1 = 0, bC2 = 0, bN0 = 0, bN1 = 0, bN2 = 0;
        int bNb = -1, bN = 0;
        const double pnx = s_nx[p], pny = s_ny[p], pnz = s_nz[p];
        double ps[9];
#pragma unroll
        for (int k = 0; k < 9; k++) ps[k] = ld_agd(&gst[(size_t)p * 9 + k]);
        const int pN = s_N[p];
        for (int c0 = 0; c0 < nc; c0 += 64) {
            const int ci = c0 + lane;
            const bool in = ci < nc;
            const int nb = in ? s_cand[ci] : p;
            const bool pass = in && fabs(pnx * s_nx[nb] + pny * s_ny[nb] + pnz * s_nz[nb]) >= C.simMerge;
            double mMse = 0, cen[3] = {0, 0, 0}, nrm[3] = {0, 0, 0};
            int mN = 0;
            if (pass) {
                msl_peac_stats t;
                double y[9];
#pragma unroll
                for (int k = 0; k < 9; k++) y[k] = ld_agd(&gst[(size_t)nb * 9 + k]);
                t.sx = ps[0] + y[0]; t.sy = ps[1] + y[1]; t.sz = ps[2] + y[2]; t.sxx = ps[3] + y[3]; t.syy = ps[4] + y[4]; t.szz = ps[5] + y[5];
                t.sxy = ps[6] + y[6]; t.syz = ps[7] + y[7]; t.sxz = ps[8] + y[8]; t.N = pN + s_N[nb]; t.nouse = 0;
                double curv;
                plane_fit(t, cen, nrm, mMse, curv);
                mN = t.N;
            }
            const unsigned long long pm = __ballot(pass);
            const int lim = min(64, nc - c0);
            int jBest = -1;
            // The rule walks the candidates in order and keeps the first strict minimum.  Without NaNs and without ties that is the lane of the
            // wave-wide minimum; anything else (never seen on real data) replays the walk literally.
            const double minv = wave_min_d(pass ? mMse : __builtin_inf());
            const unsigned long long eq = __ballot(pass && mMse == minv), nanm = __ballot(pass && mMse != mMse);
            if (pm && !nanm && __popcll(eq) == 1 && !(have && (bMse == minv || bMse != bMse))) {
                if (!have || bMse > minv) { jBest = __ffsll((long long)eq) - 1; have = true; bMse = minv; bN = __shfl(mN, jBest, 64); }
            } else {
                for (int j = 0; j < lim; j++) {
                    if (!((pm >> j) & 1ull)) continue;
                    const double m = __shfl(mMse, j, 64);
                    if (!have || bMse > m || (bMse == m && (double)bN < m)) { have = true; bMse = m; bN = __shfl(mN, j, 64); jBest = j; }   // (sic: N against mse, :1005)
                }
            }
            if (jBest >= 0) {   // the best so far lies in this chunk: fetch the rest of its fit
                bNb = __shfl(nb, jBest, 64);
                bC0 = __shfl(cen[0], jBest, 64); bC1 = __shfl(cen[1], jBest, 64); bC2 = __shfl(cen[2], jBest, 64);
                bN0 = __shfl(nrm[0], jBest, 64); bN1 = __shfl(nrm[1], jBest, 64); bN2 = __shfl(nrm[2], jBest, 64);
            }
        }
        const double tm = C.depthSigma * bC2 * bC2 + C.stdTolMerge;   // ParamSet::T_mse(P_MERGING): pow(.., 2) is the product
        if (have && bMse < tm * tm) {
            const int id = nNodes++, nb = bNb;
            if (lane < 9) st_agd(&gst[(size_t)id * 9 + lane], ld_agd(&gst[(size_t)p * 9 + lane]) + ld_agd(&gst[(size_t)nb * 9 + lane]));
            if (lane == 0) {
                s_mse[id] = bMse; s_nx[id] = bN0; s_ny[id] = bN1; s_nz[id] = bN2; s_cz[id] = bC2; s_N[id] = bN; s_nouse[id] = 0;
                s_rid[id] = s_N[p] >= s_N[nb] ? s_rid[p] : s_rid[nb];
                st_agd(&gcxy[2 * id], bC0); st_agd(&gcxy[2 * id + 1], bC1);
                heap_push(id, bMse);
                // mergeNbsFrom (AHCPlaneSeg.hpp:398-436): union of the two disjoint sets
                const int xr = ds_find(s_rid[p]), yr = ds_find(s_rid[nb]);
                if (xr != yr) {
                    if (s_size[xr] < s_size[yr]) { s_parent[xr] = yr; s_size[yr] += s_size[xr]; }
                    else { s_parent[yr] = xr; s_size[xr] += s_size[yr]; }
                }
                s_nouse[p] = 1; s_nouse[nb] = 1;
            }
            hcount = __shfl(hcount, 0, 64);
            // neighbours of the new node = union of both neighbour sets without the two merged nodes; both old rows are cleared
            const int nu = list_bits([&](int w) {
                unsigned u = ld_ag(&rows[(size_t)p * words + w]) | ld_ag(&rows[(size_t)nb * words + w]);
                if (w == (p >> 5)) u &= ~(1u << (p & 31));
                if (w == (nb >> 5)) u &= ~(1u << (nb & 31));
                st_ag(&rows[(size_t)id * words + w], u);
                st_ag(&rows[(size_t)p * words + w], 0u); st_ag(&rows[(size_t)nb * words + w], 0u);
                return u;
            });
            for (int i = lane; i < nu; i += 64) {
                unsigned *r = rows + (size_t)s_cand[i] * words;
                atomicOr(&r[id >> 5], 1u << (id & 31));
                atomicAnd(&r[p >> 5], ~(1u << (p & 31)));
                atomicAnd(&r[nb >> 5], ~(1u << (nb & 31)));
            }
            drain();
        } else {
            if (s_N[p] >= C.minSupport) emit_plane(p);
            isolate(p, nc);
        }
        __syncthreads();
        ++step;
    }
    while (hcount > 0) {   // (only reached when max_step stopped the loop above; the reference does not test nouse here either)
        int p = 0;
        if (lane == 0) p = heap_pop();
        p = __shfl(p, 0, 64);
        hcount = __shfl(hcount, 0, 64);
        __syncthreads();
        if (s_N[p] >= C.minSupport) emit_plane(p);
        const int nc = list_bits([&](int w) { return ld_ag(&rows[(size_t)p * words + w]); });
        isolate(p, nc);
        __syncthreads();
    }
    if (lane == 0) C.nPlanes[f] = nPl <= C.maxPl ? nPl : -1;
    for (int b = lane; b < nB; b += 64) { C.parent[(size_t)f * nB + b] = s_parent[b]; C.setSize[(size_t)f * nB + b] = s_size[b]; }
}

// ---- host side: agglomerative clustering over the block graph -----------------------------------------------------------------
struct Thresholds {
    msl_peac_params p;
    double t_mse_init(double z) const { return std::pow(p.depth_sigma * z * z + p.std_tol_init, 2); }     // ParamSet::T_mse (AHCParamSet.hpp:87-99)
    double t_mse_merge(double z) const { return std::pow(p.depth_sigma * z * z + p.std_tol_merge, 2); }
    double t_ang_init(double z) const {                                                                   // ParamSet::T_ang (:111-131)
        double clipped_z = z;
        clipped_z = std::max(clipped_z, p.z_near);
        clipped_z = std::min(clipped_z, p.z_far);
        const double factor = (p.angle_far - p.angle_near) / (p.z_far - p.z_near);
        return std::cos(factor * clipped_z + p.angle_near - factor * p.z_near);
    }
};

// ---- the MSE of several candidate merges at once (host SIMD) ------------------------------------------------------------------------------
// ahCluster fits a plane to every neighbour's merged statistics before it picks one (AHCPlaneFitter.hpp:985-1010): for a frame that is ~30 000
// 3x3 eigenvalue problems, all but ~1 500 of them discarded, and the whole latency of a single-frame call.  The lanes below run plane_mse() for
// VW candidates in lock-step: every lane performs exactly the scalar sequence of IEEE double operations of eig33sym_t<false> (same expressions,
// same association, no contraction; divisions and square roots are correctly rounded in either form), branches become selects, and a lane whose
// QR iteration has finished is masked, so the result is the scalar result bit for bit (tests/test_peac_host.py compares both on random and
// degenerate matrices, and the whole segmentation against the oracle with every width).
template <int VW> struct Lanes {
    typedef double D __attribute__((ext_vector_type(VW)));
    typedef long M __attribute__((ext_vector_type(VW)));   // comparison results: all ones / zero per lane
};
#define MSL_SEL(m, a, b) ((m) ? (a) : (b))

// in: 10 rows of VW doubles (sx sy sz sxx syy szz sxy syz sxz N); out: VW MSEs
template <int VW>
__attribute__((always_inline)) inline void plane_mse_lanes(const double *in, double *out) {
    typedef typename Lanes<VW>::D D;
    typedef typename Lanes<VW>::M M;
    D r[10];
    for (int i = 0; i < 10; i++) __builtin_memcpy(&r[i], in + (size_t)i * VW, sizeof(D));
    const D zero = 0.0, one = 1.0;
    const M izero = 0, ione = 1, itwo = 2;
    const D sc = one / r[9];
    D a00 = r[3] - r[0] * r[0] * sc, a10 = r[6] - r[0] * r[1] * sc, a20 = r[8] - r[0] * r[2] * sc;
    D a11 = r[4] - r[1] * r[1] * sc, a21 = r[7] - r[1] * r[2] * sc, a22 = r[5] - r[2] * r[2] * sc;
    // The helpers take and return 256- to 1024-bit vectors by value, and a lambda's call operator does not inherit the target("avx2" / "avx512f")
    // attribute of the wrapper this template is inlined into: if the inliner ever declined, the call would cross an ABI boundary between feature
    // sets (-Wpsabi) and could silently change the bits.  always_inline removes the dependence on heuristics; the build adds -Werror=psabi.
    auto vabs = [](D v) __attribute__((always_inline)) { return __builtin_elementwise_abs(v); };
    auto vmax = [](D a, D b) __attribute__((always_inline)) { return __builtin_elementwise_max(a, b); };   // fmax: a NaN operand is ignored
    auto vsqrt = [](D v) __attribute__((always_inline)) { return __builtin_elementwise_sqrt(v); };
    D scale = vmax(vmax(vmax(vabs(a00), vabs(a10)), vmax(vabs(a11), vabs(a20))), vmax(vabs(a21), vabs(a22)));
    scale = MSL_SEL(scale == zero, one, scale);
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    const D tiny = 2.2250738585072014e-308, precision = 2.0 * 2.220446049250313e-16;
    const D v1norm2 = a20 * a20;
    const M small = v1norm2 <= tiny;
    const D beta = vsqrt(a10 * a10 + v1norm2), invBeta = one / beta, m01 = a10 * invBeta, m02 = a20 * invBeta;
    const D qq = (D)2.0 * m01 * a21 + m02 * (a22 - a11);
    D dg0 = a00, dg1 = MSL_SEL(small, a11, a11 + m02 * qq), dg2 = MSL_SEL(small, a22, a22 - m02 * qq);
    D sb0 = MSL_SEL(small, a10, beta), sb1 = MSL_SEL(small, a21, a21 - m01 * qq);
    M end = itwo, start = izero, iter = izero, active = ~izero;
    // Givens rotation that annihilates z against x: the three scalar cases share one division, one square root and one reciprocal
    auto givens = [&](D x, D z, D &c, D &sn) __attribute__((always_inline)) {
        const M big = vabs(x) > vabs(z);
        const D num = MSL_SEL(big, z, x), den = MSL_SEL(big, x, z);
        const D t = num / den;
        D u = vsqrt(one + t * t);
        u = MSL_SEL(den < zero, -u, u);
        const D rr = MSL_SEL(big, one, -one) / u, oo = -t * rr;   // |x| > |z|: c = 1 / u, sn = -t c; otherwise sn = -1 / u, c = -t sn
        c = MSL_SEL(big, rr, oo); sn = MSL_SEL(big, oo, rr);
        const M x0 = x == zero;
        c = MSL_SEL(x0, zero, c); sn = MSL_SEL(x0, MSL_SEL(z < zero, one, -one), sn);
    };
    for (;;) {
        const M c0 = active & (start <= izero) & (end > izero) & ((vabs(sb0) <= (vabs(dg0) + vabs(dg1)) * precision) | (vabs(sb0) <= tiny));
        sb0 = MSL_SEL(c0, zero, sb0);
        const M c1 = active & (start <= ione) & (end > ione) & ((vabs(sb1) <= (vabs(dg1) + vabs(dg2)) * precision) | (vabs(sb1) <= tiny));
        sb1 = MSL_SEL(c1, zero, sb1);
        end = MSL_SEL(active & (end == itwo) & (sb1 == zero), ione, end);
        end = MSL_SEL(active & (end == ione) & (sb0 == zero), izero, end);
        active &= end > izero;
        iter = MSL_SEL(active, iter + ione, iter);
        active &= ~(iter > (M)90);
        if (!__builtin_reduce_or(active)) break;
        start = end - ione;
        start = MSL_SEL((start == ione) & (sb0 != zero), izero, start);
        const M e2m = end == itwo;
        const D dEnd = MSL_SEL(e2m, dg2, dg1), dEm1 = MSL_SEL(e2m, dg1, dg0), e = MSL_SEL(e2m, sb1, sb0);
        const D td = (dEm1 - dEnd) * (D)0.5;
        const D ax = vabs(td), ay = vabs(e);
        const M gt = ax > ay;
        const D pp = MSL_SEL(gt, ax, ay), qp = MSL_SEL(gt, ay, ax) / pp;
        const D h = MSL_SEL(pp == zero, zero, pp * vsqrt(one + qp * qp));
        const D e2 = e * e, denom = td + MSL_SEL(td > zero, h, -h);
        const D muA = dEnd - vabs(e), muC = dEnd - e2 / denom;
        D mu = MSL_SEL(td == zero, muA, MSL_SEL(e != zero, muC, dEnd));
        const M under = active & (td != zero) & (e != zero) & (e2 == zero);   // e * e underflowed: the scalar code divides twice instead
        if (__builtin_reduce_or(under)) mu = MSL_SEL(under, dEnd - e / (denom / e), mu);
        const M s0 = start == izero;
        D x = MSL_SEL(s0, dg0, dg1) - mu, z = MSL_SEL(s0, sb0, sb1);
        const M doK0 = active & s0 & (z != zero);
        if (__builtin_reduce_or(doK0)) {   // k = 0
            const M doK = doK0;
            D c, sn;
            givens(x, z, c, sn);
            const D sdk = sn * dg0 + c * sb0, dkp1 = sn * sb0 + c * dg1;
            const D n0 = c * (c * dg0 - sn * sb0) - sn * (c * sb0 - sn * dg1), n1 = sn * sdk + c * dkp1, nsb = c * sdk - sn * dkp1;
            dg0 = MSL_SEL(doK, n0, dg0); dg1 = MSL_SEL(doK, n1, dg1); sb0 = MSL_SEL(doK, nsb, sb0);
            x = MSL_SEL(doK, nsb, x);
            const M more = doK & e2m;                      // k < end - 1
            const D nz = -sn * sb1, nsb1 = c * sb1;
            // a lane that ran k = 0 with end == 1 has left the scalar loop: end > 1 below keeps it out of k = 1
            z = MSL_SEL(more, nz, z); sb1 = MSL_SEL(more, nsb1, sb1);
        }
        const M doK1 = active & e2m & (z != zero);
        if (__builtin_reduce_or(doK1)) {   // k = 1 (a lane that skipped k = 0 at start == 0 did so with z == 0, which also ends its loop here)
            const M doK = doK1;
            D c, sn;
            givens(x, z, c, sn);
            const D sdk = sn * dg1 + c * sb1, dkp1 = sn * sb1 + c * dg2;
            const D n1 = c * (c * dg1 - sn * sb1) - sn * (c * sb1 - sn * dg2), n2 = sn * sdk + c * dkp1, nsb = c * sdk - sn * dkp1;
            sb0 = MSL_SEL(doK & s0, c * sb0 - sn * z, sb0);    // k > start
            dg1 = MSL_SEL(doK, n1, dg1); dg2 = MSL_SEL(doK, n2, dg2); sb1 = MSL_SEL(doK, nsb, sb1);
        }
    }
    const D lo01 = MSL_SEL(dg1 < dg0, dg1, dg0), lo = MSL_SEL(dg2 < lo01, dg2, lo01);   // s[0] of the selection sort
    const D mse = lo * scale * sc;
    __builtin_memcpy(out, &mse, sizeof(D));
}

#if defined(__HIP_DEVICE_COMPILE__)
#define MSL_TARGET(t)
inline int host_simd_level() { return 2; }
#else
#define MSL_TARGET(t) __attribute__((target(t)))
// instruction set the lanes may use: 8 = AVX-512F, 4 = AVX2, 2 = the x86-64 baseline (SSE2), 0 = the scalar code; MSL_PEAC_SIMD lowers it
inline int host_simd_level() {
    static const int w = [] {
        int best = __builtin_cpu_supports("avx512f") ? 8 : __builtin_cpu_supports("avx2") ? 4 : 2;
        if (const char *e = getenv("MSL_PEAC_SIMD")) { const int v = atoi(e); if (v == 0 || v == 2 || v == 4 || v == 8) best = std::min(best, v); }
        return best;
    }();
    return w;
}
#endif
inline int host_lane_cap() {   // MSL_PEAC_LANES = 2 / 4 / 8 / 16 caps the candidates per group (tests run every width)
    static const int c = [] { const char *e = getenv("MSL_PEAC_LANES"); const int v = e ? atoi(e) : 16; return v == 2 || v == 4 || v == 8 ? v : 16; }();
    return c;
}
// The solver is a single dependent chain of divisions and square roots, so a group twice as wide as the registers (two independent chains the
// core interleaves) costs little more than one register's worth: 16 lanes on AVX-512, 8 on AVX2.
void plane_mse_x2(const double *in, double *out) { plane_mse_lanes<2>(in, out); }
MSL_TARGET("avx2") void plane_mse_x4(const double *in, double *out) { plane_mse_lanes<4>(in, out); }
MSL_TARGET("avx2") void plane_mse_x8_avx2(const double *in, double *out) { plane_mse_lanes<8>(in, out); }
MSL_TARGET("avx512f") void plane_mse_x8(const double *in, double *out) { plane_mse_lanes<8>(in, out); }
MSL_TARGET("avx512f") void plane_mse_x16(const double *in, double *out) { plane_mse_lanes<16>(in, out); }
// lanes for a group when `left` candidates remain (0: scalar)
inline int lanes_for(size_t left) {
    const int simd = host_simd_level(), cap = host_lane_cap();
    if (left < 2 || simd == 0) return 0;
    int vw = 2;
    if (simd >= 4 && left > 2) vw = 4;
    if (simd >= 4 && left > 4) vw = 8;
    if (simd >= 8 && left > 8) vw = 16;
    return std::min(vw, cap);
}
inline void plane_mse_group(int vw, const double *in, double *out) {
    if (vw == 16) plane_mse_x16(in, out);
    else if (vw == 8) { if (host_simd_level() >= 8) plane_mse_x8(in, out); else plane_mse_x8_avx2(in, out); }
    else if (vw == 4) plane_mse_x4(in, out);
    else plane_mse_x2(in, out);
}

struct Node {
    msl_peac_stats st;
    double center[3], normal[3], mse, curvature;
    int N, rid;
    bool nouse;
    std::vector<int> nbs;   // adjacent node ids, ascending (= the reference's std::set<PlaneSeg*> with addresses pinned to creation order)
};

// One object per worker thread, reused for every frame that thread segments: all containers keep their capacity, so the steady state allocates
// nothing (64 threads that each mmap / munmap a few hundred KB per frame serialise on the process's address-space lock).
// Optional per-frame outputs beyond the membership image: what PlaneDetection hands on (extractedPlanes, plane_vertices_)
struct PlaneSink { msl_peac_plane *planes; int32_t *offsets, *indices; int maxPlanes; bool overflow; };

class alignas(128) FrameSegmenter {   // own cache lines: the vectors' end pointers inside the object change on every push
public:
    void set_sink(PlaneSink *s) { sink_ = s; }
    void configure(const msl_peac_params &prm, const uint16_t *halfDepth /* [ch][cw] raw depth of the cloud vertices */, int cw, int ch, float fx, float fy,
                   float cx, float cy, float factor) {
        T.p = prm; img_ = halfDepth; W = cw; H = ch; fx_ = fx; fy_ = fy; cx_ = cx; cy_ = cy; factor_ = factor;
        winW = prm.window_w; winH = prm.window_h; Nw = cw / prm.window_w; Nh = ch / prm.window_h;
    }

    // returns the number of extracted planes; member[H * W] receives PlaneFitter::membershipImg
    // Device-clustering path, phase 1: graph initialisation only (AHCPlaneFitter.hpp:756-928); the initial heap (in the order the pushes left
    // it) and the edge list go to k_peac_cluster.  Returns false if the edge list does not fit.
    bool graph_for_device(const msl_peac_block *blocks, int *heapOut, int *heapCount, int *edgesOut, int *edgeCount, int maxE) {
        parent_.resize((size_t)Nw * Nh); setSize_.assign((size_t)Nw * Nh, 1);
        for (size_t i = 0; i < parent_.size(); i++) parent_[i] = (int)i;
        nNodes_ = 0; planes_.clear(); growQ_.clear(); heap_.clear();
        edges_.clear(); recordEdges_ = true;
        build_graph(blocks);
        recordEdges_ = false;
        if ((int)edges_.size() / 2 > maxE) return false;
        std::copy(heap_.begin(), heap_.end(), heapOut); *heapCount = (int)heap_.size();
        std::copy(edges_.begin(), edges_.end(), edgesOut); *edgeCount = (int)edges_.size() / 2;
        return true;
    }
    // Phase 2: the planes k_peac_cluster extracted (extraction order) and its disjoint set; erosion, region growing and the final merge follow as
    // in run().  Only the plane nodes exist here; their ids keep the order of the original ids (the neighbour sets iterate in id order).
    int finish_from_device(const PlaneOut *pl, int np, const int *parent, const int *setSize, int32_t *member) {
        parent_.assign(parent, parent + (size_t)Nw * Nh); setSize_.assign(setSize, setSize + (size_t)Nw * Nh);
        nNodes_ = 0; planes_.clear(); growQ_.clear(); heap_.clear();
        std::vector<int> &order = relabel_, &newId = oldPlanes_;
        order.resize(np); newId.resize(np);
        for (int i = 0; i < np; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [pl](int a, int b) { return pl[a].id < pl[b].id; });
        for (int k = 0; k < np; k++) {
            const PlaneOut &O = pl[order[k]];
            Node nd;
            nd.st.sx = O.st[0]; nd.st.sy = O.st[1]; nd.st.sz = O.st[2]; nd.st.sxx = O.st[3]; nd.st.syy = O.st[4]; nd.st.szz = O.st[5];
            nd.st.sxy = O.st[6]; nd.st.syz = O.st[7]; nd.st.sxz = O.st[8]; nd.st.N = O.N; nd.st.nouse = 0;
            for (int c = 0; c < 3; c++) { nd.center[c] = O.center[c]; nd.normal[c] = O.normal[c]; }
            nd.mse = O.mse; nd.curvature = 0; nd.N = O.N; nd.rid = O.rid; nd.nouse = false;
            newId[order[k]] = add_node(nd);
        }
        for (int i = 0; i < np; i++) planes_.push_back(newId[i]);
        std::sort(planes_.begin(), planes_.end(), [this](int a, int b) { return nodes_[b].N < nodes_[a].N; });   // PlaneSegSizeCmp, as at the end of cluster()
        member_ = member;
        std::fill(member, member + (size_t)W * H, -1);
        if (T.p.do_refine) refine();
        emit_planes();
        return (int)planes_.size();
    }

    int run(const msl_peac_block *blocks, int32_t *member) {
        const char *tenv = getenv("MSL_PEAC_TIMING");
        const bool timing = tenv && atoi(tenv) >= 2;
        auto now = []() { return std::chrono::steady_clock::now(); };
        auto t0 = now();
        parent_.resize((size_t)Nw * Nh); setSize_.assign((size_t)Nw * Nh, 1);
        for (size_t i = 0; i < parent_.size(); i++) parent_[i] = (int)i;
        nNodes_ = 0; planes_.clear(); growQ_.clear(); heap_.clear();
        build_graph(blocks);
        auto t1 = now();
        cluster();
        auto t2 = now();
        member_ = member;
        std::fill(member, member + (size_t)W * H, -1);
        if (T.p.do_refine) refine();
        auto t3 = now();
        if (timing) {
            auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
            fprintf(stderr, "[msl_peac] graph %ld us, cluster %ld us (%zu nodes), refine %ld us (queue %zu)\n", us(t0, t1), us(t1, t2), (size_t)nNodes_, us(t2, t3), growQ_.size());
        }
        emit_planes();
        return (int)planes_.size();
    }

private:
    struct MseGreater {   // PlaneSegMinMSECmp: the queue's top is the node with the smallest MSE
        const FrameSegmenter *f;
        bool operator()(int a, int b) const { return f->nodes_[b].mse < f->nodes_[a].mse; }
    };
    // std::priority_queue<int, std::vector<int>, MseGreater> spelled out (push_heap / pop_heap on a member vector: the same sequence of
    // comparisons, hence the same order among equal keys, without a fresh container per run)
    std::vector<int> heap_;
    void heap_push(int id) { heap_.push_back(id); std::push_heap(heap_.begin(), heap_.end(), MseGreater{this}); }
    int heap_pop() { std::pop_heap(heap_.begin(), heap_.end(), MseGreater{this}); const int id = heap_.back(); heap_.pop_back(); return id; }

    Thresholds T;
    const uint16_t *img_ = nullptr;
    int W = 0, H = 0; float fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0, factor_ = 0;
    int winW = 0, winH = 0, Nw = 0, Nh = 0;
    std::vector<Node> nodes_;                 // node pool: [0, nNodes_) are live; the rest keep their neighbour vectors' capacity for the next frame
    int nNodes_ = 0;
    std::vector<int> G_, u_, oldPlanes_, relabel_, edges_;
    bool recordEdges_ = false;
    PlaneSink *sink_ = nullptr;
    void emit_planes() {   // plane_filter.extractedPlanes as Frame::ExtractPlanes reads them (src/Frame.cc:626-632)
        if (!sink_) return;
        if ((int)planes_.size() > sink_->maxPlanes) { sink_->overflow = true; return; }
        for (size_t j = 0; j < planes_.size(); j++) {
            const Node &nd = nodes_[planes_[j]];
            msl_peac_plane &o = sink_->planes[j];
            for (int c = 0; c < 3; c++) { o.normal[c] = nd.normal[c]; o.center[c] = nd.center[c]; }
            o.mse = nd.mse; o.N = nd.N; o._pad = 0;
        }
        if (sink_->offsets && !T.p.do_refine) for (size_t j = 0; j <= planes_.size(); j++) sink_->offsets[j] = 0;
    }
    std::vector<char> validPlane_;
    std::vector<float> distMap_;
    int add_node(const Node &src) {
        if ((size_t)nNodes_ == nodes_.size()) nodes_.emplace_back();
        nodes_[nNodes_] = src;                // (src.nbs is empty: the slot's vector is cleared, not reallocated)
        return nNodes_++;
    }
    std::vector<int> parent_, setSize_;       // disjoint set over the windows (DisjointSet.hpp)
    std::vector<int> planes_;                 // extractedPlanes, node ids
    std::vector<int> blkMap_;
    std::vector<std::pair<int, int>> growQ_;  // rfQueue: (pixel, plane)
    int32_t *member_ = nullptr;

    int find(int x) { while (parent_[x] != x) { parent_[x] = parent_[parent_[x]]; x = parent_[x]; } return x; }   // (path halving: same roots as Find())
    void unite(int x, int y) {
        const int xr = find(x), yr = find(y);
        if (xr == yr) return;
        if (setSize_[xr] < setSize_[yr]) { parent_[xr] = yr; setSize_[yr] += setSize_[xr]; }
        else { parent_[yr] = xr; setSize_[xr] += setSize_[yr]; }
    }
    static double similarity(const Node &a, const Node &b) { return std::abs(a.normal[0] * b.normal[0] + a.normal[1] * b.normal[1] + a.normal[2] * b.normal[2]); }
    static void link_one(std::vector<int> &v, int id) { auto it = std::lower_bound(v.begin(), v.end(), id); if (it == v.end() || *it != id) v.insert(it, id); }
    static void unlink_one(std::vector<int> &v, int id) { auto it = std::lower_bound(v.begin(), v.end(), id); if (it != v.end() && *it == id) v.erase(it); }
    void connect(int a, int b) {
        link_one(nodes_[a].nbs, b); link_one(nodes_[b].nbs, a);
        if (recordEdges_) { edges_.push_back(a); edges_.push_back(b); }
    }
    void isolate(int a) { for (int nb : nodes_[a].nbs) unlink_one(nodes_[nb].nbs, a); nodes_[a].nbs.clear(); }

    void build_graph(const msl_peac_block *blocks) {
        std::vector<int> &G = G_;   // node id of an accepted window
        G.assign((size_t)Nw * Nh, -1);
        for (int b = 0; b < Nw * Nh; b++) {
            const msl_peac_block &B = blocks[b];
            Node nd;
            nd.st = B.stats; nd.mse = B.mse; nd.curvature = B.curvature; nd.rid = b; nd.nouse = B.stats.nouse != 0; nd.N = nd.nouse ? 0 : B.stats.N;
            for (int k = 0; k < 3; k++) { nd.center[k] = B.center[k]; nd.normal[k] = B.normal[k]; }
            add_node(nd);
            if (nd.mse < T.t_mse_init(nd.center[2]) && !nd.nouse) { G[b] = b; heap_push(b); }
        }
        // edges between horizontally / vertically adjacent accepted windows whose two outer neighbours agree in normal (:849-927)
        auto sweep = [&](int outerN, int innerN, int outerStride, int innerStride) {
            for (int o = 0; o < outerN; ++o)
                for (int k = 1; k < innerN; k += 2) {
                    const int c = o * outerStride + k * innerStride, prev = c - innerStride, next = c + innerStride;
                    if (G[prev] < 0) { --k; continue; }
                    if (G[c] < 0) continue;
                    if (k < innerN - 1 && G[next] < 0) { ++k; continue; }
                    const double th = T.t_ang_init(nodes_[G[c]].center[2]);
                    const bool ok = k < innerN - 1 ? similarity(nodes_[G[prev]], nodes_[G[next]]) >= th : similarity(nodes_[G[c]], nodes_[G[prev]]) >= th;
                    if (ok) { connect(G[c], G[prev]); if (k < innerN - 1) connect(G[c], G[next]); }
                    else --k;
                }
        };
        sweep(Nh, Nw, Nw, 1);
        sweep(Nw, Nh, 1, Nw);
    }

    double merged_mse(int a, int b) const {
        msl_peac_stats t;
        const msl_peac_stats &x = nodes_[a].st, &y = nodes_[b].st;
        t.sx = x.sx + y.sx; t.sy = x.sy + y.sy; t.sz = x.sz + y.sz; t.sxx = x.sxx + y.sxx; t.syy = x.syy + y.syy; t.szz = x.szz + y.szz;
        t.sxy = x.sxy + y.sxy; t.syz = x.syz + y.syz; t.sxz = x.sxz + y.sxz; t.N = x.N + y.N; t.nouse = 0;
        return plane_mse(t);
    }
    // candMse_[i] = merged_mse(p, cand_[i]), the candidates taken VW at a time (a short last group is padded with its first candidate)
    std::vector<int> cand_;
    std::vector<double> candMse_;
    void candidate_mses(int p) {
        const size_t n = cand_.size();
        candMse_.resize(n);
        const msl_peac_stats &x = nodes_[p].st;
        alignas(64) double in[10 * 16], out[16];
        for (size_t i0 = 0; i0 < n;) {
            const size_t left = n - i0;
            const int vw = lanes_for(left);
            if (vw == 0) { candMse_[i0] = merged_mse(p, cand_[i0]); ++i0; continue; }
            for (int l = 0; l < vw; l++) {
                const msl_peac_stats &y = nodes_[cand_[i0 + ((size_t)l < left ? l : 0)]].st;
                in[0 * vw + l] = x.sx + y.sx; in[1 * vw + l] = x.sy + y.sy; in[2 * vw + l] = x.sz + y.sz;
                in[3 * vw + l] = x.sxx + y.sxx; in[4 * vw + l] = x.syy + y.syy; in[5 * vw + l] = x.szz + y.szz;
                in[6 * vw + l] = x.sxy + y.sxy; in[7 * vw + l] = x.syz + y.syz; in[8 * vw + l] = x.sxz + y.sxz;
                in[9 * vw + l] = (double)(x.N + y.N);
            }
            plane_mse_group(vw, in, out);
            for (int l = 0; l < vw && (size_t)l < left; l++) candMse_[i0 + l] = out[l];
            i0 += vw;
        }
    }
    Node merged_node(int a, int b) const {   // PlaneSeg(pa, pb) (AHCPlaneSeg.hpp:299-322)
        Node nd;
        const msl_peac_stats &x = nodes_[a].st, &y = nodes_[b].st;
        nd.st.sx = x.sx + y.sx; nd.st.sy = x.sy + y.sy; nd.st.sz = x.sz + y.sz; nd.st.sxx = x.sxx + y.sxx; nd.st.syy = x.syy + y.syy; nd.st.szz = x.szz + y.szz;
        nd.st.sxy = x.sxy + y.sxy; nd.st.syz = x.syz + y.syz; nd.st.sxz = x.sxz + y.sxz; nd.st.N = x.N + y.N; nd.st.nouse = 0;
        nd.nouse = false;
        nd.rid = nodes_[a].N >= nodes_[b].N ? nodes_[a].rid : nodes_[b].rid;
        nd.N = nd.st.N;
        plane_fit(nd.st, nd.center, nd.normal, nd.mse, nd.curvature);
        return nd;
    }

    void cluster() {   // ahCluster (:939-1143) on heap_
        int step = 0;
        while (!heap_.empty() && step <= T.p.max_step) {
            const int p = heap_pop();
            if (nodes_[p].nouse) continue;
            // try to merge with every neighbour (ascending id), keep the merge with the smallest MSE
            // (only the MSE of every candidate is needed to choose; the full node -- centre, normal, curvature -- is built for the winner alone)
            bool have = false;
            double bestMse = 0;
            int bestN = 0, bestNb = -1;
            cand_.clear();
            for (int nb : nodes_[p].nbs)
                if (!(similarity(nodes_[p], nodes_[nb]) < T.p.similarity_th_merge)) cand_.push_back(nb);
            candidate_mses(p);
            for (size_t ci = 0; ci < cand_.size(); ci++) {
                const int nb = cand_[ci];
                const double mse = candMse_[ci];
                if (!have || bestMse > mse || (bestMse == mse && bestN < mse)) { bestMse = mse; bestN = nodes_[p].st.N + nodes_[nb].st.N; bestNb = nb; have = true; }   // (sic: N against mse, :1005)
            }
            Node best;
            if (have) best = merged_node(p, bestNb);
            if (have && best.mse < T.t_mse_merge(best.center[2])) {
                const int id = add_node(best);   // accepted merges get ascending ids: the newest node sorts last among neighbours
                heap_push(id);
                // mergeNbsFrom (AHCPlaneSeg.hpp:398-436)
                unite(nodes_[p].rid, nodes_[bestNb].rid);
                std::vector<int> &u = u_;
                u.clear();
                std::set_union(nodes_[p].nbs.begin(), nodes_[p].nbs.end(), nodes_[bestNb].nbs.begin(), nodes_[bestNb].nbs.end(), std::back_inserter(u));
                unlink_one(u, p); unlink_one(u, bestNb);
                isolate(p); isolate(bestNb);
                for (int nb : u) link_one(nodes_[nb].nbs, id);
                nodes_[id].nbs.assign(u.begin(), u.end());
                nodes_[p].nouse = nodes_[bestNb].nouse = true;
            } else {
                if (nodes_[p].N >= T.p.min_support) planes_.push_back(p);
                isolate(p);
            }
            ++step;
        }
        while (!heap_.empty()) {
            const int p = heap_pop();
            if (nodes_[p].N >= T.p.min_support) planes_.push_back(p);
            isolate(p);
        }
        std::sort(planes_.begin(), planes_.end(), [this](int a, int b) { return nodes_[b].N < nodes_[a].N; });   // PlaneSegSizeCmp
    }

    bool point(int row, int col, double pt[3]) const {   // ImagePointCloud::get on the fly, from the packed vertex depths
        const double z = (double)img_[(size_t)row * W + col] * factor_;
        pt[2] = z;
        if (z == 0) return false;
        vertex_xy(fx_, fy_, cx_, cy_, row, col, z, pt[0], pt[1]);
        return true;
    }
    static int neighbours4(int i, int j, int Hh, int Ww, int nbs[4]) {
        const int id = i * Ww + j;
        int cnt = 0;
        if (j > 0) nbs[cnt++] = id - 1;
        if (j < Ww - 1) nbs[cnt++] = id + 1;
        if (i > 0) nbs[cnt++] = id - Ww;
        if (i < Hh - 1) nbs[cnt++] = id + Ww;
        return cnt;
    }

    void erode_blocks(std::vector<char> &validPlane) {   // findBlockMembership(isValidExtractedPlane) (:490-596)
        std::map<int, int> rid2plid;
        for (int plid = 0; plid < (int)planes_.size(); ++plid) rid2plid.insert(std::make_pair(nodes_[planes_[plid]].rid, plid));
        const int perBlk = winW * winH;
        blkMap_.assign((size_t)Nw * Nh, -1);
        validPlane.assign(planes_.size(), 0);
        for (int i = 0, blk = 0; i < Nh; ++i)
            for (int j = 0; j < Nw; ++j, ++blk) {
                const int setid = find(blk);
                if (setSize_[setid] * perBlk >= T.p.min_support) {
                    int nb4[4] = {-1, -1, -1, -1};
                    const int nNb = neighbours4(i, j, Nh, Nw, nb4);
                    bool interior = true;
                    for (int k = 0; k < nNb && T.p.erode_type != 0; ++k)
                        if (find(nb4[k]) != setid && (T.p.erode_type == 2 || setSize_[find(nb4[k])] * perBlk >= T.p.min_support)) { interior = false; break; }
                    const int plid = rid2plid[setid];   // default-inserts plane 0 for a set whose root is no extracted plane's rid, as the reference does
                    if (interior) {
                        blkMap_[blk] = plid;
                        for (int y = i * winH; y < (i + 1) * winH; y++) std::fill(member_ + (size_t)y * W + j * winW, member_ + (size_t)y * W + (j + 1) * winW, plid);
                        validPlane[plid] = 1;
                    }
                }
                // seeds of the region growing: the pixels of a plane window that face a window of another (or no) plane
                if (blkMap_[blk] < 0) {
                    if (i > 0 && blkMap_[blk - Nw] >= 0)
                        for (int k = 1; k < winW; ++k) growQ_.push_back(std::make_pair((i * winH - 1) * W + j * winW + k, blkMap_[blk - Nw]));
                    if (j > 0 && blkMap_[blk - 1] >= 0)
                        for (int k = 0; k < winH - 1; ++k) growQ_.push_back(std::make_pair((i * winH) * W + j * winW - 1 + k * W, blkMap_[blk - 1]));
                } else {
                    const int plid = blkMap_[blk];
                    if (i > 0 && blkMap_[blk - Nw] != plid)
                        for (int k = 0; k < winW - 1; ++k) growQ_.push_back(std::make_pair((i * winH) * W + j * winW + k, plid));
                    if (j > 0 && blkMap_[blk - 1] != plid)
                        for (int k = 1; k < winH; ++k) growQ_.push_back(std::make_pair((i * winH) * W + j * winW + k * W, plid));
                }
            }
    }

    void grow_regions() {   // floodFill (:422-471)
        std::vector<float> &distMap = distMap_;
        distMap.assign((size_t)H * W, std::numeric_limits<float>::max());
        for (size_t k = 0; k < growQ_.size(); ++k) {
            const int seed = growQ_[k].first, plid = growQ_[k].second;
            const int sy = seed / W, sx = seed - sy * W;
            const Node &pl = nodes_[planes_[plid]];
            int nb4[4] = {-1, -1, -1, -1};
            const int nNb = neighbours4(sy, sx, H, W, nb4);
            for (int t = 0; t < nNb; ++t) {
                const int c = nb4[t];
                int32_t &trail = member_[c];
                if (trail <= -6) continue;
                if (trail >= 0 && trail == plid) continue;
                const int cy = c / W, cx = c - cy * W;
                const int by = cy / winH, bx = cx / winW;
                if (by < Nh && bx < Nw && blkMap_[by * Nw + bx] >= 0) continue;   // only pixels outside the plane windows
                double pt[3] = {0, 0, 0};
                float cdist = -1;
                bool close = false;
                if (point(cy, cx, pt)) {
                    cdist = (float)std::abs(pl.normal[0] * (pt[0] - pl.center[0]) + pl.normal[1] * (pt[1] - pl.center[1]) + pl.normal[2] * (pt[2] - pl.center[2]));
                    close = std::pow(cdist, 2) < 9 * pl.mse + 1e-5;   // point-plane distance within 3 sigma
                }
                if (close) {
                    if (trail >= 0 && similarity(pl, nodes_[planes_[trail]]) >= T.p.similarity_th_refine) connect(planes_[trail], planes_[plid]);
                    float &old = distMap[c];
                    if (cdist < old) { trail = plid; old = cdist; growQ_.push_back(std::make_pair(c, plid)); }
                    else if (trail < 0) trail -= 1;
                } else if (trail < 0) {
                    trail -= 1;
                }
            }
        }
    }

    void refine() {   // refineDetails (:296-372)
        std::vector<char> &validPlane = validPlane_;
        erode_blocks(validPlane);
        grow_regions();
        std::vector<int> &old = oldPlanes_;
        old.assign(planes_.begin(), planes_.end());
        planes_.clear();
        heap_.clear();
        for (size_t i = 0; i < old.size(); ++i)
            if (validPlane[i]) heap_push(old[i]);
        cluster();
        std::vector<int> &relabel = relabel_;
        relabel.assign(old.size(), -1);
        for (size_t i = 0; i < old.size(); ++i) {
            if (!validPlane[i]) continue;
            const int root = find(nodes_[old[i]].rid);
            for (size_t j = 0; j < planes_.size(); ++j)
                if (root == nodes_[planes_[j]].rid) { relabel[i] = (int)j; break; }
        }
        const bool lists = sink_ && sink_->offsets && sink_->indices && (int)planes_.size() <= sink_->maxPlanes;
        if (lists) {   // pMembership (:341-361): sizes first, so every plane's pixels land contiguously and in raster order
            std::vector<int> &cur = u_;
            cur.assign(planes_.size() + 1, 0);
            for (size_t i = 0, nPx = (size_t)W * H; i < nPx; ++i) {
                const int32_t plid = member_[i];
                if (plid >= 0 && relabel[plid] >= 0) cur[relabel[plid] + 1]++;
            }
            for (size_t j = 0; j < planes_.size(); j++) cur[j + 1] += cur[j];
            for (size_t j = 0; j <= planes_.size(); j++) sink_->offsets[j] = cur[j];
        }
        for (size_t i = 0, nPx = (size_t)W * H; i < nPx; ++i) {
            int32_t &plid = member_[i];
            if (plid >= 0 && relabel[plid] >= 0) {   // anything else keeps its value (old id or visit counter), as in the reference
                plid = relabel[plid];
                if (lists) sink_->indices[u_[plid]++] = (int32_t)i;
            }
        }
    }
};

// CPUs this process may actually use: hardware threads, limited by the affinity mask and by the cgroup CPU quota (cpu.max of cgroup v2 /
// cpu.cfs_quota_us of v1).  More runnable threads than that only burn the quota early in each period and are then throttled together
// (measured on a 256-thread host with a 16-CPU quota: 64 workers -> every third call stalled for 60-80 ms).
int usable_cpus() {
    int n = std::max(1, (int)std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = 0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(fq, "%lld", &quota) != 1) quota = -1;
        fclose(fq);
        if (FILE *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &period) != 1) period = 0; fclose(fp); }
    }
    if (quota > 0 && period > 0) n = std::min(n, (int)std::max(1ll, (quota + period - 1) / period));
    return n;
}

// Host workers for the per-frame clustering (frames are independent).  The WORKSPACES persist between calls (no allocation, no page faults in
// the steady state); the threads are started per call (freshly created threads are spread over idle cores at once; ~20 us each).  The caller
// takes part with its own workspace, so a one-frame call starts no thread at all.
class SegPool {
public:
    static SegPool &get() { static SegPool p; return p; }
    int workers() const { return maxWorkers_ + 1; }   // the caller takes part
    long long thread_shortfall() const { return threadShortfall_.load(); }   // worker threads that could not be started since the process began
    // fn(frame, workspace) for frame = 0 .. nFrames-1, each exactly once; returns when all are done
    void run(int nFrames, const std::function<void(int, FrameSegmenter &)> &fn) {
        std::lock_guard<std::mutex> one(callMutex_);   // one batch at a time
        if (nFrames <= 0) return;
        const int nWorkers = std::min(nFrames - 1, maxWorkers_);
        while ((int)ws_.size() < nWorkers + 1) ws_.emplace_back(new FrameSegmenter);
        std::atomic<int> next{0};
        // An exception inside a worker (std::bad_alloc in a workspace, ...) must not terminate the process: the first one is kept, the other
        // frames are abandoned, every started thread is joined, and the caller's thread rethrows it -- into the C ABI's exception barrier.
        std::exception_ptr failure;
        std::mutex failMutex;
        auto work = [&](FrameSegmenter &ws) {
            try {
                for (;;) {
                    const int f = next.fetch_add(1);
                    if (f >= nFrames) break;
                    fn(f, ws);
                }
            } catch (...) {
                std::lock_guard<std::mutex> g(failMutex);
                if (!failure) failure = std::current_exception();
                next.store(nFrames);
            }
        };
        std::vector<std::thread> threads;
        threads.reserve(nWorkers);
        try {
            for (int t = 0; t < nWorkers; t++) threads.emplace_back([&, t]() { work(*ws_[t + 1]); });
        } catch (...) {   // thread creation failed (std::system_error): the threads that did start finish the work together with the caller
            std::lock_guard<std::mutex> g(failMutex);
            static const bool strict = getenv("MSL_PEAC_STRICT_THREADS") != nullptr;
            if (strict && !failure) failure = std::current_exception();
            // the degradation is recorded, not silent: a counter the debug hook reads, and -- once per process -- a line in msl_last_error()'s
            // buffer (the call still succeeds) and on stderr
            const int miss = nWorkers - (int)threads.size();
            if (threadShortfall_.fetch_add(miss) == 0) {
                set_error("msl_peac: only %d of %d worker threads could be started (resource limit?); the call continues with fewer", (int)threads.size(), nWorkers);
                fprintf(stderr, "[msl_peac] warning: only %d of %d worker threads could be started; continuing with fewer\n", (int)threads.size(), nWorkers);
            }
        }
        work(*ws_[0]);
        for (auto &th : threads) th.join();
        if (failure) std::rethrow_exception(failure);
    }

private:
    // MSL_PEAC_THREADS overrides the worker count (1 = everything on the calling thread)
    // and one process per GPU shares the node's CPUs with its sibling ranks: LOCAL_WORLD_SIZE (set by torch.distributed.run) divides the budget,
    // so 8 ranks do not start 8 x usable_cpus() workers
    static int worker_budget() {
        if (const char *t = getenv("MSL_PEAC_THREADS")) return atoi(t);
        const char *lws = getenv("LOCAL_WORLD_SIZE");
        const int ranks = lws ? std::max(1, atoi(lws)) : 1;
        return std::max(1, usable_cpus() / ranks);
    }
    SegPool() : maxWorkers_(std::max(0, std::min(64, worker_budget()) - 1)) {
        if (getenv("MSL_PEAC_POOL_REPORT")) fprintf(stderr, "[msl_peac] pool workers = %d (usable CPUs %d)\n", maxWorkers_ + 1, usable_cpus());
    }
    const int maxWorkers_;
    std::atomic<long long> threadShortfall_{0};
    std::mutex callMutex_;
    std::vector<std::unique_ptr<FrameSegmenter>> ws_;
};

// graph initialisation + clustering + erosion + region growing of n_frames frames: blocks [frames][nBlocks], half [frames][ch][cw]
void segment_frames(const msl_peac_params &prm, const msl_peac_block *blocks, size_t nBlocks, const uint16_t *half, int cw, int ch, int n_frames, float fx,
                    float fy, float cx, float cy, float depth_map_factor, int32_t *membership_out, int32_t *n_planes_out, PlaneSink *sinks = nullptr) {
    const bool timing = getenv("MSL_PEAC_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<long> startUs(timing ? n_frames : 0), durUs(timing ? n_frames : 0);
    SegPool::get().run(n_frames, [&](int f, FrameSegmenter &seg) {
        const auto a = std::chrono::steady_clock::now();
        seg.configure(prm, half + (size_t)f * cw * ch, cw, ch, fx, fy, cx, cy, depth_map_factor);
        seg.set_sink(sinks ? &sinks[f] : nullptr);
        const int n = seg.run(blocks + (size_t)f * nBlocks, membership_out + (size_t)f * cw * ch);
        seg.set_sink(nullptr);
        if (n_planes_out) n_planes_out[f] = n;
        if (timing) {
            startUs[f] = (long)std::chrono::duration_cast<std::chrono::microseconds>(a - t0).count();
            durUs[f] = (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - a).count();
        }
    });
    if (timing && n_frames > 1) {
        long ms = 0, md = 0, sd = 0;
        for (int f = 0; f < n_frames; f++) { ms = std::max(ms, startUs[f]); md = std::max(md, durUs[f]); sd += durUs[f]; }
        fprintf(stderr, "[msl_peac] pool: latest frame start %ld us, longest frame %ld us, mean frame %ld us\n", ms, md, sd / n_frames);
    }
}

struct Scratch {
    void *depth = nullptr, *blocks = nullptr, *cloud = nullptr, *half = nullptr; size_t depthCap = 0, blocksCap = 0, cloudCap = 0, halfCap = 0;
    // The extractor's own stream (non-blocking, highest priority): its few small kernels and copies must neither wait for nor hold up the frame-batched
    // ORB / surfel work queued on the device.  Work the caller enqueued on the legacy default stream before the call is still ordered first (event).
    hipStream_t stream = nullptr; hipEvent_t ev = nullptr;
    // device clustering (k_peac_cluster)
    void *rows = nullptr, *gst = nullptr, *gcxy = nullptr, *cin = nullptr, *cout = nullptr; size_t rowsCap = 0, gstCap = 0, gcxyCap = 0, cinCap = 0, coutCap = 0;
    bool clusterLdsSet = false;
};
Scratch g_scratch[16];
std::mutex g_scratchMutex;

hipError_t grow(void *&p, size_t &cap, size_t need) {
    if (need <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr; cap = 0;
    const hipError_t e = hipMalloc(&p, need);
    if (e == hipSuccess) cap = need;
    return e;
}

#define PEAC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("msl_peac: %s", hipGetErrorString(e_)); return MSL_ERR_HIP; } } while (0)

// cloud (optional) + block fit for n_frames images; dBlocksOut receives the device pointer of the [frames][Nh * Nw] blocks.
// The caller holds g_scratchMutex.
int device_fit(int device, const uint16_t *depth, size_t strideBytes, size_t frameStrideBytes, int width, int height, int n_frames, msl_mem mem, float fx,
               float fy, float cx, float cy, float factor, const msl_peac_params &prm, double *cloudDev, msl_peac_block **dBlocksOut, uint16_t **dHalfOut /* nullptr: not wanted */,
               msl_peac_block *blocksUser /* device output buffer or nullptr */) {
    // k_peac_fit stages 72 bytes per window point in dynamic LDS: up to 900 points (e.g. 30 x 30) fit the 64 KB a launch may ask for
    if (!depth || width < 2 || height < 2 || n_frames < 1 || prm.window_w < 1 || prm.window_h < 1 || prm.window_w * prm.window_h > 900 ||
        strideBytes < (size_t)width * 2 || (n_frames > 1 && frameStrideBytes < strideBytes * (size_t)(height - 1) + (size_t)width * 2) || fx == 0 || fy == 0) {
        set_error("msl_peac: invalid argument");
        return MSL_ERR_INVALID;
    }
    int rc = bind_device(device);
    if (rc != MSL_OK) return rc;
    PeacDev P;
    P.width = width; P.height = height; P.cw = (width + 1) / 2; P.ch = (height + 1) / 2;   // ceil(cols / 2.0), ceil(rows / 2.0) (:51-52)
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.factor = factor;
    P.winW = prm.window_w; P.winH = prm.window_h; P.Nw = P.cw / prm.window_w; P.Nh = P.ch / prm.window_h; P.loose = prm.init_loose ? 1 : 0;
    P.alpha = prm.depth_alpha; P.tol = prm.depth_change_tol;
    P.strideBytes = strideBytes; P.frameStrideBytes = frameStrideBytes;
    const size_t nBlocks = (size_t)P.Nw * P.Nh, nVert = (size_t)P.cw * P.ch;
    if (nBlocks == 0) { set_error("msl_peac: image smaller than one window"); return MSL_ERR_INVALID; }
    Scratch &sc = g_scratch[device & 15];
    if (!sc.stream) {
        int lo = 0, hi = 0;
        PEAC_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
        PEAC_TRY(hipStreamCreateWithPriority(&sc.stream, hipStreamNonBlocking, hi));
        PEAC_TRY(hipEventCreateWithFlags(&sc.ev, hipEventDisableTiming));
    }
    const hipStream_t st = sc.stream;
    if (mem == MSL_MEM_DEVICE) { PEAC_TRY(hipEventRecord(sc.ev, 0)); PEAC_TRY(hipStreamWaitEvent(st, sc.ev, 0)); }
    if (mem == MSL_MEM_HOST) {
        // bytes actually present in the caller's buffer: the last row carries no stride padding
        const size_t frameBytes = strideBytes * (size_t)(height - 1) + (size_t)width * 2, slot = (frameBytes + 255) & ~(size_t)255;
        PEAC_TRY(grow(sc.depth, sc.depthCap, slot * n_frames));
        for (int f = 0; f < n_frames; f++)
            PEAC_TRY(hipMemcpyAsync((uint8_t *)sc.depth + f * slot, (const uint8_t *)depth + f * frameStrideBytes, frameBytes, hipMemcpyHostToDevice, st));
        P.depth = (const uint16_t *)sc.depth; P.frameStrideBytes = slot;
    } else {
        P.depth = depth;
    }
    if (blocksUser) P.blocks = blocksUser;
    else { PEAC_TRY(grow(sc.blocks, sc.blocksCap, sizeof(msl_peac_block) * nBlocks * n_frames)); P.blocks = (msl_peac_block *)sc.blocks; }
    P.cloud = cloudDev;
    if (cloudDev) hipLaunchKernelGGL(k_peac_cloud, dim3((unsigned)((nVert + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, P);
    hipLaunchKernelGGL(k_peac_fit, dim3((unsigned)nBlocks, (unsigned)n_frames), dim3(64), sizeof(double) * 9 * prm.window_w * prm.window_h, st, P);
    PEAC_TRY(hipGetLastError());
    if (dHalfOut) {
        PEAC_TRY(grow(sc.half, sc.halfCap, sizeof(uint16_t) * nVert * n_frames));
        hipLaunchKernelGGL(k_peac_half, dim3((unsigned)((nVert + 255) / 256), (unsigned)n_frames), dim3(256), 0, st, P, (uint16_t *)sc.half);
        PEAC_TRY(hipGetLastError());
        *dHalfOut = (uint16_t *)sc.half;
    }
    *dBlocksOut = P.blocks;
    return MSL_OK;
}

}  // namespace

extern "C" {

void msl_peac_default_params(msl_peac_params *p) noexcept {
    try {   // ahc::ParamSet / ahc::PlaneFitter defaults (AHCParamSet.hpp:68-76, AHCPlaneFitter.hpp:157-161)
    if (!p) return;
    p->window_w = 10; p->window_h = 10; p->min_support = 3000; p->max_step = 100000; p->do_refine = 1; p->erode_type = 2; p->init_loose = 0; p->_pad = 0;
    p->depth_sigma = 1.6e-6; p->std_tol_init = 5; p->std_tol_merge = 8; p->z_near = 500; p->z_far = 4000;
    p->angle_near = ((15.0) * M_PI / 180.0); p->angle_far = ((90.0) * M_PI / 180.0);
    p->similarity_th_merge = std::cos(((60.0) * M_PI / 180.0)); p->similarity_th_refine = std::cos(((30.0) * M_PI / 180.0));
    p->depth_alpha = 0.04; p->depth_change_tol = 0.02;
    } MSL_ABI_CATCH_VOID
}

int msl_peac_block_fit(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height, int n_frames,
                       msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, const msl_peac_params *params,
                       msl_peac_block *blocks_out, msl_mem out_mem) noexcept {
    try {
    if (!params || !blocks_out) { set_error("msl_peac_block_fit: invalid argument"); return MSL_ERR_INVALID; }
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    msl_peac_block *dBlocks = nullptr;
    int rc = device_fit(device, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, mem, fx, fy, cx, cy, depth_map_factor, *params, nullptr,
                        &dBlocks, nullptr, out_mem == MSL_MEM_DEVICE ? blocks_out : nullptr);
    if (rc != MSL_OK) return rc;
    const size_t nBlocks = (size_t)(((width + 1) / 2) / params->window_w) * (((height + 1) / 2) / params->window_h);
    const hipStream_t st = g_scratch[device & 15].stream;
    if (out_mem == MSL_MEM_HOST) PEAC_TRY(hipMemcpyAsync(blocks_out, dBlocks, sizeof(msl_peac_block) * nBlocks * n_frames, hipMemcpyDeviceToHost, st));
    PEAC_TRY(hipStreamSynchronize(st));
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_peac_block_stats(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height, int n_frames,
                         msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, int window_w, int window_h, double depth_alpha,
                         double depth_change_tol, int init_loose, double *cloud_out, msl_peac_stats *stats_out, msl_mem out_mem) noexcept {
    try {
    if (!stats_out) { set_error("msl_peac_block_stats: invalid argument"); return MSL_ERR_INVALID; }
    msl_peac_params prm;
    msl_peac_default_params(&prm);
    prm.window_w = window_w; prm.window_h = window_h; prm.depth_alpha = depth_alpha; prm.depth_change_tol = depth_change_tol; prm.init_loose = init_loose;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    const size_t cw = (width + 1) / 2, ch = (height + 1) / 2, nVert = cw * ch;
    if (window_w < 1 || window_h < 1) { set_error("msl_peac_block_stats: invalid argument"); return MSL_ERR_INVALID; }
    const size_t nBlocks = (cw / window_w) * (ch / window_h);
    double *dCloud = nullptr;
    if (cloud_out) {
        if (out_mem == MSL_MEM_HOST) {
            if (bind_device(device) != MSL_OK) return MSL_ERR_NO_DEVICE;
            Scratch &sc = g_scratch[device & 15];
            PEAC_TRY(grow(sc.cloud, sc.cloudCap, sizeof(double) * 3 * nVert * n_frames));
            dCloud = (double *)sc.cloud;
        } else {
            dCloud = cloud_out;
        }
    }
    msl_peac_block *dBlocks = nullptr;
    int rc = device_fit(device, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, mem, fx, fy, cx, cy, depth_map_factor, prm, dCloud, &dBlocks,
                        nullptr, nullptr);
    if (rc != MSL_OK) return rc;
    // this entry point returns the Stats part only
    std::vector<msl_peac_block> hb(nBlocks * n_frames);
    const hipStream_t st = g_scratch[device & 15].stream;
    PEAC_TRY(hipMemcpyAsync(hb.data(), dBlocks, sizeof(msl_peac_block) * hb.size(), hipMemcpyDeviceToHost, st));
    PEAC_TRY(hipStreamSynchronize(st));
    std::vector<msl_peac_stats> hs(hb.size());
    for (size_t i = 0; i < hb.size(); i++) hs[i] = hb[i].stats;
    if (out_mem == MSL_MEM_HOST) {
        memcpy(stats_out, hs.data(), sizeof(msl_peac_stats) * hs.size());
        if (cloud_out) { PEAC_TRY(hipMemcpyAsync(cloud_out, dCloud, sizeof(double) * 3 * nVert * n_frames, hipMemcpyDeviceToHost, st)); PEAC_TRY(hipStreamSynchronize(st)); }
    } else {
        PEAC_TRY(hipMemcpyAsync(stats_out, hs.data(), sizeof(msl_peac_stats) * hs.size(), hipMemcpyHostToDevice, st));
        PEAC_TRY(hipStreamSynchronize(st));
    }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

}  // extern "C"

namespace {
// per-frame sinks over the caller's arrays (nullptr when no plane output is wanted)
std::vector<PlaneSink> make_sinks(int n_frames, int cw, int ch, int max_planes, msl_peac_plane *planes_out, int32_t *vertex_offsets_out, int32_t *vertex_indices_out) {
    std::vector<PlaneSink> sinks;
    if (!planes_out) return sinks;
    sinks.resize(n_frames);
    for (int f = 0; f < n_frames; f++) {
        sinks[f].planes = planes_out + (size_t)f * max_planes;
        sinks[f].offsets = vertex_offsets_out ? vertex_offsets_out + (size_t)f * (max_planes + 1) : nullptr;
        sinks[f].indices = vertex_indices_out ? vertex_indices_out + (size_t)f * cw * ch : nullptr;
        sinks[f].maxPlanes = max_planes; sinks[f].overflow = false;
    }
    return sinks;
}
int check_sinks(const std::vector<PlaneSink> &sinks, int max_planes) {
    for (const PlaneSink &k : sinks)
        if (k.overflow) { set_error("msl_peac: a frame has more than max_planes = %d planes", max_planes); return MSL_ERR_CAPACITY; }
    return MSL_OK;
}

int membership_impl(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height, int n_frames,
                    msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, const msl_peac_params *params,
                    int32_t *membership_out, int32_t *n_planes_out, int max_planes, msl_peac_plane *planes_out, int32_t *vertex_offsets_out,
                    int32_t *vertex_indices_out, double *cloud_out = nullptr) {
    if (!params || !membership_out || params->min_support < 1 || (planes_out && max_planes < 1) || ((vertex_offsets_out || vertex_indices_out) && !planes_out) ||
        ((vertex_offsets_out != nullptr) != (vertex_indices_out != nullptr)) || (vertex_indices_out && !params->do_refine)) {
        set_error("msl_peac: invalid argument (plane outputs need max_planes >= 1; vertex lists need planes_out, both list arrays and do_refine)");
        return MSL_ERR_INVALID;
    }
    std::vector<PlaneSink> sinks = make_sinks(n_frames, (width + 1) / 2, (height + 1) / 2, max_planes, planes_out, vertex_offsets_out, vertex_indices_out);
    PlaneSink *sinkp = sinks.empty() ? nullptr : sinks.data();
    std::vector<msl_peac_block> hb;
    std::vector<uint16_t> half;       // raw depth of the cloud vertices, [frames][ch][cw]
    std::vector<int> hOutI;           // device clustering result: plane count per frame, disjoint-set parents and sizes
    std::vector<PlaneOut> hPlanes;
    bool usedDevice = false;
    int maxPl = 0;
    const int cw = (width + 1) / 2, ch = (height + 1) / 2;
    size_t nBlocks = 0;
    const bool timing = getenv("MSL_PEAC_TIMING") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    {
        std::lock_guard<std::mutex> lock(g_scratchMutex);
        msl_peac_block *dBlocks = nullptr;
        uint16_t *dHalf = nullptr;
        double *dCloud = nullptr;   // the organised cloud of PlaneDetection::readDepthImage, when the caller wants it (k_peac_cloud)
        if (cloud_out) {
            Scratch &sc0 = g_scratch[device & 15];
            PEAC_TRY(hipSetDevice(device));
            PEAC_TRY(grow(sc0.cloud, sc0.cloudCap, sizeof(double) * 3 * (size_t)cw * ch * n_frames));
            dCloud = (double *)sc0.cloud;
        }
        int rc = device_fit(device, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, mem, fx, fy, cx, cy, depth_map_factor, *params, dCloud,
                            &dBlocks, &dHalf, nullptr);
        if (rc != MSL_OK) return rc;
        nBlocks = (size_t)(cw / params->window_w) * (ch / params->window_h);
        hb.resize(nBlocks * n_frames);
        half.resize((size_t)cw * ch * n_frames);
        const hipStream_t st = g_scratch[device & 15].stream;
        PEAC_TRY(hipMemcpyAsync(hb.data(), dBlocks, sizeof(msl_peac_block) * hb.size(), hipMemcpyDeviceToHost, st));
        PEAC_TRY(hipMemcpyAsync(half.data(), dHalf, sizeof(uint16_t) * half.size(), hipMemcpyDeviceToHost, st));
        if (cloud_out) PEAC_TRY(hipMemcpyAsync(cloud_out, dCloud, sizeof(double) * 3 * (size_t)cw * ch * n_frames, hipMemcpyDeviceToHost, st));
        PEAC_TRY(hipStreamSynchronize(st));
        // Agglomerative clustering on the device (one wave per frame) when a frame's node data fits the LDS and the call is large enough;
        // MSL_PEAC_CLUSTER=host / device forces one side (same results: tests/test_peac_gpu.py runs both).
        const int maxN = 2 * (int)nBlocks, words = (maxN + 31) / 32;
        maxPl = (int)std::min<size_t>(nBlocks, 256);
        const size_t ldsBytes = (size_t)maxN * (6 * sizeof(double) + 4 * sizeof(int) + 1) + 2 * nBlocks * sizeof(int) + 64;
        // auto: the device clusters any number of frames in the time of one (~13-20 ms, one latency-bound wave per frame), a host worker needs ~2 ms
        // per frame (candidate merges evaluated 16 at a time, plane_mse_lanes): the device wins once a call holds more than about eight frames per
        // usable CPU (measured: 64 frames on 16 workers, host 25.5 k frames/s of configuration 4 against 22.9 k with the device clustering).
        const bool wantDevice = msl_debug_peac_cluster_on_device(n_frames) != 0;
        if (ldsBytes <= 150 * 1024 && wantDevice) {
            Scratch &sc = g_scratch[device & 15];
            const int maxE = 4 * (int)nBlocks;
            // graph initialisation on the host workers -> initial heap + edge list per frame
            const size_t inInts = (size_t)n_frames * (nBlocks + 1 + 2 * (size_t)maxE + 1);
            std::vector<int> hIn(inInts);
            int *hHeap = hIn.data(), *hHeapN = hHeap + (size_t)n_frames * nBlocks, *hEdges = hHeapN + n_frames, *hEdgeN = hEdges + (size_t)n_frames * maxE * 2;
            std::atomic<int> bad{0};
            SegPool::get().run(n_frames, [&](int f, FrameSegmenter &seg) {
                seg.configure(*params, half.data() + (size_t)f * cw * ch, cw, ch, fx, fy, cx, cy, depth_map_factor);
                if (!seg.graph_for_device(hb.data() + (size_t)f * nBlocks, hHeap + (size_t)f * nBlocks, hHeapN + f, hEdges + (size_t)f * maxE * 2, hEdgeN + f, maxE)) bad++;
            });
            if (!bad.load()) {
                const size_t rowsB = sizeof(unsigned) * (size_t)n_frames * maxN * words, gstB = sizeof(double) * 9 * (size_t)n_frames * maxN,
                             gcxyB = sizeof(double) * 2 * (size_t)n_frames * maxN;
                const size_t outInts = (size_t)n_frames * (1 + 2 * nBlocks), outB = sizeof(int) * ((outInts + 1) & ~(size_t)1) + sizeof(PlaneOut) * (size_t)n_frames * maxPl;
                PEAC_TRY(grow(sc.rows, sc.rowsCap, rowsB)); PEAC_TRY(grow(sc.gst, sc.gstCap, gstB)); PEAC_TRY(grow(sc.gcxy, sc.gcxyCap, gcxyB));
                PEAC_TRY(grow(sc.cin, sc.cinCap, sizeof(int) * inInts)); PEAC_TRY(grow(sc.cout, sc.coutCap, outB));
                PEAC_TRY(hipMemcpyAsync(sc.cin, hIn.data(), sizeof(int) * inInts, hipMemcpyHostToDevice, st));
                PEAC_TRY(hipMemsetAsync(sc.rows, 0, rowsB, st));
                ClusterDev C;
                C.nB = (int)nBlocks; C.maxN = maxN; C.words = words; C.minSupport = params->min_support; C.maxStep = params->max_step; C.maxE = maxE; C.maxPl = maxPl;
                C.depthSigma = params->depth_sigma; C.stdTolMerge = params->std_tol_merge; C.simMerge = params->similarity_th_merge;
                C.blocks = dBlocks; C.rows = (unsigned *)sc.rows; C.gst = (double *)sc.gst; C.gcxy = (double *)sc.gcxy;
                int *dIn = (int *)sc.cin;
                C.heap0 = dIn; C.heapCount = dIn + (size_t)n_frames * nBlocks; C.edges = dIn + (size_t)n_frames * (nBlocks + 1);
                C.edgeCount = dIn + (size_t)n_frames * (nBlocks + 1 + 2 * (size_t)maxE);
                int *dOut = (int *)sc.cout;
                C.nPlanes = dOut; C.parent = dOut + n_frames; C.setSize = dOut + n_frames + (size_t)n_frames * nBlocks;
                C.planes = reinterpret_cast<PlaneOut *>(dOut + ((outInts + 1) & ~(size_t)1));
                if (!sc.clusterLdsSet) {
                    PEAC_TRY(hipFuncSetAttribute((const void *)k_peac_cluster, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                    sc.clusterLdsSet = true;
                }
                hipLaunchKernelGGL(k_peac_cluster, dim3((unsigned)n_frames), dim3(64), ldsBytes, st, C);
                PEAC_TRY(hipGetLastError());
                hOutI.resize(outInts); hPlanes.resize((size_t)n_frames * maxPl);
                PEAC_TRY(hipMemcpyAsync(hOutI.data(), dOut, sizeof(int) * outInts, hipMemcpyDeviceToHost, st));
                PEAC_TRY(hipMemcpyAsync(hPlanes.data(), C.planes, sizeof(PlaneOut) * hPlanes.size(), hipMemcpyDeviceToHost, st));
                PEAC_TRY(hipStreamSynchronize(st));
                usedDevice = true;
                for (int f = 0; f < n_frames; f++) if (hOutI[f] < 0) usedDevice = false;   // more planes than the hand-over holds: host path for this call
            }
        }
    }
    const auto tb1 = std::chrono::steady_clock::now();
    if (!usedDevice) segment_frames(*params, hb.data(), nBlocks, half.data(), cw, ch, n_frames, fx, fy, cx, cy, depth_map_factor, membership_out, n_planes_out, sinkp);
    else
        SegPool::get().run(n_frames, [&](int f, FrameSegmenter &seg) {
            seg.configure(*params, half.data() + (size_t)f * cw * ch, cw, ch, fx, fy, cx, cy, depth_map_factor);
            seg.set_sink(sinkp ? &sinkp[f] : nullptr);
            const int n = seg.finish_from_device(hPlanes.data() + (size_t)f * maxPl, hOutI[f], hOutI.data() + n_frames + (size_t)f * nBlocks,
                                                 hOutI.data() + n_frames + (size_t)(n_frames + f) * nBlocks, membership_out + (size_t)f * cw * ch);
            seg.set_sink(nullptr);
            if (n_planes_out) n_planes_out[f] = n;
        });
    if (timing) {
        const auto tb2 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        fprintf(stderr, "[msl_peac] batch of %d: device fit%s + copies %ld us, host stage %ld us\n", n_frames, usedDevice ? " + device clustering" : "", us(tb0, tb1),
                us(tb1, tb2));
    }
    return check_sinks(sinks, max_planes);
}
}  // namespace

extern "C" {

int msl_peac_membership_batch(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height, int n_frames,
                              msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, const msl_peac_params *params,
                              int32_t *membership_out, int32_t *n_planes_out) noexcept {
    try {
    return membership_impl(device, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, mem, fx, fy, cx, cy, depth_map_factor, params,
                           membership_out, n_planes_out, 0, nullptr, nullptr, nullptr);
    } MSL_ABI_CATCH_INT
}
int msl_peac_extract_batch(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height, int n_frames,
                           msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, const msl_peac_params *params,
                           int32_t *membership_out, int32_t *n_planes_out, int max_planes, msl_peac_plane *planes_out, int32_t *vertex_offsets_out,
                           int32_t *vertex_indices_out, double *cloud_out) noexcept {
    try {
    if (!planes_out) { set_error("msl_peac_extract_batch: planes_out is NULL (use msl_peac_membership_batch for the image alone)"); return MSL_ERR_INVALID; }
    return membership_impl(device, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, mem, fx, fy, cx, cy, depth_map_factor, params,
                           membership_out, n_planes_out, max_planes, planes_out, vertex_offsets_out, vertex_indices_out, cloud_out);
    } MSL_ABI_CATCH_INT
}

int msl_peac_extract_from_blocks(const msl_peac_block *blocks, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height,
                                 int n_frames, float fx, float fy, float cx, float cy, float depth_map_factor, const msl_peac_params *params,
                                 int32_t *membership_out, int32_t *n_planes_out, int max_planes, msl_peac_plane *planes_out, int32_t *vertex_offsets_out,
                                 int32_t *vertex_indices_out) noexcept {
    try {
    if (!blocks || !depth || !params || !membership_out || params->min_support < 1 || params->window_w < 1 || params->window_h < 1 || width < 2 || height < 2 ||
        n_frames < 0 || depth_stride_bytes < (size_t)width * 2 || (planes_out && max_planes < 1) || ((vertex_offsets_out || vertex_indices_out) && !planes_out) ||
        ((vertex_offsets_out != nullptr) != (vertex_indices_out != nullptr)) || (vertex_indices_out && !params->do_refine)) {
        set_error("msl_peac_extract_from_blocks: invalid argument");
        return MSL_ERR_INVALID;
    }
    const int cw = (width + 1) / 2, ch = (height + 1) / 2;
    const size_t nBlocks = (size_t)(cw / params->window_w) * (ch / params->window_h);
    std::vector<uint16_t> half((size_t)cw * ch * n_frames);   // raw depth of the cloud vertices (even rows / columns), as k_peac_half packs it
    for (int f = 0; f < n_frames; f++)
        for (int r = 0; r < ch; r++) {
            const uint16_t *row = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(depth) + (size_t)f * frame_stride_bytes + (size_t)(2 * r) * depth_stride_bytes);
            uint16_t *o = half.data() + ((size_t)f * ch + r) * cw;
            for (int c = 0; c < cw; c++) o[c] = row[2 * c];
        }
    std::vector<PlaneSink> sinks = make_sinks(n_frames, cw, ch, max_planes, planes_out, vertex_offsets_out, vertex_indices_out);
    segment_frames(*params, blocks, nBlocks, half.data(), cw, ch, n_frames, fx, fy, cx, cy, depth_map_factor, membership_out, n_planes_out,
                   sinks.empty() ? nullptr : sinks.data());
    return check_sinks(sinks, max_planes);
    } MSL_ABI_CATCH_INT
}
int msl_peac_membership_from_blocks(const msl_peac_block *blocks, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width,
                                    int height, int n_frames, float fx, float fy, float cx, float cy, float depth_map_factor,
                                    const msl_peac_params *params, int32_t *membership_out, int32_t *n_planes_out) noexcept {
    try {
    return msl_peac_extract_from_blocks(blocks, depth, depth_stride_bytes, frame_stride_bytes, width, height, n_frames, fx, fy, cx, cy, depth_map_factor, params,
                                        membership_out, n_planes_out, 0, nullptr, nullptr, nullptr);
    } MSL_ABI_CATCH_INT
}

// Where a call of n_frames keyframes clusters: 1 = on the device (one wave per frame, ~13-20 ms per call whatever the number of frames), 0 = on
// the host workers (~2 ms per frame and worker).  The device wins once a call holds more than about eight frames per worker this process may
// use -- and the worker count is the CPU budget divided by LOCAL_WORLD_SIZE, so the 8 ranks of a node (2 workers each on a 16-CPU allowance)
// take the device path for config 4's 128-keyframe calls instead of collapsing onto shared host cores.  MSL_PEAC_CLUSTER=host / device forces one side.
int msl_debug_peac_cluster_on_device(int n_frames) noexcept {
    try {
    const char *mode = getenv("MSL_PEAC_CLUSTER");
    if (mode && !strcmp(mode, "host")) return 0;
    if (mode && !strcmp(mode, "device")) return 1;
    return n_frames > 8 * SegPool::get().workers() ? 1 : 0;
    } MSL_ABI_CATCH_INT
}

long long msl_debug_peac_thread_shortfall(void) noexcept { try { return SegPool::get().thread_shortfall(); } MSL_ABI_CATCH_(return -1) }

int msl_debug_peac_mse(const msl_peac_stats *stats, size_t n, int lanes, double *mse_out) noexcept {
    try {
    if (n == 0) return MSL_OK;
    const int level = host_simd_level();
    if (!stats || !mse_out || !(lanes == 0 || lanes == 2 || lanes == 4 || lanes == 8 || lanes == 16)) { set_error("msl_debug_peac_mse: invalid argument"); return MSL_ERR_INVALID; }
    if ((lanes == 16 && level < 8) || (lanes >= 4 && level < 4) || (lanes >= 2 && level < 2)) { set_error("msl_debug_peac_mse: %d lanes need a wider instruction set than this CPU (or MSL_PEAC_SIMD) allows", lanes); return MSL_ERR_INVALID; }
    if (lanes == 0) { for (size_t i = 0; i < n; i++) mse_out[i] = plane_mse(stats[i]); return MSL_OK; }
    alignas(64) double in[10 * 16], out[16];
    for (size_t i0 = 0; i0 < n; i0 += lanes) {
        for (int l = 0; l < lanes; l++) {
            const msl_peac_stats &y = stats[i0 + l < n ? i0 + l : i0];
            const double v[10] = {y.sx, y.sy, y.sz, y.sxx, y.syy, y.szz, y.sxy, y.syz, y.sxz, (double)y.N};
            for (int k = 0; k < 10; k++) in[k * lanes + l] = v[k];
        }
        plane_mse_group(lanes, in, out);
        for (int l = 0; l < lanes && i0 + l < n; l++) mse_out[i0 + l] = out[l];
    }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}


// Test hook of the exception barrier (tests/test_abi.py): raises the given failure INSIDE the library, behind the boundary.
//   0: std::bad_alloc  1: std::runtime_error  2: a non-standard exception  3: std::bad_alloc in a worker thread of the plane extractor's pool
int msl_debug_throw(int kind) noexcept {
    try {
        if (kind == 0) throw std::bad_alloc();
        if (kind == 1) throw std::runtime_error("msl_debug_throw");
        if (kind == 2) throw 42;
        if (kind == 3) SegPool::get().run(4, [](int f, FrameSegmenter &) { if (f == 2) throw std::bad_alloc(); });
        return MSL_OK;
    } MSL_ABI_CATCH_INT
}

}  // extern "C"
