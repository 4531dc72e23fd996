// organised point cloud + initial block statistics of the PEAC plane extractor on the GPU.
// Everything is FP64 and reproduces the reference's evaluation order, so the outputs are bit-identical to the host code:
//   cloud  : PlaneDetection::readDepthImage           (/root/reference/src/PlaneExtractor.cpp:44-76)
//   blocks : ahc::PlaneSeg::PlaneSeg(points, ...)     (/root/reference/include/peac/AHCPlaneSeg.hpp:237-285), one block per thread
#include "msl_common.h"

#include <mutex>

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
    msl_peac_stats *stats;                 // [frames][Nh * Nw]
};

// z of cloud vertex (row, col): (double)depth(2 row, 2 col) * depthMapFactor (src/PlaneExtractor.cpp:64)
__device__ __forceinline__ double vertex_z(const PeacDev &P, const uint16_t *img, int row, int col) {
    const uint16_t d = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(img) + (size_t)(2 * row) * P.strideBytes + 2 * (2 * col));
    return (double)d * P.factor;
}
__device__ __forceinline__ void vertex_xy(const PeacDev &P, int row, int col, double z, double &x, double &y) {
    x = ((double)(2 * col) - P.cx) * z / P.fx;   // :69
    y = ((double)(2 * row) - P.cy) * z / P.fy;   // :70
}

__global__ __launch_bounds__(256) void k_peac_cloud(PeacDev P) {
    const int frame = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P.cw * P.ch) return;
    const int row = i / P.cw, col = i - row * P.cw;
    const uint16_t *img = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(P.depth) + (size_t)frame * P.frameStrideBytes);
    const double z = vertex_z(P, img, row, col);
    double x, y;
    vertex_xy(P, row, col, z, x, y);
    double *o = P.cloud + ((size_t)frame * P.cw * P.ch + i) * 3;
    o[0] = x; o[1] = y; o[2] = z;
}

// ImagePointCloud::get (include/PlaneExtractor.h:47-55): z == 0 is missing data (a 16-bit depth times a float is never NaN)
__device__ __forceinline__ bool cloud_get(const PeacDev &P, const uint16_t *img, int row, int col, double &x, double &y, double &z) {
    z = vertex_z(P, img, row, col);
    if (z == 0) return false;
    vertex_xy(P, row, col, z, x, y);
    return true;
}
__device__ __forceinline__ bool depth_discontinuous(const PeacDev &P, double d0, double d1) {   // AHCPlaneSeg.hpp:41-43
    return fabs(d0 - d1) > P.alpha * fabs(d0) + P.tol;
}

__global__ __launch_bounds__(64) void k_peac_blocks(PeacDev P) {
    const int frame = blockIdx.y;
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= P.Nw * P.Nh) return;
    const uint16_t *img = reinterpret_cast<const uint16_t *>(reinterpret_cast<const uint8_t *>(P.depth) + (size_t)frame * P.frameStrideBytes);
    const int seedRow = (b / P.Nw) * P.winH, seedCol = (b % P.Nw) * P.winW;
    msl_peac_stats S;
    S.sx = S.sy = S.sz = S.sxx = S.syy = S.szz = S.sxy = S.syz = S.sxz = 0; S.N = 0; S.nouse = 0;
    bool windowValid = true;
    int nanCnt = 0;
    const int nanCntTh = P.winH * P.winW / 2;
    for (int i = seedRow, icnt = 0; icnt < P.winH && i < P.ch; ++i, ++icnt) {
        for (int j = seedCol, jcnt = 0; jcnt < P.winW && j < P.cw; ++j, ++jcnt) {
            double x = 0, y = 0, z = 10000;
            if (!cloud_get(P, img, i, j, x, y, z)) {
                if (P.loose) {
                    ++nanCnt;
                    if (nanCnt < nanCntTh) continue;
                }
                windowValid = false;
                break;
            }
            double xn = 0, yn = 0, zn = 10000;
            if (j + 1 < P.cw && (cloud_get(P, img, i, j + 1, xn, yn, zn) && depth_discontinuous(P, z, zn))) { windowValid = false; break; }
            if (i + 1 < P.ch && (cloud_get(P, img, i + 1, j, xn, yn, zn) && depth_discontinuous(P, z, zn))) { windowValid = false; break; }
            S.sx += x; S.sy += y; S.sz += z;                       // Stats::push (:81-92)
            S.sxx += x * x; S.syy += y * y; S.szz += z * z;
            S.sxy += x * y; S.syz += y * z; S.sxz += x * z;
            ++S.N;
        }
        if (!windowValid) break;
    }
    if (!windowValid) { S.sx = S.sy = S.sz = S.sxx = S.syy = S.szz = S.sxy = S.syz = S.sxz = 0; S.N = 0; S.nouse = 1; }
    P.stats[(size_t)frame * P.Nw * P.Nh + b] = S;
}

struct Scratch { void *depth = nullptr, *stats = nullptr, *cloud = nullptr; size_t depthCap = 0, statsCap = 0, cloudCap = 0; };
Scratch g_scratch[16];
std::mutex g_scratchMutex;

}  // namespace

extern "C" {

int msl_peac_block_stats(int device, const uint16_t *depth, size_t depth_stride_bytes, size_t frame_stride_bytes, int width, int height,
                         int n_frames, msl_mem mem, float fx, float fy, float cx, float cy, float depth_map_factor, int window_w,
                         int window_h, double depth_alpha, double depth_change_tol, int init_loose, double *cloud_out,
                         msl_peac_stats *stats_out, msl_mem out_mem) {
    if (!depth || !stats_out || width < 2 || height < 2 || n_frames < 1 || window_w < 1 || window_h < 1 || depth_stride_bytes < (size_t)width * 2 ||
        (n_frames > 1 && frame_stride_bytes < depth_stride_bytes * (size_t)height) || fx == 0 || fy == 0) {
        set_error("msl_peac_block_stats: invalid argument");
        return MSL_ERR_INVALID;
    }
    int rc = bind_device(device);
    if (rc != MSL_OK) return rc;
    PeacDev P;
    P.width = width; P.height = height; P.cw = (width + 1) / 2; P.ch = (height + 1) / 2;   // ceil(cols / 2.0), ceil(rows / 2.0) (:51-52)
    P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.factor = depth_map_factor;
    P.winW = window_w; P.winH = window_h; P.Nw = P.cw / window_w; P.Nh = P.ch / window_h; P.loose = init_loose ? 1 : 0;
    P.alpha = depth_alpha; P.tol = depth_change_tol;
    P.strideBytes = depth_stride_bytes; P.frameStrideBytes = frame_stride_bytes;
    const size_t nBlocks = (size_t)P.Nw * P.Nh, nVert = (size_t)P.cw * P.ch;
    if (nBlocks == 0) { set_error("msl_peac_block_stats: image smaller than one block"); return MSL_ERR_INVALID; }
    // staging buffers for host-memory calls: cached per device (grow-only), so a per-frame caller pays no hipMalloc
    uint16_t *dDepth = nullptr; double *dCloud = nullptr; msl_peac_stats *dStats = nullptr;
#define PEAC_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("msl_peac_block_stats: %s", hipGetErrorString(e_)); return MSL_ERR_HIP; } } while (0)
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    Scratch &sc = g_scratch[device & 15];
    auto grow = [&](void *&p, size_t &cap, size_t need) -> hipError_t {
        if (need <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) cap = need;
        return e;
    };
    if (mem == MSL_MEM_HOST) {
        const size_t frameBytes = depth_stride_bytes * (size_t)height;
        PEAC_TRY(grow(sc.depth, sc.depthCap, frameBytes * n_frames));
        dDepth = (uint16_t *)sc.depth;
        for (int f = 0; f < n_frames; f++)
            PEAC_TRY(hipMemcpy((uint8_t *)dDepth + f * frameBytes, (const uint8_t *)depth + f * frame_stride_bytes, frameBytes, hipMemcpyHostToDevice));
        P.depth = dDepth; P.frameStrideBytes = frameBytes;
    } else {
        P.depth = depth;
    }
    if (out_mem == MSL_MEM_HOST) {
        PEAC_TRY(grow(sc.stats, sc.statsCap, sizeof(msl_peac_stats) * nBlocks * n_frames));
        dStats = (msl_peac_stats *)sc.stats;
        if (cloud_out) { PEAC_TRY(grow(sc.cloud, sc.cloudCap, sizeof(double) * 3 * nVert * n_frames)); dCloud = (double *)sc.cloud; }
    } else {
        dStats = stats_out; dCloud = cloud_out;
    }
    P.stats = dStats; P.cloud = dCloud;
    if (dCloud) hipLaunchKernelGGL(k_peac_cloud, dim3((unsigned)((nVert + 255) / 256), (unsigned)n_frames), dim3(256), 0, 0, P);
    hipLaunchKernelGGL(k_peac_blocks, dim3((unsigned)((nBlocks + 63) / 64), (unsigned)n_frames), dim3(64), 0, 0, P);
    PEAC_TRY(hipGetLastError());
    if (out_mem == MSL_MEM_HOST) {
        PEAC_TRY(hipMemcpy(stats_out, dStats, sizeof(msl_peac_stats) * nBlocks * n_frames, hipMemcpyDeviceToHost));
        if (cloud_out) PEAC_TRY(hipMemcpy(cloud_out, dCloud, sizeof(double) * 3 * nVert * n_frames, hipMemcpyDeviceToHost));
    } else {
        PEAC_TRY(hipDeviceSynchronize());
    }
#undef PEAC_TRY
    return MSL_OK;
}

}  // extern "C"
