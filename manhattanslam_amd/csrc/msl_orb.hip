// msl_orb.hip -- ORB extractor for gfx950 (MI355X): kernels + C ABI.
//
// Replaces ORB_SLAM2::ORBextractor (reference src/ORBextractor.cc).  Frame-batched pipeline, all
// integer/byte work, HBM/L2-bound; no MFMA (nothing here is a contraction).  Per batch of B frames:
//
//   k_pyramid  x1      all levels, tile chains through LDS (k_resize x(L-1) as fallback): OpenCV 11-bit fixed-point bilinear   (:872-893, cv::resize)
//   k_fast     x1      one workgroup per 30-px FAST cell: LDS-staged tile, FAST-9/16 score,
//                      in-cell 3x3 NMS, iniTh->minTh fallback, ordered ballot compaction (:745-780)
//   k_octree   x1      one workgroup per (frame, level): quadtree distribution, LDS resident (:531-721)
//   k_blur     x1      7x7 sigma-2 fixed-point separable Gaussian, LDS tile             (:851-852)
//   k_describe x1      one wave per keypoint: IC_Angle + steered BRIEF-256, output assembly
//                                                                      (:75-149, :470-475, :829-869)
//
// Float expressions that feed a rounding (angle, rotated pattern coordinates) are written in the
// reference's order and this file is compiled with -ffp-contract=off.

#include "msl_common.h"

#include <type_traits>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <memory>
#include <vector>

using namespace msl;

namespace {

constexpr int ML = 12;          // max pyramid levels
constexpr int MAXCELL = 64;     // max FAST cell extent (pixels)
constexpr int MAXNODE = 1024;   // max quadtree list length per level
constexpr int OCT_NT = 512;     // threads of the quadtree workgroup
constexpr int OCT_NODE_BYTES = 66;   // k_octree's LDS per node slot

struct LevelDev {
    int w, h, pitch;
    unsigned off;       // byte offset inside one frame's pyramid store (levels >= 1)
    unsigned boff;      // byte offset inside one frame's blurred store
    int nCols, nRows, wCell, hCell;
    int cellBase, nCells;
    int keyBase, keyCap;
    int quota;
    int nIni; float hX;
    float scale; int patch;
    unsigned xtabOff, ytabOff;  // element offsets into the resize tables
    int tileBase, tilesX, tilesY;  // blur tiling
};

struct CellDev {
    short level, x0, y0, cw, ch, _pad;
    unsigned keyOff;  // first key slot of the cell (frame relative)
};

struct ResizeTap { short s0, s1, c0, c1; };
// one axis of one tile on one level of the fused pyramid: the pixels the tile owns (writes to HBM) and the pixels it has to compute
// because its share of the next level reads them (level 0: the input pixels it loads)
struct PyrRange { short ownLo, ownHi, needLo, needHi; };

struct OrbDev {
    int nlevels, iniTh, minTh;
    int cellsPerFrame, keysPerFrame, selCap, outCap, blurTiles;
    unsigned long long pyrStride, blurStride;
    LevelDev lv[ML];
    int umax[16];
    const uint8_t *in; unsigned long long inRowStride, inFrameStride;
    uint8_t *pyr, *blur;
    const CellDev *cells;
    const ResizeTap *taps;
    const PyrRange *pyrX, *pyrY;   // fused pyramid: [level][tile column] / [level][tile row] ranges (pyrTX == 0: one launch per level)
    int pyrTX, pyrTY; unsigned pyrBuf0;   // bytes of the first LDS buffer
    uint32_t *cellCnt, *cellKeys, *keys;
    uint16_t *knode;
    uint32_t *sel; int *nsel, *ncand;
    msl_keypoint *kps; uint8_t *desc; int *nout; int *err;
    // Frame post-ORB epilogue; frameOn == 0: plain extractor
    int frameOn;
    msl_frame_params fp; float gridWInv, gridHInv;
    const float *depth; unsigned long long depthRowStride, depthFrameStride;   // bytes
    float *unXY, *depthOut, *uRight; int *gridCell;
    int nFrames;   // frames of this launch sequence (the kernels run 1-D, XCD-aware grids: xcd_item)
    int maxNode;   // k_octree: node-array length
    int octLds;    // k_octree: dynamic LDS bytes = max(OCT_NODE_BYTES * maxNode, 8 * (cells of the largest level + 1))
};

__constant__ int8_t c_pattern[1024] = {
#include "../../include/msl_orb_pattern.inc"
};

__device__ __forceinline__ const uint8_t *level_ptr(const OrbDev &P, int frame, int l, int &pitch) {
    if (l == 0) { pitch = (int)P.inRowStride; return P.in + (size_t)frame * P.inFrameStride; }
    pitch = P.lv[l].pitch;
    return P.pyr + (size_t)frame * P.pyrStride + P.lv[l].off;
}

// ---------------------------------------------------------------------------------------------
// k_resize: level l from level l-1.  One thread per output pixel; taps come from host-built tables
// so the coefficient arithmetic (double -> float -> cvRound) is the host's.
// ---------------------------------------------------------------------------------------------
// XCD-aware block -> (item, frame) mapping for the frame-batched kernels.  Workgroup g of a launch runs on XCD g % 8 (dispatch order; used for
// speed only) and every XCD has its own 4 MB L2.  With blockIdx.y = frame each frame's workgroups were spread over all eight L2s, so every
// pyramid line was fetched from HBM / Infinity Cache up to eight times (FAST cell rows of 36 bytes, blur tiles, descriptor patches: 15.2 MB per
// frame against 1.9 MB of algorithmic reads in round 3).  Launch a 1-D grid of xcd_grid1(nx * ny) workgroups instead: XCD x gets the contiguous
// run of virtual indices [x C, (x + 1) C), C = ceil(nx ny / 8), i.e. whole frames (and inside a frame neighbouring cells / tiles) share one L2.
__device__ __forceinline__ bool xcd_item(int nx, int ny, int &x, int &y) {
    const unsigned g = blockIdx.x, C = gridDim.x >> 3;
    const unsigned v = (g & 7u) * C + (g >> 3);
    if (v >= (unsigned)nx * (unsigned)ny) return false;
    x = (int)(v % (unsigned)nx); y = (int)(v / (unsigned)nx);
    return true;
}
inline unsigned xcd_grid1(long long items) { return (unsigned)(8 * ((items + 7) / 8)); }

__global__ __launch_bounds__(256) void k_resize(OrbDev P, int l) {
    const int frame = blockIdx.z;
    const LevelDev &D = P.lv[l];
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= D.w || dy >= D.h) return;
    int sp;
    const uint8_t *src = level_ptr(P, frame, l - 1, sp);
    const ResizeTap tx = P.taps[D.xtabOff + dx], ty = P.taps[D.ytabOff + dy];
    const uint8_t *r0 = src + (size_t)ty.s0 * sp, *r1 = src + (size_t)ty.s1 * sp;
    const int h0 = r0[tx.s0] * tx.c0 + r0[tx.s1] * tx.c1;
    const int h1 = r1[tx.s0] * tx.c0 + r1[tx.s1] * tx.c1;
    int v = (((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    P.pyr[(size_t)frame * P.pyrStride + D.off + (size_t)dy * D.pitch + dx] = (uint8_t)v;
}

// ---------------------------------------------------------------------------------------------
// k_pyramid: ALL levels in one launch.  The pyramid is a chain (level l is resized from level l-1, :872-893), but a tile of level l
// only depends on a slightly larger tile of level l-1: every level is cut into the same pyrTX x pyrTY grid of tiles, workgroup (i, j)
// owns tile (i, j) on every level, loads its part of the input once, and walks down the chain in LDS (two ping-pong buffers),
// recomputing the few halo pixels its next level reads beyond its own tile (~1.3x the arithmetic, 1/7 of the launches and no
// level ever re-read from HBM by the resize chain).  Each pixel is computed with k_resize's expression and taps from
// the same source values, so the levels are the same bytes; every pixel of a level is written by exactly one workgroup.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pyramid(OrbDev P) {
    extern __shared__ uint8_t s_pyr[];
    int tileI, frame;
    if (!xcd_item(P.pyrTX * P.pyrTY, P.nFrames, tileI, frame)) return;
    const int ti = tileI % P.pyrTX, tj = tileI / P.pyrTX;
    const int L = P.nlevels;
    PyrRange rx = P.pyrX[ti], ry = P.pyrY[tj];   // level 0: the input region
    int sx0 = rx.needLo, sy0 = ry.needLo, sw = rx.needHi - rx.needLo, sh = ry.needHi - ry.needLo;
    int spitch = (sw + 3) & ~3;   // row pitch of the level held in the source buffer: the input region is staged with whole dwords (round 6: four bytes per load --
                                  // the byte-wise loop was 40 trips of a 64-bit address computation and one byte per thread), later levels are packed (pitch = width)
    {
        int pitch;
        const uint8_t *img = level_ptr(P, frame, 0, pitch);
        const int nw = spitch >> 2, imgW = P.lv[0].w;
        const unsigned m = ((1u << 22) + nw - 1) / nw;   // exact floor(i / nw) for i < 2^15, nw < 2^7
        for (int i = threadIdx.x; i < nw * sh; i += 256) {
            const int r = (int)__umulhi((unsigned)i << 10, m), q = i - r * nw;   // (i m) >> 22
            const int x = sx0 + 4 * q;
            const uint8_t *src = img + (unsigned)((sy0 + r) * pitch + x);
            unsigned v = 0;
            if (x + 3 < imgW) __builtin_memcpy(&v, src, 4);   // (bytes beyond the region but inside the image row: staged, never read)
            else
                for (int e = 0; e < 4 && x + e < imgW; e++) v |= (unsigned)src[e] << (8 * e);
            reinterpret_cast<unsigned *>(s_pyr)[i] = v;
        }
    }
    __syncthreads();
    for (int l = 1; l < L; l++) {
        const LevelDev &D = P.lv[l];
        const uint8_t *src = s_pyr + (((l - 1) & 1) ? P.pyrBuf0 : 0u);
        uint8_t *dst = s_pyr + ((l & 1) ? P.pyrBuf0 : 0u);
        rx = P.pyrX[l * P.pyrTX + ti]; ry = P.pyrY[l * P.pyrTY + tj];
        const int dx0 = rx.needLo, dy0 = ry.needLo, dw = rx.needHi - rx.needLo, dh = ry.needHi - ry.needLo;
        uint8_t *out = P.pyr + (size_t)frame * P.pyrStride + D.off;
        const unsigned m = ((1u << 22) + dw - 1) / dw;
        for (int i = threadIdx.x; i < dw * dh; i += 256) {
            const int r = (int)__umulhi((unsigned)i << 10, m), c = i - r * dw;   // (i m) >> 22 without the 64-bit product
            const int dx = dx0 + c, dy = dy0 + r;
            const ResizeTap tx = P.taps[D.xtabOff + dx], ty = P.taps[D.ytabOff + dy];
            const uint8_t *r0 = src + (ty.s0 - sy0) * spitch - sx0, *r1 = src + (ty.s1 - sy0) * spitch - sx0;
            const int h0 = r0[tx.s0] * tx.c0 + r0[tx.s1] * tx.c1;
            const int h1 = r1[tx.s0] * tx.c0 + r1[tx.s1] * tx.c1;
            int v = (((ty.c0 * (h0 >> 4)) >> 16) + ((ty.c1 * (h1 >> 4)) >> 16) + 2) >> 2;
            v = min(max(v, 0), 255);
            dst[i] = (uint8_t)v;
            if (dx >= rx.ownLo && dx < rx.ownHi && dy >= ry.ownLo && dy < ry.ownHi) out[(size_t)dy * D.pitch + dx] = (uint8_t)v;
        }
        sx0 = dx0; sy0 = dy0; sw = dw; sh = dh; spitch = dw;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// k_fast: one workgroup (256 threads) per FAST cell.
// ---------------------------------------------------------------------------------------------
typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 pk_min(s16x2 a, s16x2 b) { return __builtin_elementwise_min(a, b); }
__device__ __forceinline__ s16x2 pk_max(s16x2 a, s16x2 b) { return __builtin_elementwise_max(a, b); }
__device__ __forceinline__ int fast_score16(const uint8_t *t, int tp) {
    // t points at the centre pixel inside the LDS tile (pitch tp).
    // score = max(a, -b) - 1 with a = max over the 16 nine-arcs of min(d), b = min over arcs of max(d); since
    // -b = max over arcs of min(-d), both halves are the same min/max network: run it once on packed (d, -d) pairs.
    const int v = t[0];
    const int ring[16] = {t[3 * tp], t[3 * tp + 1], t[2 * tp + 2], t[tp + 3], t[3], t[-tp + 3], t[-2 * tp + 2], t[-3 * tp + 1],
                          t[-3 * tp], t[-3 * tp - 1], t[-2 * tp - 2], t[-tp - 3], t[-3], t[tp - 3], t[2 * tp - 2], t[3 * tp - 1]};
    s16x2 x[16], lo2[16], lo4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const int d = v - ring[k]; x[k] = (s16x2){(short)d, (short)-d}; }
#pragma unroll
    for (int k = 0; k < 16; k++) lo2[k] = pk_min(x[k], x[(k + 1) & 15]);
#pragma unroll
    for (int k = 0; k < 16; k++) lo4[k] = pk_min(lo2[k], lo2[(k + 2) & 15]);
    s16x2 a = (s16x2){(short)-256, (short)-256};
#pragma unroll
    for (int k = 0; k < 16; k++) a = pk_max(a, pk_min(pk_min(lo4[k], lo4[(k + 4) & 15]), x[(k + 8) & 15]));
    return max((int)a.x, (int)a.y) - 1;
}

// Quick rejection (the classic FAST high-speed test on the 4 even opposite pairs): a 9-arc of 16 contains at least one pixel
// of every opposite pair, so a pixel whose score reaches th has, for each pair, one member brighter (or darker) than th.
// Pixels that fail cannot score >= th; their score is stored as 0, which changes neither the threshold tests nor the
// non-maximum suppression of any kept pixel (a kept pixel scores >= th, above every such neighbour either way).
__device__ __forceinline__ bool fast_candidate(const uint8_t *t, int tp, int th) {
    const int v = t[0];
    const int d0 = v - t[3 * tp], d8 = v - t[-3 * tp], d4 = v - t[3], d12 = v - t[-3];
    const int d2 = v - t[2 * tp + 2], d10 = v - t[-2 * tp - 2], d6 = v - t[-2 * tp + 2], d14 = v - t[2 * tp - 2];
    const bool bright = (d0 > th || d8 > th) && (d4 > th || d12 > th) && (d2 > th || d10 > th) && (d6 > th || d14 > th);
    const bool dark = (d0 < -th || d8 < -th) && (d4 < -th || d12 < -th) && (d2 < -th || d10 < -th) && (d6 < -th || d14 < -th);
    return bright || dark;
}

__global__ __launch_bounds__(256) void k_fast(OrbDev P) {
    __shared__ __attribute__((aligned(4))) uint8_t s_tile[(MAXCELL + 6) * (MAXCELL + 8)];
    __shared__ __attribute__((aligned(4))) uint8_t s_score[(MAXCELL + 2) * (MAXCELL + 2)];
    __shared__ unsigned s_bits[2][MAXCELL * MAXCELL / 32];   // kept-pixel bitmaps: [0] at iniTh, [1] at minTh
    __shared__ unsigned short s_list[MAXCELL * MAXCELL];   // pixels that pass the quick test
    __shared__ unsigned s_wave[17];
    __shared__ unsigned s_cnt[3];

    int cellI, frame;
    if (!xcd_item(P.cellsPerFrame, P.nFrames, cellI, frame)) return;
    const CellDev C = P.cells[cellI];
    const int tid = threadIdx.x;
    uint32_t *cnt_out = P.cellCnt + (size_t)frame * P.cellsPerFrame + cellI;
    const int cw = C.cw, ch = C.ch;
    if (cw <= 0 || ch <= 0) { if (tid == 0) *cnt_out = 0; return; }
    int pitch;
    const uint8_t *img = level_ptr(P, frame, C.level, pitch);

    // stage (cw+6) x (ch+6) pixels: rows y0-3.., cols x0-3..
    const int tw = cw + 6, th = ch + 6, tp = (tw + 3) & ~3;
    // exact floor(i / w) for i < 2^13, w <= 2^7 by multiply-shift: one division per thread instead of one per pixel
    const unsigned mCw = ((1u << 20) + cw - 1) / cw;
    {   // four bytes per load and LDS store (the tile's rows start at any byte; a row's last word may reach 3 bytes past the tile, still inside
        // the image row: the cells end 13 px before the level's right edge)
        const int nw = tp >> 2;
        const unsigned mNw = ((1u << 20) + nw - 1) / nw;
        const uint8_t *src = img + (size_t)(C.y0 - 3) * pitch + (C.x0 - 3);
        for (int i = tid; i < nw * th; i += 256) {
            const int r = (int)(((unsigned)i * mNw) >> 20), q = i - r * nw;
            uint32_t w4;
            __builtin_memcpy(&w4, src + (size_t)r * pitch + 4 * q, 4);
            *reinterpret_cast<uint32_t *>(&s_tile[r * tp + 4 * q]) = w4;
        }
    }
    const int sp = cw + 2;
    for (int i = tid; i < (sp * (ch + 2) + 3) >> 2; i += 256) reinterpret_cast<uint32_t *>(s_score)[i] = 0u;
    if (tid < MAXCELL * MAXCELL / 32) { s_bits[0][tid] = 0; s_bits[1][tid] = 0; }
    if (tid < 3) s_cnt[tid] = 0;
    __syncthreads();
    const int npx = cw * ch;
    const int thQuick = min(P.iniTh, P.minTh);
    for (int i0 = 0; i0 < npx; i0 += 256) {
        const int i = i0 + tid;
        bool cand = false;
        if (i < npx) {
            const int r = (int)(((unsigned)i * mCw) >> 20), c = i - r * cw;
            cand = fast_candidate(&s_tile[(r + 3) * tp + c + 3], tp, thQuick);
        }
        const unsigned long long m = __ballot(cand);
        if (m) {
            unsigned base = 0;
            if ((tid & 63) == 0) base = atomicAdd(&s_cnt[2], (unsigned)__popcll(m));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, 0);
            if (cand) s_list[base + __popcll(m & ((1ull << (tid & 63)) - 1ull))] = (unsigned short)i;
        }
    }
    __syncthreads();
    const int nlist = (int)s_cnt[2];
    for (int j = tid; j < nlist; j += 256) {
        const int i = s_list[j];
        const int r = (int)(((unsigned)i * mCw) >> 20), c = i - r * cw;
        const int s = fast_score16(&s_tile[(r + 3) * tp + c + 3], tp);
        s_score[(r + 1) * sp + c + 1] = (uint8_t)max(s, 0);
    }
    __syncthreads();
    // NMS + threshold flags over the candidates only (everything else scores 0): one bit per pixel and threshold
    unsigned c_ini = 0, c_min = 0;
    for (int j = tid; j < nlist; j += 256) {
        const int i = s_list[j];
        const int r = (int)(((unsigned)i * mCw) >> 20), c = i - r * cw;
        const uint8_t *q = &s_score[(r + 1) * sp + c + 1];
        const int s = q[0];
        if (s < thQuick) continue;
        const bool lm = s > q[-1] && s > q[1] && s > q[-sp - 1] && s > q[-sp] && s > q[-sp + 1] &&
                        s > q[sp - 1] && s > q[sp] && s > q[sp + 1];
        if (!lm) continue;
        if (s >= P.iniTh) { atomicOr(&s_bits[0][i >> 5], 1u << (i & 31)); c_ini++; }   // kept at iniTh
        if (s >= P.minTh) { atomicOr(&s_bits[1][i >> 5], 1u << (i & 31)); c_min++; }   // kept at minTh
    }
    if (c_ini) atomicAdd(&s_cnt[0], c_ini);
    if (c_min) atomicAdd(&s_cnt[1], c_min);
    __syncthreads();
    const int sel = s_cnt[0] ? 0 : 1;  // fallback to minTh when nothing survives at iniTh (:766-769)
    const unsigned total = s_cnt[0] ? s_cnt[0] : s_cnt[1];
    if (tid == 0) *cnt_out = total;
    if (total == 0) return;
    // ordered compaction, row-major inside the cell: thread t owns the 32 pixels of bitmap word t; its keys follow those
    // of the lower words (one scan of the word popcounts) in ascending bit order
    uint32_t *out = P.cellKeys + (size_t)frame * P.keysPerFrame + C.keyOff;
    static_assert(MAXCELL * MAXCELL / 32 <= 256, "one bitmap word per thread");
    unsigned word = tid < MAXCELL * MAXCELL / 32 ? s_bits[sel][tid] : 0u;
    unsigned tot;
    unsigned pos = block_excl_scan((unsigned)__popc(word), s_wave, &tot);
    while (word) {
        const int i = tid * 32 + __builtin_ctz(word);
        word &= word - 1;
        const int r = (int)(((unsigned)i * mCw) >> 20), c = i - r * cw;
        const unsigned kx = C.x0 + c - 16, ky = C.y0 + r - 16;  // border-frame coordinates (:773-774)
        out[pos++] = kx | (ky << 12) | ((unsigned)s_score[(r + 1) * sp + c + 1] << 24);
    }
}

// ---------------------------------------------------------------------------------------------
// k_octree: DistributeOctTree for one (frame, level).  The reference's std::list semantics are
// expressed with prefix sums: after a full round the list is [children of the expanded nodes, last
// expanded first, each as n4,n3,n2,n1] ++ [the one-key nodes in their old order]; in "careful"
// mode nodes are expanded largest first (ties: later created first) until the list reaches N.
// Keys never move: each keeps the list position of its node; the winner per final node is the
// max-response key, earliest in input order on ties.
// ---------------------------------------------------------------------------------------------
struct Rect { short x0, y0, x1, y1; };

__device__ __forceinline__ int quadrant(const Rect r, int x, int y) {
    const int mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1), my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
    return (x < mx ? 0 : 1) + (y < my ? 0 : 2);
}
__device__ __forceinline__ Rect child_rect(const Rect r, int q) {
    const short mx = r.x0 + ((r.x1 - r.x0 + 1) >> 1), my = r.y0 + ((r.y1 - r.y0 + 1) >> 1);
    Rect c;
    c.x0 = (q & 1) ? mx : r.x0; c.x1 = (q & 1) ? r.x1 : mx;
    c.y0 = (q & 2) ? my : r.y0; c.y1 = (q & 2) ? r.y1 : my;
    return c;
}

// Experiment builds (-DMSL_OCT_STAMPS): the level-0 workgroup of frame 0 parks (100 MHz clock, shader clock) pairs at its phase boundaries behind
// the error word; msl_orb_debug_stamps reads them.
#ifdef MSL_OCT_STAMPS
#define OCT_STAMP(P, level, frame, idx)                                                                                   \
    do {                                                                                                                  \
        const int i_ = (idx);                                                                                             \
        if ((level) == 0 && (frame) == 0 && threadIdx.x == 0 && i_ < 100) {                                               \
            unsigned long long *d_ = reinterpret_cast<unsigned long long *>(reinterpret_cast<unsigned char *>((P).err) + 128); \
            d_[2 * i_] = __builtin_amdgcn_s_memrealtime(); d_[2 * i_ + 1] = __builtin_amdgcn_s_memtime();                 \
        }                                                                                                                 \
    } while (0)
#else
#define OCT_STAMP(P, level, frame, idx) do { } while (0)
#endif
// Where a workgroup keeps the level's candidate keys and their node ids while the list is subdivided.  REG: in registers (OCT_KPT per thread:
// <= 8192 candidates per level, which covers every frame but synthetic noise) -- a key never moves, only its node id changes, so no round
// touches memory for them; otherwise in the global keys / knode arrays as in rounds 1-3 (slower: every pass is a chain of L2 round trips).
// KPT = keys per thread held in registers (16: <= 8192 candidates per level, 32: <= 16384 -- level 0 of a 1280 x 960 frame has ~14 k); 0 = global arrays.
template <int KPT> struct OctKeys {
    static constexpr bool REG = KPT > 0;
    static constexpr int OCT_KPT = KPT > 0 ? KPT : 1;
    uint32_t key[OCT_KPT];
    unsigned node[OCT_KPT], quad[OCT_KPT];
    uint32_t *gkeys; uint16_t *gnode; unsigned n; int tid;
    unsigned per;   // REG: a thread owns the `per` CONSECUTIVE keys k = tid * per + j -- neighbours in FAST-cell order, i.e. mostly in the same node
    // f(k, key, node&, quad&): quad is scratch that survives between two passes of one round (REG only; recomputed otherwise)
    template <typename F> __device__ __forceinline__ void for_each(F f) {
        if constexpr (REG) {
#pragma unroll
            for (int j = 0; j < OCT_KPT; j++) {
                const unsigned k = (unsigned)tid * per + (unsigned)j;
                if ((unsigned)j < per && k < n) f(k, key[j], node[j], quad[j]);
            }
        } else {
            for (unsigned k = tid; k < n; k += OCT_NT) {
                unsigned nd = gnode[k], q = 0xFFFFFFFFu;
                const unsigned nd0 = nd;
                f(k, gkeys[k], nd, q);
                if (nd != nd0) gnode[k] = (uint16_t)nd;
            }
        }
    }
};

template <int KPT>
__device__ __forceinline__ void octree_body(const OrbDev &P, const LevelDev &G, OctKeys<KPT> &K, unsigned char *s_dyn, unsigned *s_wave, int *s_misc, int frame,
                                            int level) {
    constexpr bool REG = KPT > 0;
    constexpr int OCT_KPT = KPT > 0 ? KPT : 1;
    const int M = P.maxNode, tid = threadIdx.x, N = G.quota;
    const unsigned n = K.n;
    Rect *const s_rectB = reinterpret_cast<Rect *>(s_dyn);                         // [2][M]
    unsigned *const s_cntB = reinterpret_cast<unsigned *>(s_rectB + 2 * M);        // [2][M]
    unsigned *const s_cc = s_cntB + 2 * M;                                         // [4 M] child key counts
    short *const s_crankB = reinterpret_cast<short *>(s_cc + 4 * M);               // [2][M] creation rank among the nodes recorded for careful mode, -1 = none
    unsigned *const s_FG = reinterpret_cast<unsigned *>(s_crankB + 2 * M);         // [4 M] scans of "child exists" (low half) and "child expandable" (high half)
    unsigned short *const s_L = reinterpret_cast<unsigned short *>(s_FG + 4 * M);  // [M] scan of kept old nodes
    unsigned short *const s_order = s_L + M;                                       // [M] careful mode: sorted candidates
    unsigned short *const s_rankOf = s_order + M;                                  // [M] careful mode: node -> sorted rank (0xFFFF = not a candidate)
    int *nsel = P.nsel + frame * P.nlevels + level;
    uint32_t *sel = P.sel + ((size_t)frame * P.nlevels + level) * P.selCap;

    const unsigned lane = (unsigned)tid & 63u;
    // Histogram update by whole waves: ctr[q] += number of lanes that hold q (q = 0xFFFFFFFF: no item).  The keys arrive in FAST-cell order, so
    // the lanes of a wave sit in the same one or two nodes round after round, and plain LDS atomics would send all 64 lanes to the same counter,
    // which the LDS serialises (measured: 78 of the kernel's 124 us went into the full rounds).  Here the lanes that share a counter are found
    // with a ballot and their leader issues ONE atomic with the count: as many steps as the wave has distinct counters.
    auto hist_add = [&](unsigned *ctr, unsigned q) {
        unsigned long long todo = __ballot(q != 0xFFFFFFFFu);
        while (todo) {
            const unsigned leader = (unsigned)__builtin_ctzll(todo);
            const unsigned qq = (unsigned)__builtin_amdgcn_readlane((int)q, (int)leader);
            const unsigned long long mm = __ballot(q == qq);
            if (lane == leader) atomicAdd(&ctr[qq], (unsigned)__popcll(mm));
            todo &= ~mm;
        }
    };
    // f(k, key, node&, quad&) -> counter index or 0xFFFFFFFF, for every key.  REG: a thread's consecutive keys mostly share a counter, so it counts
    // runs and issues one atomic per run (different lanes sit in different cells: few of them meet in a counter at the same time).
    auto for_each_hist = [&](unsigned *ctr, auto f) {
        if constexpr (REG) {
            unsigned curQ = 0xFFFFFFFFu, curC = 0;
#pragma unroll
            for (int j = 0; j < OCT_KPT; j++) {
                const unsigned k = (unsigned)tid * K.per + (unsigned)j;
                if ((unsigned)j < K.per && k < n) {
                    const unsigned q = f(k, K.key[j], K.node[j], K.quad[j]);
                    if (q != curQ) {
                        if (curQ != 0xFFFFFFFFu) atomicAdd(&ctr[curQ], curC);
                        curQ = q; curC = 0;
                    }
                    curC++;
                }
            }
            if (curQ != 0xFFFFFFFFu) atomicAdd(&ctr[curQ], curC);
        } else {
            for (unsigned k0 = (unsigned)tid - lane; k0 < n; k0 += OCT_NT) {
                const unsigned k = k0 + lane;
                unsigned q = 0xFFFFFFFFu;
                if (k < n) {
                    unsigned nd = K.gnode[k], qd = 0xFFFFFFFFu;
                    const unsigned nd0 = nd;
                    q = f(k, K.gkeys[k], nd, qd);
                    if (nd != nd0) K.gnode[k] = (uint16_t)nd;
                }
                hist_add(ctr, q);
            }
        }
    };

    int stampNo = 2;
    OCT_STAMP(P, level, frame, 1);
    // ---- root nodes (:536-572) ----
    int cur = 0, S = 0;
    const int H = G.h - 32;
    for (int i = tid; i < M; i += OCT_NT) { s_cntB[i] = 0; s_crankB[i] = -1; }
    for (int j = tid; j < 4 * M; j += OCT_NT) s_cc[j] = 0;
    __syncthreads();
    for_each_hist(s_cntB, [&](unsigned, uint32_t key, unsigned &node, unsigned &) {
        node = (unsigned)(int)((float)(key & 0xFFF) / G.hX);
        return node;
    });
    __syncthreads();
    if (tid == 0) {
        int sN = 0;
        for (int i = 0; i < G.nIni; i++) {
            const unsigned c = s_cntB[i];
            s_L[i] = (unsigned short)sN;
            if (c) {
                Rect r; r.x0 = (short)(int)(G.hX * (float)i); r.x1 = (short)(int)(G.hX * (float)(i + 1)); r.y0 = 0; r.y1 = (short)H;
                s_rectB[M + sN] = r; s_cntB[M + sN] = c; s_crankB[M + sN] = -1;
                sN++;
            }
        }
        s_misc[0] = sN;
    }
    __syncthreads();
    S = s_misc[0];
    K.for_each([&](unsigned, uint32_t, unsigned &node, unsigned &) { node = s_L[node]; });
    cur = 1;
    __syncthreads();

    // Child counts of the nodes that are being divided: s_cc[4 i + quadrant] += 1 for every key of a divided node i (s_cc is zero on entry).
    auto count_children = [&](const Rect *rect, auto is_divided) {
        for_each_hist(s_cc, [&](unsigned, uint32_t key, unsigned &node, unsigned &quad) {
            quad = is_divided(node) ? (unsigned)(4 * node + quadrant(rect[node], key & 0xFFF, (key >> 12) & 0xFFF)) : 0xFFFFFFFFu;
            return quad;
        });
    };

    bool careful = false;
    // ---- full rounds (:580-640).  Five barriers per round: the scans of "child exists" / "child expandable" / "old node kept" run as ONE pair of
    // block scans over register values (each thread owns `per` consecutive child slots), the new node records are written by the thread that scanned
    // them, and the counters of the next round are cleared while the keys move to their new nodes. ----
    while (true) {
        const int prevS = S;
        Rect *rect = (s_rectB + cur * M); unsigned *cnt = (s_cntB + cur * M);
        Rect *nrect = (s_rectB + (cur ^ 1) * M); unsigned *ncnt = (s_cntB + (cur ^ 1) * M); short *ncrank = (s_crankB + (cur ^ 1) * M);
        OCT_STAMP(P, level, frame, stampNo++);
        count_children(rect, [&](unsigned i) { return cnt[i] > 1; });
        __syncthreads();
        OCT_STAMP(P, level, frame, stampNo++);
        const int per = (4 * S + OCT_NT - 1) / OCT_NT, j0 = tid * per;
        unsigned sumFG = 0, sumL = 0;
        for (int p = 0; p < per; p++) {
            const int j = j0 + p;
            if (j < 4 * S) {
                const unsigned ci = cnt[j >> 2], c = ci > 1 ? s_cc[j] : 0;
                sumFG += (c > 0 ? 1u : 0u) | (c > 1 ? 0x10000u : 0u);
                sumL += ((j & 3) == 0 && ci == 1) ? 1u : 0u;
            }
        }
        unsigned totFG, totL, exFG, exL;
        block_excl_scan_pair(sumFG, sumL, s_wave, &totFG, &totL, exFG, exL);
        OCT_STAMP(P, level, frame, stampNo++);
        const int Ctot = (int)(totFG & 0xFFFFu), nToExpand = (int)(totFG >> 16), Ltot = (int)totL;
        const int S2 = Ctot + Ltot;
        if (S2 > M) { if (tid == 0) { atomicExch(P.err, 1); *nsel = 0; } return; }
        for (int p = 0; p < per; p++) {
            const int j = j0 + p;
            if (j < 4 * S) {
                const int i = j >> 2;
                const unsigned ci = cnt[i], c = ci > 1 ? s_cc[j] : 0;
                exFG += (c > 0 ? 1u : 0u) | (c > 1 ? 0x10000u : 0u);      // inclusive from here
                s_FG[j] = exFG;
                if (c) {
                    const int pos = Ctot - (int)(exFG & 0xFFFFu);
                    nrect[pos] = child_rect(rect[i], j & 3);
                    ncnt[pos] = c;
                    ncrank[pos] = c > 1 ? (short)((exFG >> 16) - 1) : (short)-1;
                }
                if ((j & 3) == 0) {
                    exL += ci == 1 ? 1u : 0u;
                    s_L[i] = (unsigned short)exL;
                    if (ci == 1) { const int pos = Ctot + (int)exL - 1; nrect[pos] = rect[i]; ncnt[pos] = 1; ncrank[pos] = -1; }
                }
            }
        }
        __syncthreads();
        OCT_STAMP(P, level, frame, stampNo++);
        K.for_each([&](unsigned, uint32_t key, unsigned &node, unsigned &quad) {
            const unsigned i = node;
            if (cnt[i] > 1) {
                const unsigned j = REG ? quad : (unsigned)(4 * i + quadrant(rect[i], key & 0xFFF, (key >> 12) & 0xFFF));
                node = (unsigned)(Ctot - (int)(s_FG[j] & 0xFFFFu));
            } else {
                node = (unsigned)(Ctot + s_L[i] - 1);
            }
        });
        for (int j = tid; j < 4 * S2; j += OCT_NT) s_cc[j] = 0;
        __syncthreads();
        OCT_STAMP(P, level, frame, stampNo++);
        cur ^= 1; S = S2;
        if (S >= N || S == prevS) break;
        if (S + 3 * nToExpand > N) { careful = true; break; }
    }

    // ---- careful mode (:641-700) ----
    stampNo = 40;
    while (careful) {
        OCT_STAMP(P, level, frame, stampNo++);
        const int prevS = S;
        Rect *rect = (s_rectB + cur * M); unsigned *cnt = (s_cntB + cur * M); short *crank = (s_crankB + cur * M);
        Rect *nrect = (s_rectB + (cur ^ 1) * M); unsigned *ncnt = (s_cntB + (cur ^ 1) * M); short *ncrank = (s_crankB + (cur ^ 1) * M);
        count_children(rect, [&](unsigned i) { return crank[i] >= 0; });
        OCT_STAMP(P, level, frame, stampNo++);
        // rank of a candidate = number of candidates that sort before it: larger (size, creation rank) first.  The sort keys go to LDS first
        // (0 = not a candidate; s_FG is free until the scan below) so that the comparison loop is branch-free and its loads pipeline: as a loop
        // over cnt[] / crank[] with a short-circuit test it took 6.5 of the careful round's 12 us.
        unsigned long long *const pk = reinterpret_cast<unsigned long long *>(s_FG);
        for (int i = tid; i < S; i += OCT_NT) pk[i] = crank[i] >= 0 ? ((((unsigned long long)cnt[i] << 16) | (unsigned)crank[i]) + 1ull) : 0ull;
        __syncthreads();
        unsigned nMine = 0;
        for (int i = tid; i < S; i += OCT_NT) {
            const unsigned long long me = pk[i];
            if (me == 0) { s_rankOf[i] = 0xFFFF; continue; }
            int r = 0;
#pragma unroll 8
            for (int i2 = 0; i2 < S; i2++) r += pk[i2] > me ? 1 : 0;
            s_order[r] = (unsigned short)i;
            s_rankOf[i] = (unsigned short)r;
            nMine++;
        }
        if (tid == 0) s_misc[2] = 0x7FFFFFFF;
        unsigned mTot, exUnused;
        exUnused = block_excl_scan(nMine, s_wave, &mTot);   // (its barriers also complete s_cc, s_order, s_rankOf)
        (void)exUnused;
        const int m = (int)mTot;
        OCT_STAMP(P, level, frame, stampNo++);
        if (m == 0) break;  // nothing to expand: size unchanged -> finish
        // growth prefix in sorted order -> number of expansions J: the first r + 1 with S + (children of the first r + 1 expansions) - (r + 1) >= N
        const int perR = (m + OCT_NT - 1) / OCT_NT, r0 = tid * perR;
        unsigned sumG = 0;
        for (int p = 0; p < perR; p++) {
            const int r = r0 + p;
            if (r < m) { const int i = s_order[r]; sumG += (s_cc[4 * i] > 0) + (s_cc[4 * i + 1] > 0) + (s_cc[4 * i + 2] > 0) + (s_cc[4 * i + 3] > 0); }
        }
        unsigned totG, exG;
        exG = block_excl_scan(sumG, s_wave, &totG);
        for (int p = 0; p < perR; p++) {
            const int r = r0 + p;
            if (r < m) {
                const int i = s_order[r];
                exG += (s_cc[4 * i] > 0) + (s_cc[4 * i + 1] > 0) + (s_cc[4 * i + 2] > 0) + (s_cc[4 * i + 3] > 0);
                if (S + (int)exG - (r + 1) >= N) atomicMin(&s_misc[2], r + 1);
            }
        }
        __syncthreads();
        const int J = min(s_misc[2], m);
        OCT_STAMP(P, level, frame, stampNo++);
        // children of the J expanded nodes, flat index jj = 4 r + q in sorted order; kept nodes in their old order
        const int perC = (4 * J + OCT_NT - 1) / OCT_NT, jj0 = tid * perC, perS = (S + OCT_NT - 1) / OCT_NT, i0 = tid * perS;
        unsigned sumFG = 0, sumL = 0;
        for (int p = 0; p < perC; p++) {
            const int jj = jj0 + p;
            if (jj < 4 * J) { const unsigned c = s_cc[4 * s_order[jj >> 2] + (jj & 3)]; sumFG += (c > 0 ? 1u : 0u) | (c > 1 ? 0x10000u : 0u); }
        }
        for (int p = 0; p < perS; p++) { const int i = i0 + p; if (i < S) sumL += !(s_rankOf[i] < J) ? 1u : 0u; }
        unsigned totFG, totL, exFG, exL;
        block_excl_scan_pair(sumFG, sumL, s_wave, &totFG, &totL, exFG, exL);
        OCT_STAMP(P, level, frame, stampNo++);
        const int Ctot = (int)(totFG & 0xFFFFu);
        const int S2 = Ctot + (S - J);
        if (S2 > M) { if (tid == 0) { atomicExch(P.err, 1); *nsel = 0; } return; }
        for (int p = 0; p < perC; p++) {
            const int jj = jj0 + p;
            if (jj < 4 * J) {
                const int i = s_order[jj >> 2];
                const unsigned c = s_cc[4 * i + (jj & 3)];
                exFG += (c > 0 ? 1u : 0u) | (c > 1 ? 0x10000u : 0u);
                s_FG[jj] = exFG;
                if (c) {
                    const int pos = Ctot - (int)(exFG & 0xFFFFu);
                    nrect[pos] = child_rect(rect[i], jj & 3);
                    ncnt[pos] = c;
                    ncrank[pos] = c > 1 ? (short)((exFG >> 16) - 1) : (short)-1;
                }
            }
        }
        for (int p = 0; p < perS; p++) {
            const int i = i0 + p;
            if (i < S) {
                const bool keep = !(s_rankOf[i] < J);
                exL += keep ? 1u : 0u;
                s_L[i] = (unsigned short)exL;
                if (keep) { const int pos = Ctot + (int)exL - 1; nrect[pos] = rect[i]; ncnt[pos] = cnt[i]; ncrank[pos] = -1; }
            }
        }
        __syncthreads();
        K.for_each([&](unsigned, uint32_t key, unsigned &node, unsigned &quad) {
            const unsigned i = node;
            const int r = s_rankOf[i];
            if (r < J) {
                const unsigned q = REG ? (quad & 3u) : (unsigned)quadrant(rect[i], key & 0xFFF, (key >> 12) & 0xFFF);
                node = (unsigned)(Ctot - (int)(s_FG[4 * r + q] & 0xFFFFu));
            } else {
                node = (unsigned)(Ctot + s_L[i] - 1);
            }
        });
        __syncthreads();   // every key has read s_rankOf / s_FG / s_L / the old s_cc: they may be overwritten
        OCT_STAMP(P, level, frame, stampNo++);
        for (int j = tid; j < 4 * max(S, S2); j += OCT_NT) s_cc[j] = 0;
        cur ^= 1; S = S2;
        if (S >= N || S == prevS) break;
        __syncthreads();
    }

    __syncthreads();
    OCT_STAMP(P, level, frame, 80);
    // ---- keep the best key of every node (:703-718) ----
    unsigned *best = s_cc;
    for (int i = tid; i < S; i += OCT_NT) best[i] = 0;
    __syncthreads();
    K.for_each([&](unsigned k, uint32_t key, unsigned &node, unsigned &) { atomicMax(&best[node], (key & 0xFF000000u) | (0xFFFFFFu - k)); });   // (final lists are long: few lanes share a node)
    __syncthreads();
    if (S > P.selCap) { if (tid == 0) { atomicExch(P.err, 2); *nsel = 0; } return; }
    // the winner's key: (score, index) identifies it; its thread hands it over (REG) or it is read back (global)
    if constexpr (REG) {
        K.for_each([&](unsigned k, uint32_t key, unsigned &node, unsigned &) {
            if ((best[node] & 0xFFFFFFu) == 0xFFFFFFu - k) sel[node] = key;
        });
    } else {
        for (int i = tid; i < S; i += OCT_NT) sel[i] = K.gkeys[0xFFFFFFu - (best[i] & 0xFFFFFFu)];
    }
    if (tid == 0) *nsel = S;
    OCT_STAMP(P, level, frame, 81);
}

// MAXKPT: the largest register-resident form this instantiation carries.  The 32-keys-per-thread form needs 239 VGPRs (a 512-thread workgroup then
// fills its CU's register files), so launches for ordinary frame sizes use k_octree<16> (133 VGPRs) and only large frames k_octree<32>.
template <int MAXKPT>
__global__ __launch_bounds__(OCT_NT) void k_octree(OrbDev P) {
    // Node arrays sized for THIS extractor's longest possible list (P.maxNode = max over levels of max(quota, 4 nIni) + 2, rounded up to 64, plus
    // one block of slack; 66 bytes per node: 21 KB for 1000 features instead of a fixed 66 KB for MAXNODE = 1024), so that the frame-batched kernels
    // of the other streams keep their LDS -- and with it their occupancy -- while this latency-bound kernel runs.  The same memory first holds
    // the exclusive scan of the level's per-cell candidate counts (P.octLds covers both uses).
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    __shared__ unsigned s_wave[33];   // (block_excl_scan_pair keeps two rows of 16 wave totals)
    __shared__ int s_misc[8];
    const int tid = threadIdx.x;
    int level, frame;
    if (!xcd_item(P.nlevels, P.nFrames, level, frame)) return;
    OCT_STAMP(P, level, frame, 0);
    const LevelDev &G = P.lv[level];
    uint32_t *keys = P.keys + (size_t)frame * P.keysPerFrame + G.keyBase;
    uint16_t *knode = P.knode + (size_t)frame * P.keysPerFrame + G.keyBase;

    // ---- the cell lists in cell order (vToDistributeKeys, :757-779): key slot k belongs to the cell c with off[c] <= k < off[c + 1] ----
    const uint32_t *cellCnt = P.cellCnt + (size_t)frame * P.cellsPerFrame + G.cellBase;
    const uint32_t *cellKeys = P.cellKeys + (size_t)frame * P.keysPerFrame;
    unsigned *const s_off = reinterpret_cast<unsigned *>(s_dyn);     // [nCells + 1] exclusive scan of the cells' candidate counts
    unsigned *const s_koff = s_off + (G.nCells + 1);                 // [nCells] where each cell's list starts in cellKeys (a geometry constant)
    unsigned n = 0;
    {
        unsigned carry = 0;
        for (int c0 = 0; c0 < G.nCells; c0 += OCT_NT) {
            const int c = c0 + tid;
            const unsigned cnt = c < G.nCells ? cellCnt[c] : 0;
            const unsigned ko = c < G.nCells ? P.cells[G.cellBase + c].keyOff : 0;   // (in flight together with the count)
            unsigned tot;
            const unsigned off = carry + block_excl_scan(cnt, s_wave, &tot);
            if (c < G.nCells) { s_off[c] = off; s_koff[c] = ko; }
            carry += tot;
        }
        n = carry;
        if (tid == 0) s_off[G.nCells] = n;
    }
    if (tid == 0) P.ncand[frame * P.nlevels + level] = (int)n;
    __syncthreads();
    if (n == 0) { if (tid == 0) P.nsel[frame * P.nlevels + level] = 0; return; }
    auto cell_of = [&](unsigned k) -> int {   // largest c with off[c] <= k (empty cells share their successor's offset: the search skips them)
        int lo = 0, hi = G.nCells;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= k) lo = mid; else hi = mid; }
        return lo;
    };
    auto run_in_registers = [&](auto kptTag) {
        constexpr int OCT_KPT = decltype(kptTag)::value;
        OctKeys<OCT_KPT> K;
        K.n = n; K.tid = tid; K.gkeys = keys; K.gnode = knode; K.per = (n + OCT_NT - 1) / OCT_NT;
        int c = 0;
#pragma unroll
        for (int j = 0; j < OCT_KPT; j++) {
            const unsigned k = (unsigned)tid * K.per + (unsigned)j;
            K.key[j] = 0; K.node[j] = 0; K.quad[j] = 0xFFFFFFFFu;
            if ((unsigned)j < K.per && k < n) {
                // one search for the thread's first key, then a walk: its keys are consecutive, so they sit in one or two cells, and their loads
                // leave together instead of each waiting for its own search and cell record
                if (j == 0) c = cell_of(k);
                else while (k >= s_off[c + 1]) c++;
                K.key[j] = cellKeys[s_koff[c] + (k - s_off[c])];
            }
        }
#pragma unroll
        for (int j = 0; j < OCT_KPT; j++) {
            const unsigned k = (unsigned)tid * K.per + (unsigned)j;
            if ((unsigned)j < K.per && k < n) keys[k] = K.key[j];     // (the global copy is what msl_orb_debug_candidates reads)
        }
        __syncthreads();   // s_off is dead from here on: the node arrays take its place
        octree_body<OCT_KPT>(P, G, K, s_dyn, s_wave, s_misc, frame, level);
    };
    if (n <= 16u * OCT_NT) {
        run_in_registers(std::integral_constant<int, 16>{});
    } else if (MAXKPT >= 32 && n <= 32u * OCT_NT) {   // level 0 of a 1280 x 960 frame: ~14 k candidates
        if constexpr (MAXKPT >= 32) run_in_registers(std::integral_constant<int, 32>{});
    } else {
        OctKeys<0> K;
        K.n = n; K.tid = tid; K.gkeys = keys; K.gnode = knode; K.per = 0;
        for (unsigned k = tid; k < n; k += OCT_NT) { const int c = cell_of(k); keys[k] = cellKeys[s_koff[c] + (k - s_off[c])]; knode[k] = 0; }
        __syncthreads();
        octree_body<0>(P, G, K, s_dyn, s_wave, s_misc, frame, level);
    }
}

// ---------------------------------------------------------------------------------------------
// k_blur: GaussianBlur 7x7 sigma 2 (integer kernel {18,34,49,55,49,34,18} by default, >>16 with rounding),
// reflect-101 on the level's own borders.  Tile 64 x 32 outputs per workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int BT_W = 64, BT_H = 32;
// Which OpenCV generation's 8-bit Gaussian kernel is pinned: 0 = per-coefficient rounding {18,34,49,55,..}
// (OpenCV 3.x, sum 257), 1 = error-diffused "bit-exact" kernel {18,34,48,56,..} (later releases, sum 256).  Row sums stay <= 65535.
#ifndef MSL_BLUR_VARIANT
#define MSL_BLUR_VARIANT 0
#endif
#if MSL_BLUR_VARIANT == 1
constexpr int GK0 = 18, GK1 = 34, GK2 = 48, GK3 = 56;
#else
constexpr int GK0 = 18, GK1 = 34, GK2 = 49, GK3 = 55;
#endif

__device__ __forceinline__ int reflect101(int p, int len) {
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void k_blur(OrbDev P) {
    // 64 x 32 output tile, four horizontally adjacent pixels per thread in both passes: dword traffic to LDS and memory
    // instead of bytes.  Arithmetic as before: u16 row sums (<= 65535), 32-bit column sums, (acc + 32768) >> 16.
    constexpr int IP = BT_W + 8;                                   // staged bytes per row: x = tx0 - 4 .. tx0 + 67
    __shared__ __attribute__((aligned(16))) uint8_t s_in[(BT_H + 6) * IP];
    __shared__ __attribute__((aligned(16))) unsigned short s_row[(BT_H + 6) * BT_W];
    const int tid = threadIdx.x;
    int tileI, frame;
    if (!xcd_item(P.blurTiles, P.nFrames, tileI, frame)) return;
    int l = 0;
    while (l + 1 < P.nlevels && tileI >= P.lv[l + 1].tileBase) l++;
    const LevelDev &D = P.lv[l];
    const int t = tileI - D.tileBase;
    const int tx0 = (t % D.tilesX) * BT_W, ty0 = (t / D.tilesX) * BT_H;
    int pitch;
    const uint8_t *img = level_ptr(P, frame, l, pitch);
    if (tx0 >= 4 && tx0 + BT_W + 4 <= D.w && ty0 >= 3 && ty0 + BT_H + 3 <= D.h) {
        // interior tile: 18 dwords per row, no reflection
        for (int i = tid; i < (BT_H + 6) * (IP / 4); i += 256) {
            const int r = i / (IP / 4), q = i - r * (IP / 4);
            unsigned v;
            __builtin_memcpy(&v, img + (size_t)(ty0 - 3 + r) * pitch + (tx0 - 4 + 4 * q), 4);
            reinterpret_cast<unsigned *>(s_in)[r * (IP / 4) + q] = v;
        }
    } else {
        for (int i = tid; i < (BT_H + 6) * IP; i += 256) {
            const int r = i / IP, c = i - r * IP;
            const int y = reflect101(ty0 + r - 3, D.h), x = reflect101(tx0 + c - 4, D.w);
            s_in[i] = img[(size_t)y * pitch + x];
        }
    }
    __syncthreads();
    // Row pass with v_dot4_u32_u8 (round 6): the seven taps of an output are two dot products of four staged bytes each -- the 8-byte window that starts
    // one byte behind the output's first staged byte, cut out of the thread's three dwords with v_alignbyte -- instead of twelve byte extractions and ten
    // multiply / adds per output.  A thread computes the same four columns of TWO consecutive staged rows and stores them as one dword per column,
    // {row 2 p, row 2 p + 1}: the column pass then takes two taps per v_dot2_u32_u16.  Integer arithmetic throughout: the same sums as before.
    constexpr unsigned TAPS_LO = (unsigned)GK0 | ((unsigned)GK1 << 8) | ((unsigned)GK2 << 16) | ((unsigned)GK3 << 24);   // staged bytes k + 1 .. k + 4
    constexpr unsigned TAPS_HI = (unsigned)GK2 | ((unsigned)GK1 << 8) | ((unsigned)GK0 << 16);                           // staged bytes k + 5 .. k + 7
    auto row4 = [&](int r, int j, unsigned (&o)[4]) {
        const unsigned *p = reinterpret_cast<const unsigned *>(s_in) + r * (IP / 4) + j;
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), TAPS_HI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), TAPS_LO, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), TAPS_HI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), TAPS_LO, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), TAPS_HI, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), TAPS_LO, 0u, false), false);
        o[3] = __builtin_amdgcn_udot4(w2, TAPS_HI, __builtin_amdgcn_udot4(w1, TAPS_LO, 0u, false), false);   // output column 4 j + k reads staged bytes 4 j + k + 1 .. + 7
    };
    static_assert((BT_H + 6) % 2 == 0, "whole row pairs");
    unsigned *const s_pair = reinterpret_cast<unsigned *>(s_row);   // [(BT_H + 6) / 2][BT_W]: low half = row 2 p, high half = row 2 p + 1 (row sums <= 65535)
    for (int i = tid; i < ((BT_H + 6) / 2) * (BT_W / 4); i += 256) {
        const int rp = i / (BT_W / 4), j = i - rp * (BT_W / 4);
        unsigned oa[4], ob[4];
        row4(2 * rp, j, oa); row4(2 * rp + 1, j, ob);
        uint4 pk; pk.x = oa[0] | (ob[0] << 16); pk.y = oa[1] | (ob[1] << 16); pk.z = oa[2] | (ob[2] << 16); pk.w = oa[3] | (ob[3] << 16);
        *reinterpret_cast<uint4 *>(&s_pair[rp * BT_W + 4 * j]) = pk;
    }
    __syncthreads();
    uint8_t *out = P.blur + (size_t)frame * P.blurStride + D.boff;
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    for (int i = tid; i < BT_H * (BT_W / 4); i += 256) {
        const int r = i / (BT_W / 4), j = i - r * (BT_W / 4);
        const int x = tx0 + 4 * j, y = ty0 + r;
        if (x >= D.w || y >= D.h) continue;
        // output row r = staged rows r .. r + 6 with the taps K0 K1 K2 K3 K2 K1 K0: four row pairs from pair r / 2 on; an even r starts on a pair
        // (taps K0 K1 | K2 K3 | K2 K1 | K0 0), an odd r in the middle of one (0 K0 | K1 K2 | K3 K2 | K1 K0)
        const bool odd = r & 1;
        const unsigned t0 = odd ? ((unsigned)GK0 << 16) : ((unsigned)GK0 | ((unsigned)GK1 << 16));
        const unsigned t1 = odd ? ((unsigned)GK1 | ((unsigned)GK2 << 16)) : ((unsigned)GK2 | ((unsigned)GK3 << 16));
        const unsigned t2 = odd ? ((unsigned)GK3 | ((unsigned)GK2 << 16)) : ((unsigned)GK2 | ((unsigned)GK1 << 16));
        const unsigned t3 = odd ? ((unsigned)GK1 | ((unsigned)GK0 << 16)) : (unsigned)GK0;
        const uint4 *q = reinterpret_cast<const uint4 *>(&s_pair[(r >> 1) * BT_W + 4 * j]);
        const uint4 q0 = q[0], q1 = q[BT_W / 4], q2 = q[2 * (BT_W / 4)], q3 = q[3 * (BT_W / 4)];
        auto col = [&](unsigned a0, unsigned a1, unsigned a2, unsigned a3) -> unsigned {
            unsigned acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a0), __builtin_bit_cast(u16x2, t0), 32768u, false);   // (acc + 32768) >> 16
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a1), __builtin_bit_cast(u16x2, t1), acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a2), __builtin_bit_cast(u16x2, t2), acc, false);
            acc = __builtin_amdgcn_udot2(__builtin_bit_cast(u16x2, a3), __builtin_bit_cast(u16x2, t3), acc, false);
            return min(acc >> 16, 255u);
        };
        const unsigned res = col(q0.x, q1.x, q2.x, q3.x) | (col(q0.y, q1.y, q2.y, q3.y) << 8) | (col(q0.z, q1.z, q2.z, q3.z) << 16) | (col(q0.w, q1.w, q2.w, q3.w) << 24);
        uint8_t *dst = out + (size_t)y * D.pitch + x;
        if (x + 3 < D.w) __builtin_memcpy(dst, &res, 4);
        else
            for (int k = 0; k < 4 && x + k < D.w; k++) dst[k] = (uint8_t)(res >> (8 * k));
    }
}

// ---------------------------------------------------------------------------------------------
// k_describe: one wave per selected keypoint: IC_Angle, pinned sincos, steered BRIEF, output.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
    const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
    const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
    const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Pinned replacement for cosf/sinf at src/ORBextractor.cc:108 (DESIGN.md "pinned sincos").
__device__ __forceinline__ void sincos_pinned(float angle, float *s_out, float *c_out) {
    const double x = (double)angle;
    const double kd = floor(x * 6.36619772367581382433e-01 + 0.5);
    const int k = (int)kd;
    const double r = (x - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double v = z * r;
    const double sr = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sn = r + v * (S1 + z * sr);
    const double cr = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cs = 1.0 - (0.5 * z - z * cr);
    double s, c;
    switch (k & 3) {
        case 0: s = sn; c = cs; break;
        case 1: s = cs; c = -sn; break;
        case 2: s = -sn; c = -cs; break;
        default: s = -cs; c = sn; break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

// cv::undistortPoints(K, dist, P = K) for one point, OpenCV 3.x cvUndistortPoints plain C path (double arithmetic, 5
// fixed-point iterations); src/Frame.cc:437-463.  The zero terms of the 12-coefficient model are kept so that the
// operation order is the library's.
__device__ __forceinline__ void undistort_point(const msl_frame_params &p, float xin, float yin, float *xo, float *yo) {
    const double fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k0 = p.k1, k1 = p.k2, k2 = p.p1, k3 = p.p2, k4 = p.k3, kz = 0.0;
    double x = xin, y = yin;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((kz * r2 + kz) * r2 + kz) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + kz * r2 + kz * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + kz * r2 + kz * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void k_describe(OrbDev P) {
    int item, frame;
    const int perLevel = (P.selCap + 3) / 4;
    if (!xcd_item(perLevel * P.nlevels, P.nFrames, item, frame)) return;
    const int level = item / perLevel, blockX = item % perLevel;
    const int lane = threadIdx.x & 63;
    const int idx = blockX * 4 + (threadIdx.x >> 6);
    const int *nsel = P.nsel + frame * P.nlevels;
    if (blockX == 0 && level == 0 && threadIdx.x == 0) {
        int tot = 0;
        for (int l = 0; l < P.nlevels; l++) tot += nsel[l];
        P.nout[frame] = min(tot, P.outCap);
        if (tot > P.outCap) atomicExch(P.err, 3);
    }
    if (idx >= nsel[level]) return;
    int outIdx = idx;
    for (int l = 0; l < level; l++) outIdx += nsel[l];
    if (outIdx >= P.outCap) return;
    const LevelDev &D = P.lv[level];
    const unsigned key = P.sel[((size_t)frame * P.nlevels + level) * P.selCap + idx];
    const int x = (int)(key & 0xFFF) + 16, y = (int)((key >> 12) & 0xFFF) + 16;  // :793-794
    const int score = key >> 24;
    int pitch;
    const uint8_t *img = level_ptr(P, frame, level, pitch);
    // Stage the keypoint's neighbourhoods in LDS with row-wise dword loads (10 per lane) instead of ~20 scattered byte gathers per
    // lane: the 31 x 31 patch of the level (IC_Angle, |u|, |v| <= 15; columns x-16 .. x+15 are staged) and the 37 x 37 patch of the
    // blurred level (rotated BRIEF samples, |offset| <= 18; columns x-20 .. x+19).  A keypoint lies at least 19 px inside the level
    // (FAST cells start 16 + 3 px in), so every staged byte exists.
    constexpr int PW = 32, PH = 31, BW = 40, BH = 37;
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][PH * PW];
    __shared__ __attribute__((aligned(16))) uint8_t s_blurp[4][BH * BW];
    const int wv = threadIdx.x >> 6;
    const uint8_t *bl = P.blur + (size_t)frame * P.blurStride + D.boff + (size_t)y * D.pitch + x;
    {
        unsigned pv[4], bv[6];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = lane + 64 * k;   // dword i of the patch: row i / 8, dword i % 8
            pv[k] = 0;
            if (i < PH * (PW / 4)) __builtin_memcpy(&pv[k], img + (size_t)(y - 15 + i / (PW / 4)) * pitch + (x - 16 + 4 * (i % (PW / 4))), 4);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const int i = lane + 64 * k;
            bv[k] = 0;
            if (i < BH * (BW / 4)) __builtin_memcpy(&bv[k], bl + (i / (BW / 4) - 18) * D.pitch + (4 * (i % (BW / 4)) - 20), 4);
        }
#pragma unroll
        for (int k = 0; k < 4; k++) { const int i = lane + 64 * k; if (i < PH * (PW / 4)) reinterpret_cast<unsigned *>(s_patch[wv])[i] = pv[k]; }
#pragma unroll
        for (int k = 0; k < 6; k++) { const int i = lane + 64 * k; if (i < BH * (BW / 4)) reinterpret_cast<unsigned *>(s_blurp[wv])[i] = bv[k]; }
    }
    __builtin_amdgcn_wave_barrier();
    // IC_Angle (:75-99): m10 = sum u*I, m01 = sum v*I over the circular patch
    int m10 = 0, m01 = 0;
    {
        const int u = (lane & 31) - 15;
        const int au = u < 0 ? -u : u;
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int v = -15 + 2 * it + (lane >> 5);
            const int av = v < 0 ? -v : v;
            if (av <= 15 && au <= 15 && au <= P.umax[av]) {
                const int val = s_patch[wv][(v + 15) * PW + u + 16];
                m10 += u * val;
                m01 += v * val;
            }
        }
        // wave totals in the VALU (DPP row scan + row broadcasts, msl_common.h) instead of six rounds of ds_bpermute per sum; integer sums: any order
        m10 = __builtin_amdgcn_readlane((int)wave_incl_scan((unsigned)m10), 63);
        m01 = __builtin_amdgcn_readlane((int)wave_incl_scan((unsigned)m01), 63);
    }
    const float angle = fast_atan2_deg((float)m01, (float)m10);
    const float factorPI = (float)(M_PI / 180.f);
    float a, b;
    sincos_pinned(angle * factorPI, &b, &a);
    // steered BRIEF (:104-149): lane handles test pairs 4*lane .. 4*lane+3
    unsigned nib = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int8_t *pp = &c_pattern[(4 * lane + j) * 4];
        const float x0 = (float)pp[0], y0 = (float)pp[1], x1 = (float)pp[2], y1 = (float)pp[3];
        const int r0 = __float2int_rn(x0 * b + y0 * a), c0 = __float2int_rn(x0 * a - y0 * b);
        const int r1 = __float2int_rn(x1 * b + y1 * a), c1 = __float2int_rn(x1 * a - y1 * b);
        const int t0 = s_blurp[wv][(r0 + 18) * BW + c0 + 20], t1 = s_blurp[wv][(r1 + 18) * BW + c1 + 20];
        nib |= (t0 < t1 ? 1u : 0u) << j;
    }
    // lane 8 q collects the nibbles of lanes 8 q .. 8 q + 7 (inside one DPP row): row_shl:n hands lane i the value of lane i + n
    unsigned w = nib | ((unsigned)__builtin_amdgcn_update_dpp(0, (int)nib, 0x101, 0xF, 0xF, true) << 4);
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x102, 0xF, 0xF, true) << 8;
    w |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)w, 0x104, 0xF, 0xF, true) << 16;
    msl_keypoint *kp = P.kps + (size_t)frame * P.outCap + outIdx;
    uint32_t *dsc = (uint32_t *)(P.desc + ((size_t)frame * P.outCap + outIdx) * 32);
    if ((lane & 7) == 0) dsc[lane >> 3] = w;
    if (lane == 0) {
        float fx = (float)x, fy = (float)y;
        if (level != 0) { fx *= D.scale; fy *= D.scale; }  // :861-866
        kp->x = fx; kp->y = fy; kp->size = (float)D.patch; kp->angle = angle;
        kp->response = (float)score; kp->octave = level; kp->class_id = -1;
        if (P.frameOn) {
            // UndistortKeyPoints (src/Frame.cc:437-463), ComputeStereoFromRGBD (:495-513), PosInGrid (:418-427)
            const size_t o = (size_t)frame * P.outCap + outIdx;
            float ux = fx, uy = fy;
            if (P.fp.k1 != 0.0) undistort_point(P.fp, fx, fy, &ux, &uy);
            P.unXY[2 * o] = ux; P.unXY[2 * o + 1] = uy;
            const float d = *(const float *)((const uint8_t *)P.depth + (size_t)frame * P.depthFrameStride + (size_t)(int)fy * P.depthRowStride +
                                             sizeof(float) * (size_t)(int)fx);
            P.depthOut[o] = d > 0 ? d : -1.0f;
            P.uRight[o] = d > 0 ? ux - P.fp.bf / d : -1.0f;
            const int posX = (int)roundf((ux - P.fp.minX) * P.gridWInv), posY = (int)roundf((uy - P.fp.minY) * P.gridHInv);
            P.gridCell[o] = (posX < 0 || posX >= MSL_FRAME_GRID_COLS || posY < 0 || posY >= MSL_FRAME_GRID_ROWS) ? -1 : posX * MSL_FRAME_GRID_ROWS + posY;
        }
    }
}

// =============================================================================================
// Host side
// =============================================================================================
inline int cv_round_f(float v) { return (int)lrintf(v); }
inline int cv_round_d(double v) { return (int)lrint(v); }
inline int cv_floor_d(double v) { int i = (int)v; return i - (i > v); }
inline int cv_ceil_d(double v) { int i = (int)v; return i + (i < v); }

enum { KID_RESIZE = 0, KID_FAST, KID_OCTREE, KID_BLUR, KID_DESCRIBE, KID_COPY };
const char *kKernelNames[MSL_ORB_NKERNELS] = {"k_pyramid", "k_fast", "k_octree", "k_blur", "k_describe", "copy"};

}  // namespace

struct msl_orb {
    int device = 0;
    int nfeatures = 0, nlevels = 0, iniTh = 0, minTh = 0, maxW = 0, maxH = 0, maxBatch = 0;
    bool octBig = false;   // frames of more than 640 x 480 x 1.5 pixels: level 0 may hold more than 8192 FAST candidates -> k_octree<32>
    int outCap = 0;   // keypoints per frame the outputs are sized for: fixed by the creation geometry (msl_orb_capacity)
    double scaleFactor = 0;
    std::vector<float> scale, invScale, sigma2, invSigma2;
    std::vector<int> perLevel;
    int umax[16];
    // geometry is built for one frame size at a time (rebuilt if the size changes)
    int geomW = 0, geomH = 0;
    OrbDev dev{};
    hipStream_t stream = nullptr; bool ownStream = true;
    // device allocations
    uint8_t *d_in = nullptr; size_t inPitch = 0;
    uint8_t *d_pyr = nullptr, *d_blur = nullptr;
    CellDev *d_cells = nullptr; ResizeTap *d_taps = nullptr; PyrRange *d_pyrRanges = nullptr; size_t pyrLds = 0;
    uint32_t *d_cellCnt = nullptr, *d_cellKeys = nullptr, *d_keys = nullptr; uint16_t *d_knode = nullptr;
    uint32_t *d_sel = nullptr; int *d_nsel = nullptr, *d_ncand = nullptr;
    msl_keypoint *d_kps = nullptr; uint8_t *d_desc = nullptr; int *d_nout = nullptr; int *d_err = nullptr;
    int *h_err = nullptr;  // pinned
    uint8_t *d_outBlock = nullptr; size_t outBlockBytes = 0, outKpsOff = 0, outDescOff = 0;   // [nout[B] | kps[B][cap] | desc[B][cap]] in one allocation
    uint8_t *h_pinIn = nullptr, *h_pinOut = nullptr; size_t pinInBytes = 0, pinOutBytes = 0;   // pinned staging of the single-frame drop-in call
    hipStream_t sideStream = nullptr; hipEvent_t evFork = nullptr, evJoin = nullptr;             // single-frame calls: the blur runs beside FAST + quadtree
    float *d_depthIn = nullptr; size_t depthInCap = 0;     // staged depth frames (host input)
    float *d_unXY = nullptr, *d_depthOut = nullptr, *d_uRight = nullptr; int *d_gridCell = nullptr; bool frameBufs = false;
    int lastFrames = 0;
    KernelProfiler prof;
};

namespace {

void free_geometry(msl_orb *h) {
    auto F = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->d_in); F(h->d_pyr); F(h->d_blur); F(h->d_cells); F(h->d_taps); F(h->d_pyrRanges); F(h->d_cellCnt); F(h->d_cellKeys);
    F(h->d_keys); F(h->d_knode); F(h->d_sel); F(h->d_nsel); F(h->d_ncand); F(h->d_outBlock); h->d_kps = nullptr; h->d_desc = nullptr; h->d_nout = nullptr;
    h->geomW = h->geomH = 0;
}

// Level sizes, FAST cell grid, quadtree roots, resize taps, blur tiling for a w x h frame.
int build_geometry(msl_orb *h, int W, int H) {
    if (h->geomW == W && h->geomH == H) return MSL_OK;
    free_geometry(h);
    OrbDev &D = h->dev;
    memset(&D, 0, sizeof(D));
    const int L = h->nlevels, B = h->maxBatch;
    D.nlevels = L; D.iniTh = h->iniTh; D.minTh = h->minTh;
    for (int i = 0; i < 16; i++) D.umax[i] = h->umax[i];
    std::vector<CellDev> cells;
    std::vector<ResizeTap> taps;
    size_t pyrOff = 0, blurOff = 0;
    unsigned keyOff = 0;
    int tileBase = 0, maxQuota = 0, maxList = 0, needCap = 0, maxSel = 0;
    for (int l = 0; l < L; l++) {
        LevelDev &G = D.lv[l];
        const float s = h->invScale[l];
        G.w = cv_round_f((float)W * s); G.h = cv_round_f((float)H * s);    // src/ORBextractor.cc:875
        G.pitch = (G.w + 63) & ~63;
        G.scale = h->scale[l];
        G.patch = (int)(31 * h->scale[l]);                                  // :788
        G.quota = h->perLevel[l];
        maxQuota = std::max(maxQuota, G.quota);
        if (l > 0) { G.off = (unsigned)pyrOff; pyrOff += (size_t)G.pitch * G.h; }
        G.boff = (unsigned)blurOff; blurOff += (size_t)G.pitch * G.h;
        // FAST cell grid (:728-743)
        const int minB = 16, maxBX = G.w - 16, maxBY = G.h - 16;
        const float width = (float)(maxBX - minB), height = (float)(maxBY - minB);
        const float Wc = 30;
        G.nCols = (int)(width / Wc); G.nRows = (int)(height / Wc);
        if (G.nCols < 1 || G.nRows < 1) {
            set_error("level %d (%dx%d) is too small for the 30-px FAST grid", l, G.w, G.h);
            return MSL_ERR_INVALID;
        }
        G.wCell = (int)ceilf(width / G.nCols); G.hCell = (int)ceilf(height / G.nRows);
        if (G.wCell > MAXCELL || G.hCell > MAXCELL || G.w > 4095 + 16 || G.h > 4095 + 16) {
            set_error("unsupported level geometry %dx%d (cell %dx%d)", G.w, G.h, G.wCell, G.hCell);
            return MSL_ERR_INVALID;
        }
        G.cellBase = (int)cells.size(); G.nCells = G.nRows * G.nCols;
        G.keyBase = (int)keyOff;
        for (int i = 0; i < G.nRows; i++)
            for (int j = 0; j < G.nCols; j++) {
                // view = rows [iniY,maxY) x cols [iniX,maxX); cv::FAST computes its inner 3-px-inset region
                const float iniY = (float)(minB + i * G.hCell), iniX = (float)(minB + j * G.wCell);
                float maxY = iniY + G.hCell + 6, maxX = iniX + G.wCell + 6;
                CellDev c{}; c.level = (short)l; c.keyOff = keyOff;
                const bool skip = (iniY >= maxBY - 3) || (iniX >= maxBX - 6);
                if (maxY > maxBY) maxY = (float)maxBY;
                if (maxX > maxBX) maxX = (float)maxBX;
                const int cw = (int)maxX - (int)iniX - 6, chh = (int)maxY - (int)iniY - 6;
                if (!skip && cw > 0 && chh > 0) {
                    c.x0 = (short)((int)iniX + 3); c.y0 = (short)((int)iniY + 3); c.cw = (short)cw; c.ch = (short)chh;
                    keyOff += (unsigned)(((cw + 1) / 2) * ((chh + 1) / 2));  // strict 3x3 maxima are non-adjacent
                }
                cells.push_back(c);
            }
        G.keyCap = (int)keyOff - G.keyBase;
        if (G.keyCap >= (1 << 24)) { set_error("level too large"); return MSL_ERR_INVALID; }
        // quadtree roots (:536-552)
        G.nIni = (int)roundf((float)(maxBX - minB) / (maxBY - minB));
        if (G.nIni < 1) { set_error("unsupported aspect ratio (nIni = 0)"); return MSL_ERR_INVALID; }
        G.hX = (float)(maxBX - minB) / G.nIni;
        // longest quadtree list of this level: a full round only runs when its outcome stays <= quota (the first one makes <= 4 nIni nodes), the
        // one-by-one phase stops at the first length >= quota and every expansion adds <= 3
        if (std::max(G.quota, 4 * G.nIni) + 2 > MAXNODE) { set_error("per-level quota %d exceeds %d", G.quota, MAXNODE - 2); return MSL_ERR_INVALID; }
        maxList = std::max(maxList, std::max(G.quota, 4 * G.nIni) + 2);
        // keypoints this level can return: quota + 2 from the one-by-one phase (:691-696), or the <= 4 nIni nodes of the first full round when
        // that already reaches the quota (wide images with a small budget: nIni = round(width / height) roots, :536-552)
        needCap += std::max(G.quota + 2, 4 * G.nIni); maxSel = std::max(maxSel, std::max(G.quota + 2, 4 * G.nIni));
        // blur tiles
        G.tilesX = (G.w + BT_W - 1) / BT_W; G.tilesY = (G.h + BT_H - 1) / BT_H;
        G.tileBase = tileBase; tileBase += G.tilesX * G.tilesY;
        // resize taps (cv::resize INTER_LINEAR 8U tables)
        if (l > 0) {
            const int sw = D.lv[l - 1].w, sh = D.lv[l - 1].h;
            const double scale_x = 1. / ((double)G.w / sw), scale_y = 1. / ((double)G.h / sh);
            G.xtabOff = (unsigned)taps.size();
            for (int dx = 0; dx < G.w; dx++) {
                float fx = (float)((dx + 0.5) * scale_x - 0.5);
                int sx = cv_floor_d(fx);
                fx -= sx;
                if (sx < 0) { fx = 0; sx = 0; }
                if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
                ResizeTap t;
                t.s0 = (short)sx; t.s1 = (short)std::min(sx + 1, sw - 1);
                t.c0 = (short)std::min(std::max(cv_round_f((1.f - fx) * 2048), -32768), 32767);
                t.c1 = (short)std::min(std::max(cv_round_f(fx * 2048), -32768), 32767);
                if (sx + 1 >= sw) { t.c0 = 2048; t.c1 = 0; }
                taps.push_back(t);
            }
            G.ytabOff = (unsigned)taps.size();
            for (int dy = 0; dy < G.h; dy++) {
                float fy = (float)((dy + 0.5) * scale_y - 0.5);
                int sy = cv_floor_d(fy);
                fy -= sy;
                ResizeTap t;
                t.s0 = (short)std::min(std::max(sy, 0), sh - 1); t.s1 = (short)std::min(std::max(sy + 1, 0), sh - 1);
                t.c0 = (short)std::min(std::max(cv_round_f((1.f - fy) * 2048), -32768), 32767);
                t.c1 = (short)std::min(std::max(cv_round_f(fy * 2048), -32768), 32767);
                taps.push_back(t);
            }
        }
    }
    // ---- fused pyramid (k_pyramid): one tile grid for all levels, ranges per axis ----
    std::vector<PyrRange> pyrX, pyrY;
    D.pyrTX = D.pyrTY = 0;
    if (L >= 2) {
        const int TX = std::max(1, D.lv[L - 1].w / 22), TY = std::max(1, D.lv[L - 1].h / 22);
        auto axis = [&](int T, bool isX, std::vector<PyrRange> &out, std::vector<int> &extent) {
            out.assign((size_t)L * T, PyrRange{0, 0, 0, 0});
            extent.assign(L, 0);
            for (int i = 0; i < T; i++) {
                for (int l = 1; l < L; l++) {
                    const int n = isX ? D.lv[l].w : D.lv[l].h;
                    PyrRange &r = out[(size_t)l * T + i];
                    r.ownLo = (short)((long long)i * n / T); r.ownHi = (short)((long long)(i + 1) * n / T);
                }
                int lo = out[(size_t)(L - 1) * T + i].ownLo, hi = out[(size_t)(L - 1) * T + i].ownHi;
                for (int l = L - 1; l >= 1; l--) {
                    PyrRange &r = out[(size_t)l * T + i];
                    r.needLo = (short)lo; r.needHi = (short)hi;
                    extent[l] = std::max(extent[l], hi - lo);
                    const ResizeTap *tab = taps.data() + (isX ? D.lv[l].xtabOff : D.lv[l].ytabOff);
                    int slo = tab[lo].s0, shi = tab[hi - 1].s1 + 1;   // source pixels of level l-1 this range reads (taps are monotone)
                    for (int q = lo; q < hi; q++) { slo = std::min(slo, (int)std::min(tab[q].s0, tab[q].s1)); shi = std::max(shi, (int)std::max(tab[q].s0, tab[q].s1) + 1); }
                    if (l - 1 >= 1) { const PyrRange &o = out[(size_t)(l - 1) * T + i]; lo = std::min(slo, (int)o.ownLo); hi = std::max(shi, (int)o.ownHi); }
                    else { lo = slo; hi = shi; }
                }
                PyrRange &r0 = out[i];
                r0.ownLo = r0.ownHi = 0; r0.needLo = (short)lo; r0.needHi = (short)hi;
                extent[0] = std::max(extent[0], hi - lo);
            }
        };
        std::vector<int> ex, ey;
        axis(TX, true, pyrX, ex); axis(TY, false, pyrY, ey);
        size_t b0 = 0, b1 = 0;
        bool ok = true;
        for (int l = 0; l < L; l++) {
            const size_t a = (size_t)(l == 0 ? (ex[l] + 3) & ~3 : ex[l]) * ey[l];   // (k_pyramid stages the input region with a dword pitch)
            if (l & 1) b1 = std::max(b1, a); else b0 = std::max(b0, a);
            if (l >= 1 && (D.lv[l].w < TX || D.lv[l].h < TY)) ok = false;
        }
        b0 = (b0 + 15) & ~(size_t)15;
        for (int l = 0; l < L; l++) if (ex[l] >= 128 || (size_t)ex[l] * ey[l] >= (1u << 15)) ok = false;   // k_pyramid's multiply-shift row index
        if (ok && b0 + b1 <= 48 * 1024) { D.pyrTX = TX; D.pyrTY = TY; D.pyrBuf0 = (unsigned)b0; h->pyrLds = b0 + b1; }
    }
    D.cellsPerFrame = (int)cells.size();
    D.keysPerFrame = (int)keyOff;
    D.selCap = maxSel;
    D.maxNode = ((maxList + 63) & ~63) + 64;   // the analytic bound, rounded up, plus one 64-node block of slack (4 KB): an overrun would zero a whole level (P.err)
    {
        int maxCells = 0;
        for (int l = 0; l < L; l++) maxCells = std::max(maxCells, D.lv[l].nCells);
        D.octLds = (int)((std::max<size_t>((size_t)OCT_NODE_BYTES * D.maxNode, 2 * sizeof(unsigned) * (size_t)(maxCells + 1)) + 15) & ~(size_t)15);
    }
    if (D.octLds > 32 * 1024)   // (a large feature budget: more dynamic LDS than a launch gets by default)
    {
        MSL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<16>), hipFuncAttributeMaxDynamicSharedMemorySize, D.octLds));
        MSL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(k_octree<32>), hipFuncAttributeMaxDynamicSharedMemorySize, D.octLds));
    }
    if (h->outCap == 0) h->outCap = std::max(h->nfeatures + 2 * L, needCap);   // creation: the handle's capacity follows its (max_width, max_height) geometry
    if (needCap > h->outCap) {
        set_error("frame %dx%d can return %d keypoints (aspect ratio: %d quadtree roots), the extractor was created for %d; create it with this frame size", W, H,
                  needCap, D.lv[0].nIni, h->outCap);
        return MSL_ERR_INVALID;
    }
    D.outCap = h->outCap;
    h->octBig = (long long)W * H > 640ll * 480 * 3 / 2;
    D.blurTiles = tileBase;
    D.pyrStride = (pyrOff + 255) & ~(size_t)255;
    D.blurStride = (blurOff + 255) & ~(size_t)255;
    h->inPitch = (size_t)((W + 63) & ~63);

    MSL_HIP_TRY(hipMalloc(&h->d_in, h->inPitch * H * B));
    MSL_HIP_TRY(hipMalloc(&h->d_pyr, std::max<size_t>(D.pyrStride, 256) * B));
    MSL_HIP_TRY(hipMalloc(&h->d_blur, D.blurStride * B));
    MSL_HIP_TRY(hipMalloc(&h->d_cells, sizeof(CellDev) * cells.size()));
    MSL_HIP_TRY(hipMalloc(&h->d_taps, sizeof(ResizeTap) * std::max<size_t>(taps.size(), 1)));
    if (D.pyrTX) {
        MSL_HIP_TRY(hipMalloc(&h->d_pyrRanges, sizeof(PyrRange) * (pyrX.size() + pyrY.size())));
        MSL_HIP_TRY(hipMemcpy(h->d_pyrRanges, pyrX.data(), sizeof(PyrRange) * pyrX.size(), hipMemcpyHostToDevice));
        MSL_HIP_TRY(hipMemcpy(h->d_pyrRanges + pyrX.size(), pyrY.data(), sizeof(PyrRange) * pyrY.size(), hipMemcpyHostToDevice));
        D.pyrX = h->d_pyrRanges; D.pyrY = h->d_pyrRanges + pyrX.size();
    }
    MSL_HIP_TRY(hipMalloc(&h->d_cellCnt, sizeof(uint32_t) * cells.size() * B));
    MSL_HIP_TRY(hipMalloc(&h->d_cellKeys, sizeof(uint32_t) * (size_t)keyOff * B));
    MSL_HIP_TRY(hipMalloc(&h->d_keys, sizeof(uint32_t) * (size_t)keyOff * B));
    MSL_HIP_TRY(hipMalloc(&h->d_knode, sizeof(uint16_t) * (size_t)keyOff * B));
    MSL_HIP_TRY(hipMalloc(&h->d_sel, sizeof(uint32_t) * (size_t)D.selCap * L * B));
    MSL_HIP_TRY(hipMalloc(&h->d_nsel, sizeof(int) * L * B));
    MSL_HIP_TRY(hipMalloc(&h->d_ncand, sizeof(int) * L * B));
    // outputs of a call in ONE allocation, counts first: the single-frame drop-in call fetches everything with one copy
    h->outKpsOff = (sizeof(int) * (size_t)B + 255) & ~(size_t)255;
    h->outDescOff = h->outKpsOff + ((sizeof(msl_keypoint) * (size_t)D.outCap * B + 255) & ~(size_t)255);
    h->outBlockBytes = h->outDescOff + (size_t)32 * D.outCap * B;
    MSL_HIP_TRY(hipMalloc(&h->d_outBlock, h->outBlockBytes));
    h->d_nout = reinterpret_cast<int *>(h->d_outBlock);
    h->d_kps = reinterpret_cast<msl_keypoint *>(h->d_outBlock + h->outKpsOff);
    h->d_desc = h->d_outBlock + h->outDescOff;
    MSL_HIP_TRY(hipMemcpy(h->d_cells, cells.data(), sizeof(CellDev) * cells.size(), hipMemcpyHostToDevice));
    if (!taps.empty())
        MSL_HIP_TRY(hipMemcpy(h->d_taps, taps.data(), sizeof(ResizeTap) * taps.size(), hipMemcpyHostToDevice));
    D.pyr = h->d_pyr; D.blur = h->d_blur; D.cells = h->d_cells; D.taps = h->d_taps;
    D.cellCnt = h->d_cellCnt; D.cellKeys = h->d_cellKeys; D.keys = h->d_keys; D.knode = h->d_knode;
    D.sel = h->d_sel; D.nsel = h->d_nsel; D.ncand = h->d_ncand; D.err = h->d_err;
    h->geomW = W; h->geomH = H;
    return MSL_OK;
}

// Launch the whole pipeline for n frames whose pixels are already on the device.
struct FrameEpilogue {
    msl_frame_params fp; const float *depth; size_t depthRowStride, depthFrameStride;
    float *unXY, *depthOut, *uRight; int *gridCell;
};

int launch_pipeline(msl_orb *h, const uint8_t *d_gray, size_t rowStride, size_t frameStride, int n,
                    msl_keypoint *d_kps, uint8_t *d_desc, int *d_nout, const FrameEpilogue *ep = nullptr) {
    OrbDev P = h->dev;
    P.in = d_gray; P.inRowStride = rowStride; P.inFrameStride = frameStride;
    P.kps = d_kps; P.desc = d_desc; P.nout = d_nout;
    P.frameOn = ep ? 1 : 0;
    P.nFrames = n;
    if (ep) {
        P.fp = ep->fp;
        P.gridWInv = (float)MSL_FRAME_GRID_COLS / (float)(ep->fp.maxX - ep->fp.minX);   // src/Frame.cc:137-138
        P.gridHInv = (float)MSL_FRAME_GRID_ROWS / (float)(ep->fp.maxY - ep->fp.minY);
        P.depth = ep->depth; P.depthRowStride = ep->depthRowStride; P.depthFrameStride = ep->depthFrameStride;
        P.unXY = ep->unXY; P.depthOut = ep->depthOut; P.uRight = ep->uRight; P.gridCell = ep->gridCell;
    }
    hipStream_t s = h->stream;
    const int L = h->nlevels;
    static const bool perLevel = getenv("MSL_ORB_PYRAMID") && !strcmp(getenv("MSL_ORB_PYRAMID"), "levels");
    if (P.pyrTX && !perLevel) {
        h->prof.begin(KID_RESIZE, s);
        hipLaunchKernelGGL(k_pyramid, dim3(xcd_grid1((long long)P.pyrTX * P.pyrTY * n)), dim3(256), h->pyrLds, s, P);
        h->prof.end(s);
    } else {
        for (int l = 1; l < L; l++) {
            const LevelDev &G = P.lv[l];
            h->prof.begin(KID_RESIZE, s);
            hipLaunchKernelGGL(k_resize, dim3((G.w + 63) / 64, (G.h + 3) / 4, n), dim3(256), 0, s, P, l);
            h->prof.end(s);
        }
    }
    // One frame cannot fill the GPU and every kernel is a dependent launch: the blur (needs the pyramid only) then runs on a side stream beside
    // FAST + quadtree instead of behind them.  Batched calls keep one stream: their kernels fill the GPU and the order keeps each frame's data warm.
    const bool fork = n == 1 && h->sideStream && !h->prof.on;
    if (fork) {
        MSL_HIP_TRY(hipEventRecord(h->evFork, s));
        MSL_HIP_TRY(hipStreamWaitEvent(h->sideStream, h->evFork, 0));
        hipLaunchKernelGGL(k_blur, dim3(xcd_grid1((long long)P.blurTiles * n)), dim3(256), 0, h->sideStream, P);
        MSL_HIP_TRY(hipEventRecord(h->evJoin, h->sideStream));
    }
    h->prof.begin(KID_FAST, s);
    hipLaunchKernelGGL(k_fast, dim3(xcd_grid1((long long)P.cellsPerFrame * n)), dim3(256), 0, s, P);
    h->prof.end(s);
    h->prof.begin(KID_OCTREE, s);
    if (h->octBig) hipLaunchKernelGGL(k_octree<32>, dim3(xcd_grid1((long long)L * n)), dim3(OCT_NT), (size_t)P.octLds, s, P);
    else hipLaunchKernelGGL(k_octree<16>, dim3(xcd_grid1((long long)L * n)), dim3(OCT_NT), (size_t)P.octLds, s, P);
    h->prof.end(s);
    if (fork) {
        MSL_HIP_TRY(hipStreamWaitEvent(s, h->evJoin, 0));
    } else {
        h->prof.begin(KID_BLUR, s);
        hipLaunchKernelGGL(k_blur, dim3(xcd_grid1((long long)P.blurTiles * n)), dim3(256), 0, s, P);
        h->prof.end(s);
    }
    h->prof.begin(KID_DESCRIBE, s);
    hipLaunchKernelGGL(k_describe, dim3(xcd_grid1((long long)((P.selCap + 3) / 4) * L * n)), dim3(256), 0, s, P);
    h->prof.end(s);
    MSL_HIP_TRY(hipGetLastError());
    h->lastFrames = n;
    return MSL_OK;
}

int check_device_error(msl_orb *h) {
    MSL_HIP_TRY(hipMemcpyAsync(h->h_err, h->d_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    if (*h->h_err) {
        const int e = *h->h_err;
        (void)hipMemsetAsync(h->d_err, 0, sizeof(int), h->stream);
        set_error("device-side bound exceeded in ORB pipeline (code %d)", e);
        return MSL_ERR_OVERFLOW;
    }
    return MSL_OK;
}

}  // namespace

extern "C" {

msl_orb *msl_orb_create(int nfeatures, float scaleFactorF, int nlevels, int iniThFAST, int minThFAST, int max_width,
                        int max_height, int max_batch, int device) noexcept {
    try {
    if (nfeatures < 1 || nlevels < 1 || nlevels > ML || !(scaleFactorF > 1.0f) || iniThFAST < 1 || iniThFAST > 255 ||
        minThFAST < 1 || minThFAST > 255 || max_width < 1 || max_height < 1 || max_batch < 1) {
        set_error("msl_orb_create: invalid argument");
        return nullptr;
    }
    if (bind_device(device) != MSL_OK) return nullptr;
    // (owned by a guard until the handle is complete: an exception from the containers below -- std::bad_alloc -- lands in the catch barrier, and the
    // streams, events and device buffers created so far must go with it)
    std::unique_ptr<msl_orb, void (*)(msl_orb *)> guard(new msl_orb, [](msl_orb *p) { msl_orb_destroy(p); });
    msl_orb *h = guard.get();
    h->device = device; h->nfeatures = nfeatures; h->nlevels = nlevels; h->iniTh = iniThFAST; h->minTh = minThFAST;
    h->maxW = max_width; h->maxH = max_height; h->maxBatch = max_batch;
    h->scaleFactor = scaleFactorF;  // include/ORBextractor.h:97 keeps it as double
    // scale tables and per-level quotas, src/ORBextractor.cc:416-445
    h->scale.resize(nlevels); h->sigma2.resize(nlevels); h->invScale.resize(nlevels); h->invSigma2.resize(nlevels);
    h->scale[0] = 1.0f; h->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        h->scale[i] = (float)(h->scale[i - 1] * h->scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nlevels; i++) { h->invScale[i] = 1.0f / h->scale[i]; h->invSigma2[i] = 1.0f / h->sigma2[i]; }
    h->perLevel.resize(nlevels);
    const float factor = (float)(1.0f / h->scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int level = 0; level < nlevels - 1; level++) {
        h->perLevel[level] = cv_round_f(nDesired);
        sum += h->perLevel[level];
        nDesired *= factor;
    }
    h->perLevel[nlevels - 1] = std::max(nfeatures - sum, 0);
    // circular patch row ends, :453-467
    {
        int v, v0;
        const int vmax = cv_floor_d(15 * sqrtf(2.f) / 2 + 1), vmin = cv_ceil_d(15 * sqrtf(2.f) / 2);
        for (v = 0; v < 16; v++) h->umax[v] = 0;
        for (v = 0; v <= vmax; ++v) h->umax[v] = cv_round_d(sqrt(225.0 - v * v));
        for (v = 15, v0 = 0; v >= vmin; --v) {
            while (h->umax[v0] == h->umax[v0 + 1]) ++v0;
            h->umax[v] = v0;
            ++v0;
        }
    }
    int prLo = 0, prHi = 0;   // frame-batched throughput work: lowest priority, so latency-critical streams of the process go first
    (void)hipDeviceGetStreamPriorityRange(&prLo, &prHi);
    if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prLo) != hipSuccess ||
        hipMalloc(&h->d_err, 2048) != hipSuccess || hipMemset(h->d_err, 0, 2048) != hipSuccess ||   // [0] deferred error; from byte 128: 200 device-clock stamps of experiment builds
        hipHostMalloc(&h->h_err, sizeof(int)) != hipSuccess ||
        hipStreamCreateWithPriority(&h->sideStream, hipStreamNonBlocking, prLo) != hipSuccess ||
        hipEventCreateWithFlags(&h->evFork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&h->evJoin, hipEventDisableTiming) != hipSuccess) {
        set_error("msl_orb_create: HIP resource allocation failed");
        return nullptr;
    }
    h->prof.nk = MSL_ORB_NKERNELS;
    if (build_geometry(h, max_width, max_height) != MSL_OK) return nullptr;
    return guard.release();
    } MSL_ABI_CATCH_PTR
}

void msl_orb_destroy(msl_orb *h) noexcept {
    try {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->prof.destroy();
    free_geometry(h);
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->h_err) (void)hipHostFree(h->h_err);
    if (h->h_pinIn) (void)hipHostFree(h->h_pinIn);
    if (h->h_pinOut) (void)hipHostFree(h->h_pinOut);
    if (h->d_depthIn) (void)hipFree(h->d_depthIn);
    if (h->d_unXY) (void)hipFree(h->d_unXY);
    if (h->d_depthOut) (void)hipFree(h->d_depthOut);
    if (h->d_uRight) (void)hipFree(h->d_uRight);
    if (h->d_gridCell) (void)hipFree(h->d_gridCell);
    if (h->sideStream) { (void)hipStreamSynchronize(h->sideStream); (void)hipStreamDestroy(h->sideStream); }
    if (h->evFork) (void)hipEventDestroy(h->evFork);
    if (h->evJoin) (void)hipEventDestroy(h->evJoin);
    if (h->stream && h->ownStream) (void)hipStreamDestroy(h->stream);
    delete h;
    } MSL_ABI_CATCH_VOID
}

int msl_orb_scale_tables(const msl_orb *h, float *sf, float *isf, float *s2, float *is2) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    for (int i = 0; i < h->nlevels; i++) {
        if (sf) sf[i] = h->scale[i];
        if (isf) isf[i] = h->invScale[i];
        if (s2) s2[i] = h->sigma2[i];
        if (is2) is2[i] = h->invSigma2[i];
    }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_orb_features_per_level(const msl_orb *h, int32_t *out) noexcept {
    try {
    if (!h || !out) return MSL_ERR_INVALID;
    for (int i = 0; i < h->nlevels; i++) out[i] = h->perLevel[i];
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_orb_capacity(const msl_orb *h) noexcept { try { return h ? h->outCap : MSL_ERR_INVALID; } MSL_ABI_CATCH_INT }
int msl_orb_levels(const msl_orb *h) noexcept { try { return h ? h->nlevels : MSL_ERR_INVALID; } MSL_ABI_CATCH_INT }

int msl_orb_set_stream(msl_orb *h, void *hip_stream) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->ownStream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream; h->ownStream = false;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_wait_event(msl_orb *h, void *hip_event) noexcept {
    try {
    if (!h || !hip_event) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamWaitEvent(h->stream, (hipEvent_t)hip_event, 0));   // (the side stream forks from this one inside every call)
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_sync(msl_orb *h) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    return check_device_error(h);
    } MSL_ABI_CATCH_INT
}

int msl_orb_extract_batch(msl_orb *h, const uint8_t *gray, int n_frames, int width, int height, size_t row_stride,
                          size_t frame_stride, msl_mem in_mem, msl_keypoint *kps, uint8_t *desc32, int cap,
                          int32_t *n_out, msl_mem out_mem) noexcept {
    try {
    if (!h || !n_out || n_frames < 0) { set_error("msl_orb_extract_batch: invalid argument"); return MSL_ERR_INVALID; }
    if (n_frames == 0) return MSL_OK;
    if (!gray || width == 0 || height == 0) {  // empty image: silent return (src/ORBextractor.cc:815-816)
        if (out_mem == MSL_MEM_HOST) for (int f = 0; f < n_frames; f++) n_out[f] = 0;
        else { MSL_HIP_TRY(hipSetDevice(h->device)); MSL_HIP_TRY(hipMemsetAsync(n_out, 0, sizeof(int) * n_frames, h->stream)); }
        return MSL_OK;
    }
    if (n_frames > h->maxBatch || width > h->maxW || height > h->maxH || row_stride < (size_t)width || !kps || !desc32) {
        set_error("msl_orb_extract_batch: frame %dx%d x%d exceeds the handle's limits (%dx%d x%d) or bad pointers", width,
                  height, n_frames, h->maxW, h->maxH, h->maxBatch);
        return MSL_ERR_INVALID;
    }
    const int outCap = h->outCap;
    if (cap < outCap) { set_error("msl_orb_extract_batch: cap %d < required %d", cap, outCap); return MSL_ERR_CAPACITY; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = build_geometry(h, width, height);
    if (rc != MSL_OK) return rc;
    const uint8_t *d_gray = gray; size_t rs = row_stride, fs = frame_stride;
    if (in_mem == MSL_MEM_HOST) {
        h->prof.begin(KID_COPY, h->stream);
        if (row_stride == (size_t)width && h->inPitch == (size_t)width && frame_stride == (size_t)width * height) {
            // tightly packed frames (the streaming case): one copy for the whole batch
            MSL_HIP_TRY(hipMemcpyAsync(h->d_in, gray, (size_t)width * height * n_frames, hipMemcpyHostToDevice, h->stream));
        } else {
            for (int f = 0; f < n_frames; f++)
                MSL_HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * h->inPitch * height, h->inPitch, gray + (size_t)f * frame_stride,
                                             row_stride, width, height, hipMemcpyHostToDevice, h->stream));
        }
        h->prof.end(h->stream);
        d_gray = h->d_in; rs = h->inPitch; fs = h->inPitch * height;
    }
    if (out_mem == MSL_MEM_DEVICE && cap == outCap) {
        return launch_pipeline(h, d_gray, rs, fs, n_frames, kps, desc32, n_out);
    }
    rc = launch_pipeline(h, d_gray, rs, fs, n_frames, h->d_kps, h->d_desc, h->d_nout);
    if (rc != MSL_OK) return rc;
    const hipMemcpyKind kind = out_mem == MSL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    MSL_HIP_TRY(hipMemcpyAsync(n_out, h->d_nout, sizeof(int) * n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(kps, sizeof(msl_keypoint) * cap, h->d_kps, sizeof(msl_keypoint) * outCap,
                                 sizeof(msl_keypoint) * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(desc32, (size_t)32 * cap, h->d_desc, (size_t)32 * outCap, (size_t)32 * outCap, n_frames, kind,
                                 h->stream));
    if (out_mem == MSL_MEM_HOST) return check_device_error(h);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

// The reference's call pattern: one frame per call, host buffers in and out, the result needed before the caller goes on (src/Frame.cc:100,
// 175-177).  Latency is everything here, so this form avoids every pageable-memory transfer: the frame goes through a pinned staging buffer
// (one CPU copy, one DMA), the counts, keypoints and descriptors come back as ONE copy of the output block into pinned memory next to the
// error word, and a single stream synchronisation ends the call (the batch form issues four device-to-host copies into pageable memory).
static int extract_one_host(msl_orb *h, const uint8_t *gray, int width, int height, size_t stride, msl_keypoint *kps, uint8_t *desc32, int cap, int *n_out) {
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = build_geometry(h, width, height);
    if (rc != MSL_OK) return rc;
    const int outCap = h->outCap;
    if (cap < outCap) { set_error("msl_orb_extract: cap %d < required %d", cap, outCap); return MSL_ERR_CAPACITY; }
    const size_t inBytes = h->inPitch * (size_t)height;
    // frame 0's share of the output block: counts (all B of them: a few bytes), its keypoints and -- B == 1 only -- its descriptors contiguous
    const size_t outBytes = h->maxBatch == 1 ? h->outBlockBytes : 0;
    if (inBytes > h->pinInBytes) {
        if (h->h_pinIn) (void)hipHostFree(h->h_pinIn);
        h->h_pinIn = nullptr; h->pinInBytes = 0;
        MSL_HIP_TRY(hipHostMalloc(&h->h_pinIn, inBytes));
        h->pinInBytes = inBytes;
    }
    const size_t needOut = h->outKpsOff + sizeof(msl_keypoint) * (size_t)outCap + (size_t)32 * outCap + 256;
    if (needOut > h->pinOutBytes) {
        if (h->h_pinOut) (void)hipHostFree(h->h_pinOut);
        h->h_pinOut = nullptr; h->pinOutBytes = 0;
        MSL_HIP_TRY(hipHostMalloc(&h->h_pinOut, needOut));
        h->pinOutBytes = needOut;
    }
    hipStream_t s = h->stream;
    h->prof.begin(KID_COPY, s);
    // (one copy: splitting it so that the DMA of the first half overlaps the CPU copy of the second measured 6 us SLOWER -- an enqueue costs more than it hides)
    if (stride == h->inPitch) memcpy(h->h_pinIn, gray, stride * (size_t)(height - 1) + width);
    else for (int y = 0; y < height; y++) memcpy(h->h_pinIn + (size_t)y * h->inPitch, gray + (size_t)y * stride, (size_t)width);
    MSL_HIP_TRY(hipMemcpyAsync(h->d_in, h->h_pinIn, inBytes, hipMemcpyHostToDevice, s));
    h->prof.end(s);
    rc = launch_pipeline(h, h->d_in, h->inPitch, inBytes, 1, h->d_kps, h->d_desc, h->d_nout);
    if (rc != MSL_OK) return rc;
    const size_t kpsBytes = sizeof(msl_keypoint) * (size_t)outCap, descBytes = (size_t)32 * outCap;
    uint8_t *hk = h->h_pinOut + h->outKpsOff, *hd = hk + ((kpsBytes + 255) & ~(size_t)255);
    if (outBytes) {   // a one-frame handle: counts | keypoints | descriptors are one contiguous block
        MSL_HIP_TRY(hipMemcpyAsync(h->h_pinOut, h->d_outBlock, outBytes, hipMemcpyDeviceToHost, s));
        hd = h->h_pinOut + h->outDescOff;
    } else {
        MSL_HIP_TRY(hipMemcpyAsync(h->h_pinOut, h->d_nout, sizeof(int), hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipMemcpyAsync(hk, h->d_kps, kpsBytes, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipMemcpyAsync(hd, h->d_desc, descBytes, hipMemcpyDeviceToHost, s));
    }
    rc = check_device_error(h);   // error word into pinned memory, then the call's only synchronisation
    if (rc != MSL_OK) { *n_out = 0; return rc; }
    const int n = *reinterpret_cast<const int *>(h->h_pinOut);
    memcpy(kps, hk, sizeof(msl_keypoint) * (size_t)n);
    memcpy(desc32, hd, (size_t)32 * n);
    *n_out = n;
    return MSL_OK;
}

int msl_orb_extract(msl_orb *h, const uint8_t *gray, int width, int height, size_t stride, msl_keypoint *kps,
                    uint8_t *desc32, int cap, int *n_out) noexcept {
    try {
    if (!n_out) { set_error("msl_orb_extract: n_out is NULL"); return MSL_ERR_INVALID; }
    if (h && gray && width > 0 && height > 0 && width <= h->maxW && height <= h->maxH && stride >= (size_t)width && kps && desc32)
        return extract_one_host(h, gray, width, height, stride, kps, desc32, cap, n_out);
    int32_t n = 0;
    const int rc = msl_orb_extract_batch(h, gray, 1, width, height, stride, stride * (size_t)height, MSL_MEM_HOST, kps, desc32,
                                         cap, &n, MSL_MEM_HOST);
    *n_out = n;
    return rc;
    } MSL_ABI_CATCH_INT
}

// host twin of the device undistortion (same expression order), used by ComputeImageBounds only
static void undistort_point_host(const msl_frame_params &p, float xin, float yin, float *xo, float *yo) {
    const double fx = p.fx, fy = p.fy, cx = p.cx, cy = p.cy, ifx = 1. / fx, ify = 1. / fy;
    const double k0 = p.k1, k1 = p.k2, k2 = p.p1, k3 = p.p2, k4 = p.k3, kz = 0.0;
    double x = xin, y = yin;
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = (1 + ((kz * r2 + kz) * r2 + kz) * r2) / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        const double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x) + kz * r2 + kz * r2 * r2;
        const double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y + kz * r2 + kz * r2 * r2;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = fx * x + 0.0 * y + cx, yy = 0.0 * x + fy * y + cy, ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

int msl_frame_image_bounds(msl_frame_params *p, int width, int height) noexcept {
    try {   // ComputeImageBounds, src/Frame.cc:465-494
    if (!p || width < 1 || height < 1 || p->fx == 0 || p->fy == 0) { set_error("msl_frame_image_bounds: invalid argument"); return MSL_ERR_INVALID; }
    if (p->k1 != 0.0) {
        const float c[4][2] = {{0.f, 0.f}, {(float)width, 0.f}, {0.f, (float)height}, {(float)width, (float)height}};
        float u[4][2];
        for (int i = 0; i < 4; i++) undistort_point_host(*p, c[i][0], c[i][1], &u[i][0], &u[i][1]);
        p->minX = std::min(u[0][0], u[2][0]); p->maxX = std::max(u[1][0], u[3][0]);
        p->minY = std::min(u[0][1], u[1][1]); p->maxY = std::max(u[2][1], u[3][1]);
    } else {
        p->minX = 0.0f; p->maxX = (float)width; p->minY = 0.0f; p->maxY = (float)height;
    }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_extract_frame_batch(msl_orb *h, const uint8_t *gray, const float *depth, int n_frames, int width, int height,
                                size_t gray_row_stride, size_t gray_frame_stride, size_t depth_row_stride, size_t depth_frame_stride,
                                msl_mem in_mem, const msl_frame_params *params, msl_keypoint *kps, uint8_t *desc32, float *kps_un_xy,
                                float *depth_out, float *uright_out, int32_t *grid_cell, int cap, int32_t *n_out, msl_mem out_mem) noexcept {
    try {
    if (!h || !gray || !depth || !params || !kps || !desc32 || !kps_un_xy || !depth_out || !uright_out || !grid_cell || !n_out ||
        n_frames < 1 || n_frames > h->maxBatch || width < 1 || height < 1 || width > h->maxW || height > h->maxH ||
        gray_row_stride < (size_t)width || depth_row_stride < (size_t)width * 4 || (depth_row_stride & 3) ||
        !(params->maxX > params->minX) || !(params->maxY > params->minY) || params->fx == 0 || params->fy == 0) {
        set_error("msl_orb_extract_frame_batch: invalid argument (call msl_frame_image_bounds first?)");
        return MSL_ERR_INVALID;
    }
    const int outCap = h->outCap;
    if (cap < outCap) { set_error("msl_orb_extract_frame_batch: cap %d < required %d", cap, outCap); return MSL_ERR_CAPACITY; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = build_geometry(h, width, height);
    if (rc != MSL_OK) return rc;
    const size_t B = (size_t)h->maxBatch;
    if (!h->frameBufs) {
        MSL_HIP_TRY(hipMalloc(&h->d_unXY, sizeof(float) * 2 * outCap * B)); MSL_HIP_TRY(hipMalloc(&h->d_depthOut, sizeof(float) * outCap * B));
        MSL_HIP_TRY(hipMalloc(&h->d_uRight, sizeof(float) * outCap * B)); MSL_HIP_TRY(hipMalloc(&h->d_gridCell, sizeof(int) * outCap * B));
        h->frameBufs = true;
    }
    const uint8_t *d_gray = gray; size_t rs = gray_row_stride, fs = gray_frame_stride;
    FrameEpilogue ep{};
    ep.fp = *params; ep.depth = depth; ep.depthRowStride = depth_row_stride; ep.depthFrameStride = depth_frame_stride;
    if (in_mem == MSL_MEM_HOST) {
        const size_t dpitch = (size_t)width * 4, dframe = dpitch * height;
        if (dframe * B > h->depthInCap) {
            MSL_HIP_TRY(hipStreamSynchronize(h->stream));
            if (h->d_depthIn) (void)hipFree(h->d_depthIn);
            h->d_depthIn = nullptr; h->depthInCap = 0;
            MSL_HIP_TRY(hipMalloc(&h->d_depthIn, dframe * B));
            h->depthInCap = dframe * B;
        }
        for (int f = 0; f < n_frames; f++) {
            MSL_HIP_TRY(hipMemcpy2DAsync(h->d_in + (size_t)f * h->inPitch * height, h->inPitch, gray + (size_t)f * gray_frame_stride, gray_row_stride,
                                         width, height, hipMemcpyHostToDevice, h->stream));
            MSL_HIP_TRY(hipMemcpy2DAsync((uint8_t *)h->d_depthIn + (size_t)f * dframe, dpitch, (const uint8_t *)depth + (size_t)f * depth_frame_stride,
                                         depth_row_stride, dpitch, height, hipMemcpyHostToDevice, h->stream));
        }
        d_gray = h->d_in; rs = h->inPitch; fs = h->inPitch * height;
        ep.depth = h->d_depthIn; ep.depthRowStride = dpitch; ep.depthFrameStride = dframe;
    }
    const bool direct = out_mem == MSL_MEM_DEVICE && cap == outCap;
    ep.unXY = direct ? kps_un_xy : h->d_unXY; ep.depthOut = direct ? depth_out : h->d_depthOut;
    ep.uRight = direct ? uright_out : h->d_uRight; ep.gridCell = direct ? grid_cell : h->d_gridCell;
    if (direct) return launch_pipeline(h, d_gray, rs, fs, n_frames, kps, desc32, n_out, &ep);
    rc = launch_pipeline(h, d_gray, rs, fs, n_frames, h->d_kps, h->d_desc, h->d_nout, &ep);
    if (rc != MSL_OK) return rc;
    const hipMemcpyKind kind = out_mem == MSL_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
    MSL_HIP_TRY(hipMemcpyAsync(n_out, h->d_nout, sizeof(int) * n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(kps, sizeof(msl_keypoint) * cap, h->d_kps, sizeof(msl_keypoint) * outCap, sizeof(msl_keypoint) * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(desc32, (size_t)32 * cap, h->d_desc, (size_t)32 * outCap, (size_t)32 * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(kps_un_xy, sizeof(float) * 2 * cap, h->d_unXY, sizeof(float) * 2 * outCap, sizeof(float) * 2 * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(depth_out, sizeof(float) * cap, h->d_depthOut, sizeof(float) * outCap, sizeof(float) * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(uright_out, sizeof(float) * cap, h->d_uRight, sizeof(float) * outCap, sizeof(float) * outCap, n_frames, kind, h->stream));
    MSL_HIP_TRY(hipMemcpy2DAsync(grid_cell, sizeof(int) * cap, h->d_gridCell, sizeof(int) * outCap, sizeof(int) * outCap, n_frames, kind, h->stream));
    if (out_mem == MSL_MEM_HOST) return check_device_error(h);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_debug_stamps(msl_orb *h, uint64_t *out, int n) noexcept {
    try {
    if (!h || !out || n < 0 || n > 200) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    MSL_HIP_TRY(hipMemcpy(out, reinterpret_cast<unsigned char *>(h->d_err) + 128, sizeof(uint64_t) * n, hipMemcpyDeviceToHost));
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_debug_level_size(const msl_orb *h, int level, int *w, int *h_out) noexcept {
    try {
    if (!h || level < 0 || level >= h->nlevels) return MSL_ERR_INVALID;
    *w = h->dev.lv[level].w; *h_out = h->dev.lv[level].h;
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_debug_level(msl_orb *h, int frame, int level, int blurred, uint8_t *out) noexcept {
    try {
    if (!h || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastFrames) return MSL_ERR_INVALID;
    if (level == 0 && !blurred) { set_error("level 0 is the caller's image"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    const LevelDev &G = h->dev.lv[level];
    const uint8_t *src = blurred ? h->d_blur + (size_t)frame * h->dev.blurStride + G.boff
                                 : h->d_pyr + (size_t)frame * h->dev.pyrStride + G.off;
    MSL_HIP_TRY(hipMemcpy2D(out, G.w, src, G.pitch, G.w, G.h, hipMemcpyDeviceToHost));
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_debug_candidates(msl_orb *h, int frame, int level, int32_t *xys, int cap, int *n_out) noexcept {
    try {
    if (!h || level < 0 || level >= h->nlevels || frame < 0 || frame >= h->lastFrames) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    int n = 0;
    MSL_HIP_TRY(hipMemcpy(&n, h->d_ncand + frame * h->nlevels + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) return MSL_ERR_CAPACITY;
    std::vector<uint32_t> k(n);
    if (n) MSL_HIP_TRY(hipMemcpy(k.data(), h->d_keys + (size_t)frame * h->dev.keysPerFrame + h->dev.lv[level].keyBase,
                                 sizeof(uint32_t) * n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        xys[3 * i] = (int)(k[i] & 0xFFF) + 16; xys[3 * i + 1] = (int)((k[i] >> 12) & 0xFFF) + 16; xys[3 * i + 2] = (int)(k[i] >> 24);
    }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}

int msl_orb_profile_enable(msl_orb *h, int on) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    h->prof.set_mode(on);
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
int msl_orb_profile_read(msl_orb *h, float *ms, int32_t *launches) noexcept {
    try {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    MSL_HIP_TRY(hipStreamSynchronize(h->stream));
    h->prof.drain();
    for (int i = 0; i < MSL_ORB_NKERNELS; i++) { if (ms) ms[i] = h->prof.ms[i]; if (launches) launches[i] = h->prof.launches[i]; }
    return MSL_OK;
    } MSL_ABI_CATCH_INT
}
const char *msl_orb_kernel_name(int k) noexcept { try { return (k >= 0 && k < MSL_ORB_NKERNELS) ? kKernelNames[k] : ""; } MSL_ABI_CATCH_PTR }

}  // extern "C"
