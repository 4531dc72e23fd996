// msl_surfel.hip -- superpixel surfel fusion for gfx950 (MI355X): kernels + C ABI.
//
// Replaces SurfelFusion (reference src/SurfelFusion.cpp) and the slot refill / tail compaction of
// SurfelMapping::fuseMap (src/SurfelMapping.cpp:353-392).
//
// MI355X-first structure: generateSuperPixels() of a keyframe depends only on that keyframe's images,
// never on the map, so it is FRAME-BATCHED (blockIdx.y = keyframe slot) on a "pre" stream; only the
// map stage (fuse -> new surfels -> compaction) is sequential per keyframe, on the "map" stream, and the
// two overlap across batches (double-buffered slot sets).
//
//   pre stream, one launch per batch of F keyframes:
//     kb_seed_init                        one thread per 8x8 superpixel seed                  (:528-584)
//     3 x { kb_assign                     one thread per pixel: argmin over <= 9 seeds        (:333-415)
//           [kb_prop_lds,                 raster-order `stable` semantics as a min-fixpoint   (App. B.7.1)
//            kb_commit_px]                  over a compact worklist of the only pixels that can extend a chain (one launch, LDS)
//           kb_update_seeds               16 lanes per seed: ordered window gather, Huber mean (:428-515)
//           kb_commit_seeds }             chunk-abort (`return`) semantics: restore-only      (App. B.7.2)
//     kb_seed_plane                       16 lanes per seed: back-projection, pixel normals, Huber plane
//                                         fit with FP64 4x4 normal equations                  (:91-165, :597-773)
//   map stream, per keyframe (two launches):
//     k_fuse                              live surfels, hot/cold record map resident in HBM   (:167-283)
//     k_compact                           deleted-slot list (handed over by k_fuse / sub-block scan), ordered emission of un-fused seeds (:285-331),
//                                         deleted-slot refill + tail compaction               (SurfelMapping.cpp:366-391)
//
// HBM-bound integer/float streaming; no MFMA.  Every float expression keeps the reference's evaluation
// order and float/double promotions; compiled with -ffp-contract=off.

#include "msl_common.h"

#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <fstream>
#include <type_traits>
#include <vector>

using namespace msl;

namespace {

constexpr int SP = 8;
constexpr int NCHUNK = 10;  // THREAD_NUM, include/SurfelFusion.h:34
constexpr double MAX_ANGLE_COS = 0.1, HUBER_RANGE = 0.4, MIN_TOLERATE_DIFF = 0.1;   // (BASELINE 0.5 and DISPARITY_ERROR 4.0 appear as exact float factors in k_fuse)
constexpr unsigned T_INF = 0xFFFFFFFFu;
constexpr int PROP_ROUNDS = 6;          // worklist relaxation rounds before the single-workgroup finisher
constexpr unsigned short IDX_NONE = 0xFFFF, IDX_PLANE = 0xFFFE;
constexpr int LIST_D = 256;             // fastest compaction path: k_fuse hands over the few deleted slots directly
constexpr int SCAN_ITEMS = 1024;        // surfels per workgroup and pass in the map-maintenance kernels (k_select_*)
constexpr int SUB_ITEMS = 256;          // sub-block: the surfels one k_fuse wave owns = granularity of the deleted / updated partials

// Device-resident surfel map, split hot/cold: the fuse kernel streams only the 20-byte hot records (what decides a
// surfel's fate for the ~90 % that leave early) and touches the 36-byte cold record of the few it updates; an update
// writes two contiguous records (3-4 cache lines) instead of 14 scattered 4-byte fields.
struct HotRec { float px, py, pz; int updateTimes, lastUpdate; };                       // 20 B
// 32 bytes, 32-byte aligned: a fused surfel touches exactly one 32-byte sector of its cold record (the 36-byte record of round 2
// straddled sectors and cache lines).  r, g, b always come from a cv::Vec3b (src/SurfelFusion.cpp:484, 551), so they travel as three
// bytes; a record whose ints do not fit a byte (only possible for maps uploaded by the caller) sets COLD_WIDE and keeps the exact ints in
// rgbWide[3 i ..] -- every accessor below honours it, so upload -> download stays the identity for arbitrary values.
struct alignas(32) ColdRec { float nx, ny, nz, size, color, weight; unsigned rgbf; unsigned _spare; };
constexpr unsigned COLD_WIDE = 1u << 24;
struct MapSoA {
    HotRec *hot;
    ColdRec *cold;
    int *rgbWide;          // [cap][3]
    long long *wideFlag;   // ctr[13]: set once any COLD_WIDE record has been stored (the map copies then carry rgbWide along)
};
__host__ __device__ inline bool rgb_fits(int r, int g, int b) { return ((unsigned)r | (unsigned)g | (unsigned)b) < 256u; }
__host__ __device__ inline unsigned rgb_pack(int r, int g, int b) { return (unsigned)r | ((unsigned)g << 8) | ((unsigned)b << 16); }

// Per-keyframe parameters of one slot (device memory, uploaded per batch).
// Image pointers travel through memory, so the compiler only knows them as generic pointers and would emit FLAT loads
// (which also count against lgkmcnt and so serialise with LDS / scalar traffic); the accessors restore the global
// address space.
template <typename T> using gptr = const T __attribute__((address_space(1))) *;
struct FrameDev {
    const uint8_t *gray; const float *depth; const int32_t *member;
    float pose[16], invPose[16];
    int ref, _pad;
    __device__ __forceinline__ gptr<uint8_t> grayG() const { return (gptr<uint8_t>)gray; }
    __device__ __forceinline__ gptr<float> depthG() const { return (gptr<float>)depth; }
    __device__ __forceinline__ gptr<int32_t> memberG() const { return (gptr<int32_t>)member; }
};

// What the pixel pass needs of a seed, 32 bytes, so that a wave fetches a candidate with one scalar load (and the two candidates of a lattice row,
// neighbours in memory, with a single 64-byte one): written wherever a seed's x / y / meanDepth / meanIntensity change.
struct alignas(32) AssignRec {
    float x, y, meanIntensity;
    unsigned stable;               // the seed's stable flag as the next pixel pass finds it (t(s) == T_INF)
    double invDepth;               // meanDepth > 0: 1.0 / (double)meanDepth, the value calculateCost's divide gives (:349) -- never negative;
                                   // otherwise -1.0, i.e. the sign doubles as the seed's "has depth" test (:348) and the record needs no meanDepth
    unsigned long long _pad;
};
__device__ __forceinline__ AssignRec assign_rec(const msl_seed &s) {
    AssignRec a;
    a.x = s.x; a.y = s.y; a.meanIntensity = s.meanIntensity; a.stable = s.stable ? 1u : 0u;
    a.invDepth = s.meanDepth > 0 ? 1.0 / (double)s.meanDepth : -1.0; a._pad = 0;
    return a;
}

struct SfDev {
    int W, H, spW, spH, nseeds, npx;   // npx = W * H (the flat pixel index range of the reference); spW = W / 8, spH = H / 8 (truncated, :29-38)
    float fx, fy, cx, cy, fuseFar, fuseNear;
    unsigned long long gstride, gbytes, dstride, mstride;   // gray bytes, depth floats, member ints
    const FrameDev *frames;      // [slots]
    msl_seed *seeds, *seedsTmp;  // [slots][nseeds]
    msl_surfel *cand;            // [slots][nseeds] world-frame surfel a seed would spawn
    uint8_t *candOk;             // [slots][nseeds]
    uint8_t *fused;              // [slots][nseeds] seed consumed by a fusion
    uint2 *tex;                  // [slots][npx] {depth bits, final superpixel index} of every pixel: k_fuse's ONE gather per in-view surfel
    float4 *fuseRec;             // [slots][nseeds][3] what k_fuse needs of a seed, per-seed terms of :236-277 evaluated once (FuseRec below)
    unsigned short *index, *amap;  // [slots][npx]
    unsigned *tmin;              // [slots][nseeds]
    AssignRec *arec;             // [slots][nseeds] (+ one record of padding at either end) what kb_assign reads of a seed
    float *pxInv;                // [slots][npx] (float)(1.0 / (double)depth) of every pixel (0 when depth <= 0.01): pass 0 writes, passes 1-2 read
    unsigned *wl;                // [slots][npx] relaxation worklist: pixels on a stable seed that pick a different seed
    unsigned *wlCount;           // [slots]
    int *chunkAbort;             // [slots][2][16]
    int *changed;                // [slots][8]
    MapSoA map;
    unsigned long long cap;
    // ctr[0]=n_live  [1]=K new (last)  [2]=D deleted (last)  [3]=updated (last)  [4]=n before (last)  [5]=err  [6]=n after (last)  [7]=tail fallback flag
    // ctr[8..12] = running totals: new, deleted, updated, keyframes, live-before
    long long *ctr;
    msl_surfel *newSurfels;
    unsigned *blockSums, *blockUpd, *delList, *srcOf;
    unsigned *tickets;           // [2] hand-off counters (k_fuse, k_compact)
    unsigned *delU;              // [LIST_D] unordered list of the slots k_fuse found deleted (fast path of k_compact)
    unsigned *delUCount;         // number of slots appended (may exceed LIST_D: then the list is incomplete and unused)
    const float *colX, *rowY;    // [W+1], [H+1]: (u - cx) / fx and (v - cy) / fy of the integer pixel coordinates (back_project)
    int pxStride;                // per-slot stride of the per-pixel arrays: npx rounded up to 64 (16-byte vector accesses stay aligned); last, so that
                                 // the kernel-argument offsets of everything above are those the map-stage kernels were tuned with
    // Overlap of compaction j with fusion j + 1 (run_batch).  The per-keyframe hand-over data (delU, delUCount, blockSums, blockUpd above) rotate
    // through three slots, j % 3; the fields below point at what keyframe j - 1 left and at where live counts are published.
    int fuseMode;                      // k_fuse: 0 = every sub-block; 1 = only the sub-blocks compaction j - 1 cannot touch; 2 = only the others
    const unsigned *prevBlockSums;     // deleted slots per sub-block of keyframe j - 1 (a sub-block with any gets a new surfel or a tail element)
    const unsigned *prevDelUCount;     // D of keyframe j - 1: the tail moves of compaction j - 1 start at n - D at the earliest
    const long long *nPubPrev;         // live count before compaction j - 1 (= after compaction j - 2)
    long long *nPubOut;                // k_compact: the live count after this keyframe; k_fuse mode 0: the live count it found (for the next keyframe's mode 1)
    unsigned *resetDelUCount;          // k_compact: the hand-over count of the slot keyframe j + 2 will use
    // merged launches (k_fuse_merged: workgroup 0 compacts keyframe j - 1 while the others fuse keyframe j)
    const unsigned *prevDelU;          // hand-over list of keyframe j - 1
    const unsigned *prevBlockUpd;      // updated surfels per sub-block of keyframe j - 1
    long long *nPubCompact;            // where the compaction of keyframe j - 1 publishes the live count it leaves
    unsigned *resetDelUCountCompact;   // the hand-over count that compaction re-arms (keyframe j + 1's slot)
    unsigned *doneFlag;                // set to `epoch` (write-through) when the compaction is complete; the dependent waves poll it
    unsigned epoch;
    int prevSlot;                      // superpixel slot of keyframe j - 1 (candidates, fused flags)
    unsigned *updCtr;                  // [64] updated-surfel counts of this keyframe, hashed by sub-block (merged batches only; NULL otherwise)
    unsigned *prevUpdCtr;              // the same of keyframe j - 1: summed and cleared by its compaction
};

__device__ __forceinline__ int seed_chunk(int seedI, int nseeds) {   // THREAD_NUM partition of :430-434
    const int step = nseeds / NCHUNK;
    if (step == 0) return NCHUNK - 1;
    const int c = seedI / step;
    return c > NCHUNK - 1 ? NCHUNK - 1 : c;
}
__device__ __forceinline__ uint8_t gray_at(const SfDev &P, const FrameDev &F, int y, int x) { return F.grayG()[(size_t)y * P.gstride + x]; }
__device__ __forceinline__ float depth_at(const SfDev &P, const FrameDev &F, int y, int x) { return F.depthG()[(size_t)y * P.dstride + x]; }
__device__ __forceinline__ void vec3b(const SfDev &P, const FrameDev &F, float row, float col, int &r, int &g, int &b) {
    const unsigned long long off = (unsigned long long)(int)row * P.gstride + 3ull * (unsigned long long)(int)col;
    r = off < P.gbytes ? F.grayG()[off] : 0;
    g = off + 1 < P.gbytes ? F.grayG()[off + 1] : 0;
    b = off + 2 < P.gbytes ? F.grayG()[off + 2] : 0;
}
__device__ __forceinline__ void back_project(const SfDev &P, float u, float v, float d, float &x, float &y, float &z) {
    x = (u - P.cx) / P.fx * d;   // src/SurfelFusion.cpp:80-85 (float expression, stored to double there)
    y = (v - P.cy) / P.fy * d;
    z = d;
}
// std::min(1.0 / depth / depth, 1.0) (:87-89) is `(1.0 < a) ? 1.0 : a`: a NaN depth (a seed whose plane fit produced NaN) gives NaN, where
// fmin() would give 1.0 -- found by the furnished-room parity tests of round 4.
__device__ __forceinline__ float get_weight(float d) { const double a = 1.0 / (double)d / (double)d; return (float)(1.0 < a ? 1.0 : a); }
__device__ __forceinline__ void mul4(const float *m, float v0, float v1, float v2, float v3, float out[4]) {
#pragma unroll
    for (int r = 0; r < 4; r++) out[r] = ((m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2) + m[12 + r] * v3;
}
__device__ __forceinline__ void mul3(const float *m, float v0, float v1, float v2, float out[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++) out[r] = (m[r] * v0 + m[4 + r] * v1) + m[8 + r] * v2;
}

// adjugate / determinant inverse of a 4x4 (column-major); pins Eigen's Matrix4::inverse()
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

// Correctly rounded x / 100.0 (x >= 0 finite) without the ~35-instruction f64 divide: two Markstein steps with
// y = RN(1/100).  q1 is a faithful quotient (error < 1 ulp), so the final fused correction rounds to RN(x/100)
// (Markstein's theorem; 100 = 1.5625 * 2^6 is not an all-ones significand).  Checked against true division on
// the GPU by tests/test_surfel_gpu.py::test_div100_exact.
__device__ __forceinline__ double div100_exact(double x) {
    const double y = 0.01;                       // RN(1/100)
    const double q0 = x * y;
    const double q1 = fma(fma(-q0, 100.0, x), y, q0);
    return fma(fma(-q1, 100.0, x), y, q1);
}

// Strictly sequential (left-to-right) float sums over 16-byte aligned LDS arrays; wide LDS reads are issued
// ahead of the dependent add chain so the chain runs at VALU latency instead of LDS latency.
__device__ __forceinline__ float seq_sum_f32(const float *a, int n, float s) {
    int p = 0;
    for (; p + 8 <= n; p += 8) {
        const float4 u = *reinterpret_cast<const float4 *>(a + p), v = *reinterpret_cast<const float4 *>(a + p + 4);
        s += u.x; s += u.y; s += u.z; s += u.w; s += v.x; s += v.y; s += v.z; s += v.w;
    }
    for (; p < n; p++) s += a[p];
    return s;
}
// Huber/Newton numerator (:494-503) in list order: finite terms are 2*residual (a float add; identical to the double
// add rounded to float), +-inf marks a tail element whose contribution is the DOUBLE constant +-HUBER_RANGE.
__device__ __forceinline__ float huber_term_add(float s, float t) {
    return __builtin_isinf(t) ? (float)((double)s + (t > 0 ? HUBER_RANGE : -1 * HUBER_RANGE)) : s + t;
}
__device__ __forceinline__ float seq_sum_huber(const float *t, int n, float s) {
    int e = 0;
    for (; e + 8 <= n; e += 8) {
        const float4 u = *reinterpret_cast<const float4 *>(t + e), v = *reinterpret_cast<const float4 *>(t + e + 4);
        s = huber_term_add(s, u.x); s = huber_term_add(s, u.y); s = huber_term_add(s, u.z); s = huber_term_add(s, u.w);
        s = huber_term_add(s, v.x); s = huber_term_add(s, v.y); s = huber_term_add(s, v.z); s = huber_term_add(s, v.w);
    }
    for (; e < n; e++) s = huber_term_add(s, t[e]);
    return s;
}

// XCD-aware block -> (keyframe slot, block-in-frame) mapping for the frame-batched kernels.  Workgroup `lin` runs on
// XCD lin % 8 (observed dispatch order; used for speed only), and each XCD has its own 4 MB L2: give every XCD whole
// keyframes (slot = xcd, xcd + 8, ...) so that one keyframe's images, index map and seeds (~3 MB) stay L2 resident
// while its workgroups stream through, instead of all 8 L2s thrashing over the whole batch.
// Launch with a 1-D grid of 8 * ceil(n_slots / 8) * blocksPerFrame workgroups.
__device__ __forceinline__ bool xcd_slot(int blocksPerFrame, int nSlots, int &slot, int &blk) {
    const unsigned lin = blockIdx.x;
    const unsigned j = lin >> 3;
    slot = (int)(lin & 7u) + 8 * (int)(j / (unsigned)blocksPerFrame);
    blk = (int)(j % (unsigned)blocksPerFrame);
    return slot < nSlots;
}
__host__ inline unsigned xcd_grid(int blocksPerFrame, int nSlots) { return 8u * (unsigned)((nSlots + 7) / 8) * (unsigned)blocksPerFrame; }

// =============================================================================================
// Frame-batched superpixel stage
// =============================================================================================
__global__ __launch_bounds__(256) void kb_seed_init(SfDev P) {
    const int slot = blockIdx.y;
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    if (seedI == 0) P.wlCount[slot] = 0;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    int imageX = spX * SP + SP / 2, imageY = spY * SP + SP / 2;
    imageX = imageX < (P.W - 1) ? imageX : (P.W - 1);
    imageY = imageY < (P.H - 1) ? imageY : (P.H - 1);
    msl_seed s;
    memset(&s, 0, sizeof(s));
    P.fused[(size_t)slot * P.nseeds + seedI] = 0;
    if (F.memberG()[(size_t)(imageY / 2) * P.mstride + imageX / 2] != -1) {
        P.seeds[(size_t)slot * P.nseeds + seedI] = s; P.arec[(size_t)slot * P.nseeds + seedI] = assign_rec(s);
        return;
    }
    s.use = 1;
    s.x = (float)imageX; s.y = (float)imageY;
    vec3b(P, F, (float)imageY, (float)imageX, s.r, s.g, s.b);
    s.meanIntensity = gray_at(P, F, imageY, imageX);
    s.meanDepth = depth_at(P, F, imageY, imageX);
    if (s.meanDepth < 0.01) {
        int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
        int xe = xb + SP * 2, ye = yb + SP * 2;
        xb = xb > 0 ? xb : 0; yb = yb > 0 ? yb : 0;
        xe = xe < P.W - 1 ? xe : P.W - 1; ye = ye < P.H - 1 ? ye : P.H - 1;
        bool found = false;
        for (int j = yb; j < ye && !found; j++)
            for (int i = xb; i < xe; i++) {
                const float d = depth_at(P, F, j, i);
                if (d > 0.01) { s.meanDepth = d; found = true; break; }
            }
    }
    P.seeds[(size_t)slot * P.nseeds + seedI] = s;
    P.arec[(size_t)slot * P.nseeds + seedI] = assign_rec(s);
}

// kb_assign: a(p) = argmin seed of pixel p (:357-415 without the `stable` gate).  it == 0: every seed is
// unstable, so every free pixel is processed: write the index map directly.  it > 0: store a(p) and run
// relaxation round 0 (pixels whose current seed is unstable at pass start are processed for sure).
//
// One wave per "dual cell" [8 bx + 4, 8 bx + 12) x [8 by + 4, 8 by + 12), bx / by from -1.  Of the 3x3 neighbourhood only the seeds with
// |8c + 4 - x| < 8 on both axes are candidates (:384-389): per axis the pixel's own cell plus the left / upper neighbour when (x mod 8) < 4 or
// the right / lower one when (x mod 8) > 4 -- so ALL pixels of a dual cell have the same candidates {bx, bx + 1} x {by, by + 1} (its first
// column / row, x mod 8 == 4, only the first of each pair).  The candidates are therefore wave-uniform: their fields are scalar operands, and
// the per-pixel work is the four cost evaluations and nothing else.  Enumeration in the reference's order (checkI outer, checkJ inner, ascending).
constexpr int ASSIGN_NY = 2;   // dual cells (one below the other) per wave.  Everything the wave reads -- the NY + 1 lattice rows of candidate records
                               // (scalar loads) and the pixels' member / gray / depth / index words -- is requested before the first use: with one
                               // pixel per lane and loads that wait for one another the kernel had too few bytes in flight to keep HBM busy while
                               // other waves computed (35 us of memory time and 43 us of cost arithmetic per pass simply added up).
__global__ __launch_bounds__(256) void kb_assign(SfDev P, int it, int nSlots, int nbx, int nby) {
    const int bpr = (nbx + 3) >> 2;   // workgroups per row of dual cells (four waves = four dual cells along x)
    const int nbyG = (nby + ASSIGN_NY - 1) / ASSIGN_NY;
    int slot, blk;
    if (!xcd_slot(bpr * nbyG, nSlots, slot, blk)) return;
    if (blk == 0) {
        if (it > 0 && threadIdx.x < 8) P.changed[slot * 8 + threadIdx.x] = threadIdx.x == 0 ? 1 : 0;
        if (threadIdx.x >= 64 && threadIdx.x < 64 + NCHUNK) P.chunkAbort[(slot * 2 + (it & 1)) * 16 + threadIdx.x - 64] = 0x7FFFFFFF;
    }
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int byg = blk / bpr, bxi = (blk - byg * bpr) * 4 + wv;
    if (bxi >= nbx) return;
    const int bx = bxi - 1, by0 = byg * ASSIGN_NY - 1;
    // The candidates (wave-uniform): cell k uses lattice rows by0 + k and by0 + k + 1, in each the neighbours bx and bx + 1 -- two records that
    // are adjacent in memory.  Rows / columns outside the lattice are clamped for the address (the array has a record of padding either side)
    // and never evaluated (the range test of :384-389).
    const bool okx0 = bx >= 0 && bx < P.spW, okx1 = bx + 1 < P.spW;
    const int bxc = min(bx, P.spW - 1);
    const AssignRec *arec = P.arec + (unsigned)slot * (unsigned)P.nseeds;
    AssignRec cr[ASSIGN_NY + 1][2];
    int rowIdx[ASSIGN_NY + 1];
#pragma unroll
    for (int r = 0; r <= ASSIGN_NY; r++) {
        const int rc = min(max(by0 + r, 0), P.spH - 1);
        rowIdx[r] = rc * P.spW + bx;                       // seed index of (bx, by0 + r) when valid
        const AssignRec *rp = arec + (rc * P.spW + bxc);
        cr[r][0] = rp[0]; cr[r][1] = rp[1];
    }
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    float *pxInv = P.pxInv + (size_t)slot * P.pxStride;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const int lane = threadIdx.x & 63, lx = lane & 7, ly = lane >> 3;
    const int colI = 8 * bx + 4 + lx;
    const float colF = (float)colI;
    const bool colIn = colI >= 0 && colI < P.W;
    // ---- all loads of the wave's pixels ----
    bool inImg[ASSIGN_NY];
    int mem[ASSIGN_NY], cur[ASSIGN_NY];
    float gI[ASSIGN_NY], dIn[ASSIGN_NY];
    unsigned tCur[ASSIGN_NY];
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++) {
        const int rowI = 8 * (by0 + k) + 4 + ly;
        inImg[k] = colIn && rowI >= 0 && rowI < P.H && by0 + k + 1 < nby;
        const int rowC = min(max(rowI, 0), P.H - 1), colC = min(max(colI, 0), P.W - 1), pc = rowC * P.W + colC;   // (a clamped address: loaded, never used)
        mem[k] = F.memberG()[(size_t)(rowC / 2) * P.mstride + colC / 2];
        gI[k] = gray_at(P, F, rowC, colC);
        dIn[k] = it == 0 ? depth_at(P, F, rowC, colC) : pxInv[pc];
        cur[k] = it == 0 ? 0 : (int)index[pc];
    }
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++)
        tCur[k] = it == 0 ? 0u : tmin[cur[k]];   // (a plain load: 0 stays 0 and non-zero stays non-zero during the pass, so a stale line answers the same)
    // ---- per cell: the four cost evaluations ----
#pragma unroll
    for (int k = 0; k < ASSIGN_NY; k++) {
        const int by = by0 + k;
        if (__ballot(inImg[k]) == 0) continue;
        const bool oky0 = by >= 0 && by < P.spH, oky1 = by + 1 < P.spH;
        const int rowI = 8 * by + 4 + ly;
        const int p = rowI * P.W + colI;
        const bool isPlane = mem[k] != -1;
        const float myIntensity = gI[k];
        // (float)(1.0 / (double)depth) is the same in all three passes: computed (one f64 divide) in pass 0, read back afterwards
        float myInvDepth = dIn[k];
        if (it == 0) {
            myInvDepth = 0.0f;
            if (dIn[k] > 0.01) myInvDepth = (float)(1.0 / (double)dIn[k]);
            if (inImg[k] && !isPlane) pxInv[p] = myInvDepth;
        }
        const bool pxHasDepth = myInvDepth > 0;
        const double myInvD = (double)myInvDepth;
        const float rowF = (float)rowI;
        float minDistDepth = 1e6f, minDistNodepth = 1e6f;
        int minSpIndexDepth = -1, minSpIndexNodepth = -1;
        bool allHasDepth = true;
        // calculateCost (:333-355) + the two running minima (:398-410) for one candidate; `use` = this pixel has the candidate (x mod 8 == 4:
        // the pixel's own cell only).  Selects instead of branches.
        auto consider = [&](const AssignRec &C, int spIndex, bool use) {
            float nodepthCost = 0;
            const float dist = (C.x - colF) * (C.x - colF) + (C.y - rowF) * (C.y - rowF);
            nodepthCost += dist / ((SP / 2) * (SP / 2));
            const float intensityDiff = C.meanIntensity - myIntensity;
            nodepthCost = (float)((double)nodepthCost + div100_exact((double)(intensityDiff * intensityDiff)));
            const bool has = C.invDepth >= 0 && pxHasDepth;
            const float inverseDepthDiff = (float)(C.invDepth - myInvD);
            const float withDepth = (float)((double)nodepthCost + (double)(inverseDepthDiff * inverseDepthDiff) * 400.0);
            const float depthCost = has ? withDepth : nodepthCost;
            allHasDepth = allHasDepth && (has || !use);
            const bool bd = use && depthCost < minDistDepth, bn = use && nodepthCost < minDistNodepth;
            minDistDepth = bd ? depthCost : minDistDepth; minSpIndexDepth = bd ? spIndex : minSpIndexDepth;
            minDistNodepth = bn ? nodepthCost : minDistNodepth; minSpIndexNodepth = bn ? spIndex : minSpIndexNodepth;
        };
        const bool anyStable = (cr[k][0].stable | cr[k][1].stable | cr[k + 1][0].stable | cr[k + 1][1].stable) != 0;   // (wave-uniform; rare)
        // the reference's order: checkI (x) outer, checkJ (y) inner, ascending
        if (okx0 && oky0) consider(cr[k][0], rowIdx[k], true);
        if (okx0 && oky1) consider(cr[k + 1][0], rowIdx[k + 1], ly != 0);
        if (okx1 && oky0) consider(cr[k][1], rowIdx[k] + 1, lx != 0);
        if (okx1 && oky1) consider(cr[k + 1][1], rowIdx[k + 1] + 1, lx != 0 && ly != 0);
        const int pick = allHasDepth ? minSpIndexDepth : minSpIndexNodepth;
        if (!inImg[k]) continue;
        if (it == 0) { index[p] = isPlane ? (unsigned short)0 : (unsigned short)(pick >= 0 ? pick : 0); continue; }
        amap[p] = isPlane ? IDX_PLANE : (pick >= 0 ? (unsigned short)pick : IDX_NONE);
        if (!isPlane && pick >= 0) {
            // the current seed is unstable at pass start <=> t(cur) == 0 (kb_update_seeds / kb_commit_seeds left 0 or T_INF, and this pass
            // only ever lowers a t to p + 1 >= 1, so a value read at any time during the pass answers the same)
            if (tCur[k] == 0) {
                // processed for sure (round 0): t(pick) = min(t(pick), p + 1) -- only a candidate that entered the pass stable has a t above 0
                if (anyStable && tmin[pick] > (unsigned)p + 1u) atomicMin(&tmin[pick], (unsigned)p + 1u);
            } else if (pick != cur[k]) {
                // Only these pixels can extend a chain: p is processed iff its (stable) seed gets unstabilised before p, and it
                // then unstabilises a DIFFERENT seed.  (pick == cur would only re-lower t(cur) above its current value.)
                P.wl[(size_t)slot * P.pxStride + atomicAdd(&P.wlCount[slot], 1u)] = (unsigned)p;
            }
        }
    }
}

// t(s) = raster position from which seed s counts as unstable: 0 if unstable at pass start, else
// 1 + the first processed pixel that picked it (min-fixpoint).
__device__ __forceinline__ bool relax_pixel(unsigned *tmin, const unsigned short *index, const unsigned short *amap, int p) {
    const unsigned short a = amap[p];
    if (a >= IDX_PLANE) return false;
    const unsigned tc = __hip_atomic_load(&tmin[index[p]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tc == 0 || tc > (unsigned)p) return false;     // tc == 0: handled in round 0; tc > p: not processed (yet)
    if (__hip_atomic_load(&tmin[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)p + 1u) return false;
    return atomicMin(&tmin[a], (unsigned)p + 1u) > (unsigned)p + 1u;
}

constexpr int PROP_BLOCKS = 16;   // workgroups per keyframe over the (small) worklist
__global__ __launch_bounds__(256) void kb_prop(SfDev P, int round, int nSlots) {
    int slot, blk;
    if (!xcd_slot(PROP_BLOCKS, nSlots, slot, blk)) return;
    if (!P.changed[slot * 8 + round]) return;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    const unsigned nwl = P.wlCount[slot];
    bool any = false;
    for (unsigned e = blk * 256 + threadIdx.x; e < nwl; e += PROP_BLOCKS * 256) any |= relax_pixel(tmin, index, amap, (int)wl[e]);
    if (any) P.changed[slot * 8 + round + 1] = 1;
}

// Finisher: one workgroup per keyframe iterates the relaxation to its fixpoint (normally zero rounds).
__global__ __launch_bounds__(1024) void kb_prop_finish(SfDev P) {
    __shared__ int s_ch;
    const int slot = blockIdx.x;
    if (threadIdx.x == 0) s_ch = P.changed[slot * 8 + PROP_ROUNDS];
    __syncthreads();
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    const unsigned nwl = P.wlCount[slot];
    while (s_ch) {
        __syncthreads();
        if (threadIdx.x == 0) s_ch = 0;
        __syncthreads();
        bool any = false;
        for (unsigned e = threadIdx.x; e < nwl; e += 1024) any |= relax_pixel(tmin, index, amap, (int)wl[e]);
        if (any) s_ch = 1;
        __syncthreads();
    }
}

// The whole relaxation in ONE launch: one workgroup per keyframe keeps t(s) in LDS (4 B per seed) and its share of the
// worklist in registers, so a round costs a few LDS operations instead of a kernel boundary plus agent-scope round trips.
// The min-fixpoint is unique, so the evaluation order does not matter.  (kb_prop / kb_prop_finish remain as the fallback
// for seed counts whose t(s) does not fit the LDS.)
constexpr int PROP_LDS_MAX_SEEDS = 36 * 1024;   // 144 KB
__global__ __launch_bounds__(256) void kb_prop_lds(SfDev P) {
    extern __shared__ unsigned s_t[];
    const int slot = blockIdx.x;
    const unsigned nwl = P.wlCount[slot];
    if (nwl == 0) return;
    unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    const unsigned short *index = P.index + (size_t)slot * P.pxStride, *amap = P.amap + (size_t)slot * P.pxStride;
    const unsigned *wl = P.wl + (size_t)slot * P.pxStride;
    constexpr int NT = 256, R = 16;   // a 256-thread workgroup finds room on a busy GPU; a 16-wave one waits for a whole CU
    unsigned ep[R];
    unsigned short ec[R], ea[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const unsigned e = threadIdx.x + r * NT;
        ep[r] = 0xFFFFFFFFu; ec[r] = 0; ea[r] = 0;
        if (e < nwl) { const unsigned p = wl[e]; ep[r] = p; ec[r] = index[p]; ea[r] = amap[p]; }
    }
    for (int i = threadIdx.x; i < P.nseeds; i += NT) s_t[i] = tmin[i];
    __syncthreads();
    auto relax = [&](unsigned p, unsigned short cur, unsigned short a) -> bool {
        if (a >= IDX_PLANE) return false;
        const unsigned tc = s_t[cur];
        if (tc == 0 || tc > p) return false;            // tc == 0: handled in round 0; tc > p: not processed (yet)
        if (s_t[a] <= p + 1u) return false;
        return atomicMin(&s_t[a], p + 1u) > p + 1u;
    };
    int any;
    do {
        bool ch = false;
#pragma unroll
        for (int r = 0; r < R; r++)
            if (ep[r] != 0xFFFFFFFFu) ch |= relax(ep[r], ec[r], ea[r]);
        for (unsigned e = threadIdx.x + R * NT; e < nwl; e += NT) { const unsigned p = wl[e]; ch |= relax(p, index[p], amap[p]); }
        any = __syncthreads_or(ch ? 1 : 0);
    } while (any);
    for (int i = threadIdx.x; i < P.nseeds; i += NT) {
        const unsigned t = s_t[i];
        if (t != tmin[i]) tmin[i] = t;
    }
}

__global__ __launch_bounds__(256) void kb_commit_px(SfDev P, int nSlots) {
    // 8 consecutive pixels per thread (16-byte loads of both maps; the slot stride is a multiple of 64).  A pixel whose pick equals its
    // current seed cannot change, so t(s) is only looked up for the few pixels that picked a different seed.
    int slot, blk;
    if (!xcd_slot(((P.npx + 7) / 8 + 255) / 256, nSlots, slot, blk)) return;
    const int p0 = (blk * 256 + threadIdx.x) * 8;
    if (p0 >= P.npx) return;
    unsigned short *index = P.index + (size_t)slot * P.pxStride;
    const uint4 a4 = *reinterpret_cast<const uint4 *>(P.amap + (size_t)slot * P.pxStride + p0);
    uint4 i4 = *reinterpret_cast<const uint4 *>(index + p0);
    const unsigned *tmin = P.tmin + (size_t)slot * P.nseeds;
    unsigned aw[4] = {a4.x, a4.y, a4.z, a4.w}, iw[4] = {i4.x, i4.y, i4.z, i4.w};
    bool changed = false;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const unsigned a = (aw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu, cur = (iw[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
        if (a >= IDX_PLANE || a == cur || p0 + k >= P.npx) continue;   // (the last group may reach into the slot's padding)
        if (tmin[cur] <= (unsigned)(p0 + k)) {
            iw[k >> 1] = (iw[k >> 1] & ~(0xFFFFu << (16 * (k & 1)))) | (a << (16 * (k & 1)));
            changed = true;
        }
    }
    if (changed) { i4.x = iw[0]; i4.y = iw[1]; i4.z = iw[2]; i4.w = iw[3]; *reinterpret_cast<uint4 *>(index + p0) = i4; }
}

// Four consecutive elements loaded as one access of whatever alignment the element type guarantees (global memory
// tolerates dword-/byte-aligned wide loads).
template <typename T> struct Quad { T v[4]; };
template <typename T> __device__ __forceinline__ Quad<T> load_quad(const T *p) { Quad<T> q; __builtin_memcpy(&q, p, sizeof(q)); return q; }
template <typename T> __device__ __forceinline__ Quad<T> load_quad(gptr<T> p) {
    Quad<T> q;
#pragma unroll
    for (int e = 0; e < 4; e++) q.v[e] = p[e];
    return q;
}
// Inclusive prefix sum over the 16 lanes of a DPP row (= one seed group); lanes without a source read 0.
__device__ __forceinline__ int row_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);   // row_shr:8
    return v;
}

// kb_update_seeds (:428-515): 16 lanes per seed (lane = window row), 16 seeds per workgroup.
// Integer-valued sums are exact in any order; the float depth sum and the Huber/Newton sums run in window
// raster order on the group's first lane, fed by terms the 16 lanes prepare in parallel.
template <bool STRADDLE>   // STRADDLE: W mod 8 in {1, 2, 3} (a window quad can stick out over the right edge); the common instantiation stays at 80 VGPRs,
                           // so that three k_fuse waves (64 VGPRs) fit next to its four waves per SIMD -- with 88 only two did (+0.5 us per k_fuse launch)
__global__ __launch_bounds__(256) void kb_update_seeds(SfDev P, int it, int nSlots) {
    __shared__ __attribute__((aligned(16))) float s_depth[16][256];
    __shared__ __attribute__((aligned(16))) float s_term[16][256];   // in-range: 2*residual; Huber tails: +-inf markers
    __shared__ float s_mean[16];
    __shared__ int s_cnt[16], s_done[16];
    int slot, blk;
    if (!xcd_slot((P.nseeds + 15) / 16, nSlots, slot, blk)) return;
#ifdef MSL_FUSE_STAMPS   // section cycle counts of the waves of slot 0, summed into delList[96 ..] (tools/fuse_stamps.py)
    unsigned long long ust[8]; int usn = 0;
#define USTAMP() ust[usn++] = __builtin_amdgcn_s_memtime()
#else
#define USTAMP()
#endif
    USTAMP();
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15;
    const int seedI = blk * 16 + g;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
    const unsigned short *index = P.index + (size_t)slot * P.pxStride;
    msl_seed S;
    memset(&S, 0, sizeof(S));
    bool active = seedI < P.nseeds;
    bool stable = false;
    if (active) {
        S = P.seeds[(size_t)slot * P.nseeds + seedI];
        stable = it > 0 ? (P.tmin[(size_t)slot * P.nseeds + seedI] == T_INF) : (S.stable != 0);
        // Seeds are updated in place.  A processed seed first saves its old record in seedsTmp[] (marked _pad = 2), so the
        // commit pass can restore it when the chunk turns out to have ended earlier; everyone else clears that mark.
        if (!S.use || stable) {
            if (l == 0) {   // skipped: only the stable flag (as left by the pixel pass) and t(s) change
                P.seeds[(size_t)slot * P.nseeds + seedI].stable = stable;
                P.arec[(size_t)slot * P.nseeds + seedI].stable = stable ? 1u : 0u;
                P.seedsTmp[(size_t)slot * P.nseeds + seedI]._pad = 0;
                P.tmin[(size_t)slot * P.nseeds + seedI] = stable ? T_INF : 0u;
            }
            active = false;
        }
    }
    if (!__ballot(active)) return;   // all four seeds of the wave are skipped (stable or unused): nothing to gather
    const int spX = seedI % P.spW, spY = seedI / P.spW;
    const int xb0 = spX * SP + SP / 2 - SP, yb0 = spY * SP + SP / 2 - SP;
    const int xb = xb0 > 0 ? xb0 : 0, yb = yb0 > 0 ? yb0 : 0;
    const int xe = (xb0 + SP * 2) < P.W - 1 ? (xb0 + SP * 2) : P.W - 1, ye = (yb0 + SP * 2) < P.H - 1 ? (yb0 + SP * 2) : P.H - 1;
    int sumX = 0, sumY = 0, sumI = 0, cnt = 0, nd = 0;
    {
        // Lane = (row r of a group of four window rows, quad q of four window columns): 12 wide loads per lane (8 B of
        // index, 16 B of depth, 4 B of gray, four times) instead of 48 scalar ones.  Window columns start at a multiple
        // of 4: a quad lies left of the image as a whole (first lattice column) or starts inside it.  When W is not a multiple of 4 the last
        // quad of a window may stick out over the right edge: it is then loaded from W - 4 (inside the row) and its first elements, which
        // belong to the neighbouring lane's quad, are masked (col >= col0) -- no element-wise path, the window order is unchanged.  Raster
        // order of the window = (iteration, lane, element), which the ordered depth list below follows.
        const int rq = l >> 2, cq = l & 3;
        const int col0 = xb0 + 4 * cq;
        const bool quadIn = col0 >= 0 && (STRADDLE ? col0 < P.W : col0 + 3 < P.W);
        const int colc = quadIn ? (STRADDLE ? min(col0, P.W - 4) : col0) : 0;
        Quad<unsigned short> idq[4];
        Quad<float> dq[4];
        Quad<uint8_t> gq[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int jc = min(max(yb0 + 4 * m + rq, 0), P.H - 1);
            idq[m] = load_quad(index + (size_t)jc * P.W + colc);
            dq[m] = load_quad(F.depthG() + (size_t)jc * P.dstride + colc);
            gq[m] = load_quad(F.grayG() + (size_t)jc * P.gstride + colc);
        }
        const int g15 = (threadIdx.x & 48) | 15;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = yb0 + 4 * m + rq;
            const bool rowOk = active && quadIn && j >= yb && j < ye;
            bool hd[4];
            int c = 0;
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int col = colc + e;
                const bool own = rowOk && (!STRADDLE || col >= col0) && col >= xb && col < xe && idq[m].v[e] == seedI;
                hd[e] = own && dq[m].v[e] > 0.1;
                if (own) { sumX += col; sumY += j; sumI += gq[m].v[e]; cnt++; }
                c += hd[e] ? 1 : 0;
            }
            const int incl = row_incl_scan(c);
            int o = nd + incl - c;
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (hd[e]) s_depth[g][o++] = dq[m].v[e];
            nd += __shfl(incl, g15, 64);
        }
    }
    USTAMP();   // 1: seed record + window gather + ordered depth list
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        sumX += __shfl_xor(sumX, d, 16); sumY += __shfl_xor(sumY, d, 16);
        sumI += __shfl_xor(sumI, d, 16); cnt += __shfl_xor(cnt, d, 16);
    }
    __builtin_amdgcn_wave_barrier();
    msl_seed T = S;
    bool depthLoop = false, aborted = false;
    if (l == 0) {
        if (active) {
            if (cnt == 0) {  // `return`: ends the chunk (:473-474); the seed itself stays as it is, unstable
                atomicMin(&P.chunkAbort[(slot * 2 + (it & 1)) * 16 + seed_chunk(seedI, P.nseeds)], seedI);
                aborted = true;
            } else {
                const float sumIntensityNum = (float)cnt;
                const float sumIntensity = (float)sumI / sumIntensityNum, mX = (float)sumX / sumIntensityNum, mY = (float)sumY / sumIntensityNum;
                const float preIntensity = S.meanIntensity, preX = S.x, preY = S.y;
                T.meanIntensity = sumIntensity; T.x = mX; T.y = mY;
                vec3b(P, F, mY, mX, T.r, T.g, T.b);
                const float updateDiff = fabsf(preIntensity - sumIntensity) + fabsf(preX - mX) + fabsf(preY - mY);
                T.stable = (updateDiff < 0.2) ? 1 : 0;
                if (nd > 0) {
                    const float sumDepth = seq_sum_f32(s_depth[g], nd, 0.0f);
                    s_mean[g] = sumDepth / (float)nd;
                    depthLoop = true;
                } else {
                    T.meanDepth = 0.0f;
                }
            }
        }
        s_done[g] = depthLoop ? 0 : 1;
    }
    __builtin_amdgcn_wave_barrier();
    USTAMP();   // 2: means, colour fetch, sequential depth sum
    // Huber mean depth: <= 5 Newton steps (:492-512); terms in parallel, accumulation in list order
    for (int newtonI = 0; newtonI < 5; newtonI++) {
        if (s_done[g]) break;
        if (l == 0) s_cnt[g] = 0;
        __builtin_amdgcn_wave_barrier();
        const float meanDepth = s_mean[g];
        int inr = 0;
        for (int e = l; e < nd; e += 16) {
            const float residual = meanDepth - s_depth[g][e];
            if (residual < HUBER_RANGE && residual > -HUBER_RANGE) { s_term[g][e] = 2 * residual; inr++; }
            else s_term[g][e] = residual > 0 ? __builtin_inff() : -__builtin_inff();
        }
        if (inr) atomicAdd(&s_cnt[g], inr);
        __builtin_amdgcn_wave_barrier();
        if (l == 0) {
            // no Huber tails (the common case): a plain float chain, 1 VALU op per element instead of ~8
            const float sumA = s_cnt[g] == nd ? seq_sum_f32(s_term[g], nd, 0.0f) : seq_sum_huber(s_term[g], nd, 0.0f);
            const float sumB = (float)(2 * s_cnt[g]);
            const float deltaDepth = (float)((double)(-sumA) / ((double)sumB + 10.0));
            const float m = meanDepth + deltaDepth;
            s_mean[g] = m;
            if ((deltaDepth < 0.01 && deltaDepth > -0.01) || newtonI == 4) { s_done[g] = 1; }
        }
        __builtin_amdgcn_wave_barrier();
    }
    USTAMP();   // 3: Newton steps
    if (active && l == 0) {
        const size_t si = (size_t)slot * P.nseeds + seedI;
        if (aborted) {
            P.seeds[si].stable = 0; P.arec[si].stable = 0u; P.seedsTmp[si]._pad = 0; P.tmin[si] = 0u;
        } else {
            if (depthLoop) T.meanDepth = s_mean[g];
            msl_seed old = S;
            old._pad = 2;
            P.seedsTmp[si] = old;
            T._pad = 0;
            P.seeds[si] = T;
            P.tmin[si] = T.stable ? T_INF : 0u;
            P.arec[si] = assign_rec(T);
        }
    }
#ifdef MSL_FUSE_STAMPS
    USTAMP();   // 4: stores
    if (slot == 0 && (threadIdx.x & 63) == 0) {
        for (int q = 1; q < usn; q++) atomicAdd(&P.delList[96 + q], (unsigned)(ust[q] - ust[q - 1]));
        atomicAdd(&P.delList[96], 1u);
    }
#endif
}


// kb_commit_seeds: the chunk-abort rule.  Normally nothing to do (no chunk ended early); otherwise a seed that was processed
// although its chunk had already ended gets its old record back, unstable ("values untouched", :473-474).
// The rule can never fire: a used seed (lattice position spX < W / 8, spY < H / 8) always owns the pixel at its lattice
// centre (8 spX + 4, 8 spY + 4).  That pixel is free (what `use` means, :541-545); its ONLY updatePixels candidate is this seed
// (|8 c + 4 - x| < 8 holds for c = spX alone when x mod 8 == 4, :384-389); pass 0 assigns it with cost 0 < 1e6 whatever intensity / depth
// are; no later pass can move it; and it lies inside the clipped window updateSeeds counts.  So the owned-pixel count is >= 1 and the
// restore path is kept for fidelity only (property-tested on adversarial inputs in the CPU suite).
__global__ __launch_bounds__(256) void kb_commit_seeds(SfDev P, int it) {
    const int slot = blockIdx.y;
    const int seedI = blockIdx.x * 256 + threadIdx.x;
    if (seedI >= P.nseeds) return;
    if (seedI == 0) P.wlCount[slot] = 0;   // the next pixel pass rebuilds the relaxation worklist
    if (seedI < P.chunkAbort[(slot * 2 + (it & 1)) * 16 + seed_chunk(seedI, P.nseeds)]) return;
    const size_t si = (size_t)slot * P.nseeds + seedI;
    if (P.seedsTmp[si]._pad != 2) return;   // skipped, or the seed that ended the chunk: already as it should be
    msl_seed out = P.seedsTmp[si];
    out.stable = 0; out._pad = 0;
    P.seeds[si] = out;
    P.tmin[si] = 0u;
    P.arec[si] = assign_rec(out);
}

// kb_seed_plane: calculateNorms (:775-803) fused per seed, 16 lanes per seed, 4 seeds per wave/workgroup.
// Pixel positions and cross-product normals are recomputed from depth instead of materialising spaceMap
// (7.4 MB f64) / normMap.  Also prepares the surfel the seed would spawn (initializeSurfels, :285-331).
__device__ __forceinline__ void pixel_normal(const SfDev &P, int row, int col, float myX, float myY, float myZ, float rightDepth,
                                             float downDepth, float cxr, float cx1, float ryr, float ry1,
                                             float &nX, float &nY, float &nZ) {
    nX = nY = nZ = 0.0f;
    if (row < 1 || row > P.H - 2 || col < 1 || col > P.W - 2) return;  // never written (:620-625)
    // back_project of the right / down neighbours with the tabulated quotients: (col+1, row) and (col, row+1)
    float rightX = cx1 * rightDepth, rightY = ryr * rightDepth, rightZ = rightDepth;
    float downX = cxr * downDepth, downY = ry1 * downDepth, downZ = downDepth;
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

// Sum over the 16 lanes of a DPP row (= one seed group); every lane receives the total.  Row rotations by 8 and 4
// and quad permutes run in the VALU (a few cycles) instead of ds_bpermute round trips through the LDS crossbar.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double group_sum_d(double v) {
    v += dpp_mov_d<0x128>(v);   // row_ror:8
    v += dpp_mov_d<0x124>(v);   // row_ror:4
    v += dpp_mov_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov_d<0xB1>(v);    // quad_perm [1,0,3,2]
    return v;
}

// FuseRec: the 48 bytes of a seed that fuseSurfelsKernel reads (three 16-byte loads instead of the 64-byte msl_seed), with the terms
// that depend on the seed alone evaluated once per seed instead of once per fused surfel -- same expressions, same operands:
//   [0] normX, normY, normZ (camera frame), meanDepth
//   [1] pose * (posX, posY, posZ, 1) (:240-245), getWeight(meanDepth) (:236)
//   [2] size * fabs(meanDepth / (cameraF * viewCos)) (:270-271), meanIntensity, r | g << 8 | b << 16, valid
// valid = !(norm == 0) && !(viewCos < MAX_ANGLE_COS), the two seed tests of :214-219.
//
// LDS: one pool per wave.  The four seeds of a wave form a 2x2 block of the seed lattice, so their 16x16 windows cover
// 24x24 = 576 distinct pixels; every pixel belongs to one seed, hence the four ordered lists hold <= 576 entries in total
// (+ 3 x 3 for 16-byte alignment of each list) instead of 4 x 256.  14 KB per wave: 11 waves per CU instead of 5.
constexpr int PLANE_POOL = 24 * 24 + 12;
template <bool STRADDLE>   // STRADDLE: W mod 8 in {1, 2, 3} -- a window quad can stick out over the right edge (instantiated separately: the common
                           // geometry carries none of that code)
__global__ __launch_bounds__(64) void kb_seed_plane(SfDev P, int nSlots) {
    __shared__ __attribute__((aligned(16))) float s_pool[6][PLANE_POOL];   // position x y z, normal x y z
    __shared__ __attribute__((aligned(16))) double s_h[4][16];
    int slot, blk;
    const int bW = (P.spW + 1) / 2, bH = (P.spH + 1) / 2;
    if (!xcd_slot(bW * bH, nSlots, slot, blk)) return;
#ifdef MSL_FUSE_STAMPS   // section cycle counts of the waves of slot 0, summed into delList[64 ..] (tools/fuse_stamps.py)
    unsigned long long sst[14]; int ssn = 0;
#define SECTION_STAMP() sst[ssn++] = __builtin_amdgcn_s_memtime()
#else
#define SECTION_STAMP()
#endif
    SECTION_STAMP();
    const int g = threadIdx.x >> 4, l = threadIdx.x & 15, lane = threadIdx.x;
    const int spX = (blk % bW) * 2 + (g & 1), spY = (blk / bW) * 2 + (g >> 1);
    const bool inRange = spX < P.spW && spY < P.spH;
    const int seedI = inRange ? spY * P.spW + spX : 0;
    const FrameDev F = P.frames[slot];   // by value: one load up front instead of re-reading fields around every store
#ifdef MSL_FUSE_STAMPS
    { unsigned long long a = (unsigned long long)F.depth; asm volatile("" :: "s"(a)); }
    SECTION_STAMP();   // 0a: kernel arguments + frame record
#endif
    const unsigned short *index = P.index + (size_t)slot * P.pxStride;
    msl_seed S;
    memset(&S, 0, sizeof(S));
    if (inRange) S = P.seeds[(size_t)slot * P.nseeds + seedI];
#ifdef MSL_FUSE_STAMPS
    asm volatile("" :: "v"(S.x), "v"(S.meanDepth));
    SECTION_STAMP();   // 0b: seed record
#endif
    const int xb = spX * SP + SP / 2 - SP, yb = spY * SP + SP / 2 - SP;
    // ---- gather: lane = (row r of a group of four window rows, quad q of four window columns), four iterations; the
    // unclipped window is guarded by the flat index range (:680-684).  16 wide loads per lane: 8 B of index, 16 B of depth,
    // 16 B of the row below, 4 B right of the quad (the other right neighbours are the quad's own elements). ----
    float maxDist = 0;
    int nvalid = 0, base = 0, poolUsed = 0;
    {
        const int rq = l >> 2, cq = l & 3;
        // wrapped pixels (App. B.6) without an integer division: a quad left / right of the image (window columns start at a multiple
        // of 4) belongs to the previous / next row of the flat index.  When W is not a multiple of 4 the last quad of a window in the last
        // lattice column can straddle the right edge: its elements beyond W - 1 are the first pixels of the next row (`straddle`, rare:
        // element-wise loads).
        const int cx0 = xb + 4 * cq;
        const int wrapRow = cx0 < 0 ? -1 : (cx0 >= P.W ? 1 : 0), wcol0 = cx0 - wrapRow * P.W;
        const bool straddle = STRADDLE && cx0 < P.W && cx0 + 3 >= P.W;
        auto elem_wrap = [&](int e) -> int { return (straddle && cx0 + e >= P.W) ? 1 : 0; };   // extra row wrap of element e of a straddling quad
        Quad<unsigned short> idq[4];
        Quad<float> dq[4], ddq[4];
        float dr3[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int wr = yb + 4 * m + rq + wrapRow;
            const int row = min(max(wr, 0), P.H - 1);     // rows outside the image fail the flat-index test below
            if (!straddle) {
                idq[m] = load_quad(index + (size_t)row * P.W + wcol0);
                dq[m] = load_quad(F.depthG() + (size_t)row * P.dstride + wcol0);
                ddq[m] = load_quad(F.depthG() + (size_t)min(row + 1, P.H - 1) * P.dstride + wcol0);
                dr3[m] = F.depthG()[(size_t)row * P.dstride + min(wcol0 + 4, P.W - 1)];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const int we = elem_wrap(e), col = cx0 + e - we * P.W, rowe = min(max(wr + we, 0), P.H - 1);
                    idq[m].v[e] = index[(size_t)rowe * P.W + col];
                    dq[m].v[e] = F.depthG()[(size_t)rowe * P.dstride + col];
                    ddq[m].v[e] = F.depthG()[(size_t)min(rowe + 1, P.H - 1) * P.dstride + col];
                }
                dr3[m] = 0.0f;
            }
        }
        // Texel map for k_fuse: every pixel's {depth, final index} as one 8-byte word.  The seed's own 8x8 cell is rows / columns
        // [4, 12) of its window (iterations 1, 2; column quads 1, 2), and the cells tile the image, so each pixel is written exactly
        // once from values this lane holds anyway: two 16-byte stores per iteration for half of the lanes.
        if (inRange && (cq == 1 || cq == 2)) {
            uint2 *tex = P.tex + (size_t)slot * P.pxStride;
#pragma unroll
            for (int m = 1; m <= 2; m++) {
                uint4 *t4 = reinterpret_cast<uint4 *>(tex + (size_t)(yb + 4 * m + rq) * P.W + cx0);
                t4[0] = make_uint4(__float_as_uint(dq[m].v[0]), idq[m].v[0], __float_as_uint(dq[m].v[1]), idq[m].v[1]);
                t4[1] = make_uint4(__float_as_uint(dq[m].v[2]), idq[m].v[2], __float_as_uint(dq[m].v[3]), idq[m].v[3]);
            }
        }
        unsigned vm = 0;   // bit 4 m + e: pixel e of the quad in iteration m is a valid-depth pixel of the seed
#pragma unroll
        for (int m = 0; m < 4; m++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int i = xb + 4 * cq + e, jrow = yb + 4 * m + rq;
                const int pixelIndex = jrow * P.W + i;
                if (inRange && pixelIndex >= 0 && pixelIndex < P.npx && idq[m].v[e] == seedI) {
                    const float xDiff = i - S.x, yDiff = jrow - S.y;
                    const float dist = xDiff * xDiff + yDiff * yDiff;
                    if (dist > maxDist) maxDist = dist;
                    if (dq[m].v[e] > 0.05) vm |= 1u << (4 * m + e);
                }
            }
        nvalid = __popc(vm);
        SECTION_STAMP();   // 1a: window loads arrived, ownership tests
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) nvalid += __shfl_xor(nvalid, d, 16);
        {   // list bases inside the pool, each rounded up to 4 entries
            const int pad = (nvalid + 3) & ~3;
            const int n0 = __shfl(pad, 0, 64), n1 = __shfl(pad, 16, 64), n2 = __shfl(pad, 32, 64), n3 = __shfl(pad, 48, 64);
            base = g == 0 ? 0 : g == 1 ? n0 : g == 2 ? n0 + n1 : n0 + n1 + n2;
            poolUsed = n0 + n1 + n2 + n3;
            // the <= 3 padding entries behind a list take part in the wave-wide pass below: give them a valid pixel (row 0, column 0)
            if (l < pad - nvalid) { s_pool[2][base + nvalid + l] = 0.0f; s_pool[3][base + nvalid + l] = 0.0f; s_pool[4][base + nvalid + l] = 0.0f; s_pool[5][base + nvalid + l] = 0.0f; }
        }
        const int g15 = (lane & 48) | 15;
        int run = base;
#pragma unroll
        for (int m = 0; m < 4; m++) {   // ordered compaction in window raster order = (iteration, lane, element)
            const unsigned q = (vm >> (4 * m)) & 0xFu;
            const int c = __popc(q);
            const int incl = row_incl_scan(c);
            int o = run + incl - c;
            const int rc = ((yb + 4 * m + rq + wrapRow) << 16) | wcol0;   // a valid pixel lies inside the image
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (q & (1u << e)) {   // depth, right depth, down depth, (row, col)
                    float right = e < 3 ? dq[m].v[e < 3 ? e + 1 : 3] : dr3[m];
                    int rce = rc + e;
                    if (straddle) {   // (row, col) and the right neighbour of an element of a straddling quad, fetched here (rare)
                        const int we = elem_wrap(e), col = cx0 + e - we * P.W, rowe = yb + 4 * m + rq + we;
                        rce = (rowe << 16) | col;
                        right = F.depthG()[(size_t)rowe * P.dstride + min(col + 1, P.W - 1)];
                    }
                    s_pool[2][o] = dq[m].v[e]; s_pool[3][o] = right;
                    s_pool[4][o] = ddq[m].v[e]; s_pool[5][o] = __int_as_float(rce);
                    o++;
                }
            run += __shfl(incl, g15, 64);
        }
    }
    SECTION_STAMP();   // 1: gather + ordered lists
    float *const pX = s_pool[0] + base, *const pY = s_pool[1] + base, *const pZ = s_pool[2] + base;
    float *const qX = s_pool[3] + base, *const qY = s_pool[4] + base, *const qZ = s_pool[5] + base;
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) maxDist = fmaxf(maxDist, __shfl_xor(maxDist, d, 16));
    __builtin_amdgcn_wave_barrier();
    // entry e -> position + cross-product normal, written back in place (order preserved).  The work per entry does not depend on the seed, so
    // the 64 lanes walk the whole pool together: ceil(pool / 64) rounds instead of ceil(longest list / 16) -- the four superpixels of a wave
    // rarely have the same size.
    for (int e = lane; e < poolUsed; e += 64) {
        const int rc = __float_as_int(s_pool[5][e]);
        const int row = rc >> 16, col = rc & 0xFFFF;     // a valid pixel lies inside the image: (row, col) of its flat index
        const float myDepth = s_pool[2][e], rightD = s_pool[3][e], downD = s_pool[4][e];
        const float cxr = P.colX[col], cx1 = P.colX[col + 1], ryr = P.rowY[row], ry1 = P.rowY[row + 1];
        const float x = cxr * myDepth, y = ryr * myDepth;   // back_project(col, row, myDepth)
        float nX, nY, nZ;
        pixel_normal(P, row, col, x, y, myDepth, rightD, downD, cxr, cx1, ryr, ry1, nX, nY, nZ);
        s_pool[0][e] = x; s_pool[1][e] = y;
        s_pool[3][e] = nX; s_pool[4][e] = nY; s_pool[5][e] = nZ;
    }
    __builtin_amdgcn_wave_barrier();
    SECTION_STAMP();   // 2: positions + pixel normals
    bool active = inRange && nvalid >= 16;   // validDepthNum < 16 -> continue (:702)
    float meanDepth = S.meanDepth;
    // ---- inliers, kept in order (:707-720).  Count first: when every valid pixel is an inlier (the common case)
    // the list is already in place; otherwise in-place ordered compaction, 16 entries per round. ----
    int ninl = 0;
    {
        int c = 0;
        if (active)
            for (int o = l; o < nvalid; o += 16) {
                const float residual = meanDepth - pZ[o];
                c += (residual < HUBER_RANGE && residual > -HUBER_RANGE) ? 1 : 0;
            }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) c += __shfl_xor(c, d, 16);
        ninl = c;
    }
    const bool needCompact = active && ninl != nvalid;
    if (__ballot(needCompact)) {
        int w0 = 0;
        for (int t = 0; t < 16; t++) {
            const int o = t * 16 + l;
            bool inl = false;
            float a0 = 0, a1 = 0, a2 = 0, b0 = 0, b1 = 0, b2 = 0;
            if (needCompact && o < nvalid) {
                const float residual = meanDepth - pZ[o];
                inl = residual < HUBER_RANGE && residual > -HUBER_RANGE;
                a0 = pX[o]; a1 = pY[o]; a2 = pZ[o];
                b0 = qX[o]; b1 = qY[o]; b2 = qZ[o];
            }
            const unsigned gm = (unsigned)((__ballot(inl) >> (g * 16)) & 0xFFFFull);
            __builtin_amdgcn_wave_barrier();   // every lane has read its slot before anyone overwrites (w <= o)
            if (inl) {
                const int w = w0 + __popc(gm & ((1u << l) - 1u));
                pX[w] = a0; pY[w] = a1; pZ[w] = a2;
                qX[w] = b0; qY[w] = b1; qZ[w] = b2;
            }
            w0 += __popc(gm);
            __builtin_amdgcn_wave_barrier();
        }
    }
    SECTION_STAMP();   // 3: inlier count / compaction
    if (active && (float)ninl / (float)nvalid < 0.8) active = false;
    // Six strictly sequential f32 sums (inlier normals x,y,z and positions x,y,z, :709-713 and :95-99) run side by side:
    // lane q < 6 of the group walks array q in list order, so the serial latency is one chain instead of six.
    float normX, normY, normZ, sumX, sumY, sumZ;
    {
        float acc = 0.0f;
        if (active && l < 6) acc = seq_sum_f32(s_pool[l < 3 ? 3 + l : l - 3] + base, ninl, 0.0f);
        const int gb = lane & 48;
        normX = __shfl(acc, gb + 0, 64); normY = __shfl(acc, gb + 1, 64); normZ = __shfl(acc, gb + 2, 64);
        sumX = __shfl(acc, gb + 3, 64); sumY = __shfl(acc, gb + 4, 64); sumZ = __shfl(acc, gb + 5, 64);
        const float normLength = sqrtf(normX * normX + normY * normY + normZ * normZ);
        normX = normX / normLength; normY = normY / normLength; normZ = normZ / normLength;
        sumX /= ninl; sumY /= ninl; sumZ /= ninl;
    }
    SECTION_STAMP();   // 4: six sequential sums
    // ---- getHuberNorm (:91-165): 5 Gauss-Newton steps, FP64 normal equations reduced over the 16 lanes ----
    float nx = normX, ny = normY, nz = normZ, nb = 0.0f;
    // The Hessian depends only on WHICH points lie inside the Huber band; while that set is unchanged between
    // iterations (the common case: all of them) its sums -- and the inverse -- are bit-identical and are reused.
    unsigned prevMask = 0xFFFFFFFFu;   // impossible mask: forces the first evaluation
    // Cooperative 4x4 inverse: lane l = 4a+b of the group evaluates cofactor (a,b) with exactly the DET3 expression of
    // inverse4(), so lane l ends up holding inv[l] (column-major) -- 1/16 of the work and 2 instead of 32 registers.
    double invl = 0;
    const int ca = l >> 2, cb = l & 3;
    const int r0 = ca == 0 ? 1 : 0, r1 = ca <= 1 ? 2 : 1, r2 = ca <= 2 ? 3 : 2;
    const int c0 = cb == 0 ? 1 : 0, c1 = cb <= 1 ? 2 : 1, c2 = cb <= 2 ? 3 : 2;
    const int gbase = lane & 48;
    int tRounds = active ? (ninl + 15) >> 4 : 0;
#pragma unroll
    for (int d = 32; d >= 16; d >>= 1) tRounds = max(tRounds, __shfl_xor(tRounds, d, 64));   // ninl is uniform inside a group of 16 lanes
    tRounds = __builtin_amdgcn_readfirstlane(tRounds);
    for (int gnI = 0; gnI < 5; gnI++) {
        double J0 = 0, J1 = 0, J2 = 0, J3 = 0;
        unsigned mask = 0;
        if (active) {
#pragma unroll 1   // (not unrolled: 128 instead of 130 VGPRs = 3 x 128 per SIMD, which leaves room for two 64-register k_fuse waves instead of one)
            for (int t = 0; t < tRounds; t++) {   // (a wave-uniform bound: the longest inlier list of the four seeds, typically 4-6 of the 16 rounds)
                const int o = l + 16 * t;
                if (o < ninl) {
                    const float px = pX[o] - sumX, py = pY[o] - sumY, pz = pZ[o] - sumZ;
                    const float residual = px * nx + py * ny + pz * nz + nb;
                    if (residual < HUBER_RANGE && residual > -1 * HUBER_RANGE) {
                        mask |= 1u << t;
                        J0 += 2 * residual * px; J1 += 2 * residual * py; J2 += 2 * residual * pz; J3 += 2 * residual;
                    } else if (residual >= HUBER_RANGE) {
                        J0 += HUBER_RANGE * px; J1 += HUBER_RANGE * py; J2 += HUBER_RANGE * pz; J3 += HUBER_RANGE;
                    } else if (residual <= -1 * HUBER_RANGE) {
                        J0 += -1 * HUBER_RANGE * px; J1 += -1 * HUBER_RANGE * py; J2 += -1 * HUBER_RANGE * pz; J3 += -1 * HUBER_RANGE;
                    }
                }
            }
        }
        J0 = group_sum_d(J0); J1 = group_sum_d(J1); J2 = group_sum_d(J2); J3 = group_sum_d(J3);
        const bool sameSet = mask == prevMask;
        const unsigned diffGroups = (unsigned)((__ballot(!sameSet) >> (g * 16)) & 0xFFFFull);   // uniform per group
        prevMask = mask;
        if (__ballot(diffGroups != 0)) {
            double H00 = 0, H01 = 0, H02 = 0, H03 = 0, H11 = 0, H12 = 0, H13 = 0, H22 = 0, H23 = 0, H33 = 0;
            if (active && diffGroups) {
#pragma unroll 1
                for (int t = 0; t < tRounds; t++)
                    if (mask & (1u << t)) {
                        const int o = l + 16 * t;
                        const float px = pX[o] - sumX, py = pY[o] - sumY, pz = pZ[o] - sumZ;
                        H00 += 2 * px * px; H01 += 2 * px * py; H02 += 2 * px * pz; H03 += 2 * px;
                        H11 += 2 * py * py; H12 += 2 * py * pz; H13 += 2 * py;
                        H22 += 2 * pz * pz; H23 += 2 * pz; H33 += 2;
                    }
            }
            H00 = group_sum_d(H00); H01 = group_sum_d(H01); H02 = group_sum_d(H02); H03 = group_sum_d(H03);
            H11 = group_sum_d(H11); H12 = group_sum_d(H12); H13 = group_sum_d(H13);
            H22 = group_sum_d(H22); H23 = group_sum_d(H23); H33 = group_sum_d(H33);
            if (l == 0) {   // the (symmetric) Hessian + 5 I, column-major
                double *m = s_h[g];
                m[0] = H00 + 5; m[1] = H01; m[2] = H02; m[3] = H03; m[4] = H01; m[5] = H11 + 5; m[6] = H12; m[7] = H13;
                m[8] = H02; m[9] = H12; m[10] = H22 + 5; m[11] = H23; m[12] = H03; m[13] = H13; m[14] = H23; m[15] = H33 + 5;
            }
            __builtin_amdgcn_wave_barrier();
            {
                const double *m = s_h[g];
#define M_(r, c) m[(c) * 4 + (r)]
                const double d3 = M_(r0, c0) * (M_(r1, c1) * M_(r2, c2) - M_(r1, c2) * M_(r2, c1)) -
                                  M_(r0, c1) * (M_(r1, c0) * M_(r2, c2) - M_(r1, c2) * M_(r2, c0)) +
                                  M_(r0, c2) * (M_(r1, c0) * M_(r2, c1) - M_(r1, c1) * M_(r2, c0));
                const double cof = ((ca + cb) & 1) ? -d3 : d3;
                const double f0 = __shfl(cof, gbase + 0, 64), f1 = __shfl(cof, gbase + 1, 64), f2 = __shfl(cof, gbase + 2, 64),
                             f3 = __shfl(cof, gbase + 3, 64);
                const double det = ((M_(0, 0) * f0 + M_(0, 1) * f1) + M_(0, 2) * f2) + M_(0, 3) * f3;
#undef M_
                if (diffGroups) invl = cof / det;
            }
            __builtin_amdgcn_wave_barrier();
        }
        // upd[r] = ((inv[0*4+r] J0 + inv[1*4+r] J1) + inv[2*4+r] J2) + inv[3*4+r] J3; lane l holds inv[l], its column is l >> 2
        const double prod = invl * (ca == 0 ? J0 : ca == 1 ? J1 : ca == 2 ? J2 : J3);
        const double q1 = __shfl(prod, gbase + 4 + cb, 64), q2 = __shfl(prod, gbase + 8 + cb, 64), q3 = __shfl(prod, gbase + 12 + cb, 64);
        const double q0 = __shfl(prod, gbase + cb, 64);
        const double updr = ((q0 + q1) + q2) + q3;            // lane with cb == r now holds upd[r]
        const double u0 = __shfl(updr, gbase + 0, 64), u1 = __shfl(updr, gbase + 1, 64), u2 = __shfl(updr, gbase + 2, 64),
                     u3 = __shfl(updr, gbase + 3, 64);
        nx = (float)((double)nx - u0); ny = (float)((double)ny - u1); nz = (float)((double)nz - u2); nb = (float)((double)nb - u3);
        SECTION_STAMP();   // 5-9: Gauss-Newton steps
    }
#ifdef MSL_FUSE_STAMPS
    if (slot == 0 && lane == 0) {
        for (int q = 1; q < ssn; q++) atomicAdd(&P.delList[64 + q], (unsigned)(sst[q] - sst[q - 1]));
        atomicAdd(&P.delList[64], 1u);
    }
#endif
    if (!inRange || l != 0) return;
    if (active) {
        nb = nb - (nx * sumX + ny * sumY + nz * sumZ);
        {
            const float normLength = sqrtf(nx * nx + ny * ny + nz * nz);
            nx /= normLength; ny /= normLength; nz /= normLength; nb /= normLength;
        }
        normX = nx; normY = ny; normZ = nz;
        const float normB = nb;
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
        P.seeds[(size_t)slot * P.nseeds + seedI] = S;
    }
    // what the map stage reads of this seed (FuseRec) and the candidate new surfel (:291-329, everything except the `fused` test,
    // which needs the map stage); both use the same per-seed terms
    const bool valid = !(S.viewCos < MAX_ANGLE_COS) && !(S.normX == 0 && S.normY == 0 && S.normZ == 0);
    const bool ok = valid && !(S.meanDepth == 0);
    P.candOk[(size_t)slot * P.nseeds + seedI] = ok ? 1 : 0;
    float pw[4] = {0, 0, 0, 0};
    float seedWeight = 0, seedSize = 0;
    if (valid) {
        mul4(F.pose, S.posX, S.posY, S.posZ, 1.0f, pw);
        const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
        seedSize = S.size * fabsf(S.meanDepth / (cameraF * S.viewCos));
        seedWeight = get_weight(S.meanDepth);
    }
    {
        float4 *fr = P.fuseRec + ((size_t)slot * P.nseeds + seedI) * 3;
        fr[0] = make_float4(S.normX, S.normY, S.normZ, S.meanDepth);
        fr[1] = make_float4(pw[0], pw[1], pw[2], seedWeight);
        fr[2] = make_float4(seedSize, S.meanIntensity, __uint_as_float(rgb_pack(S.r, S.g, S.b)), __uint_as_float(valid ? 1u : 0u));
    }
    if (ok) {
        float nw[3];
        mul3(F.pose, S.normX, S.normY, S.normZ, nw);
        msl_surfel e;
        e.px = pw[0]; e.py = pw[1]; e.pz = pw[2];
        e.r = S.r; e.g = S.g; e.b = S.b;
        e.nx = nw[0]; e.ny = nw[1]; e.nz = nw[2];
        e.size = seedSize;
        e.color = S.meanIntensity;
        e.weight = seedWeight;
        e.updateTimes = 1;
        e.lastUpdate = F.ref;
        P.cand[(size_t)slot * P.nseeds + seedI] = e;
    }
}

// Image sizes that are not multiples of 8: the strips right of / below the last whole 8x8 cell belong to no cell, so kb_seed_plane does not
// write their texels; this (tiny, rarely launched) kernel does.
__global__ __launch_bounds__(256) void kb_tex_strips(SfDev P) {
    const int slot = blockIdx.y;
    const int wStrip = P.W - P.spW * SP, hStrip = P.H - P.spH * SP;
    const int nRight = wStrip * P.spH * SP, nBottom = P.W * hStrip;   // right strip over the cell rows, bottom strip over the full width
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nRight + nBottom) return;
    int x, y;
    if (i < nRight) { y = i / wStrip; x = P.spW * SP + i % wStrip; }
    else { const int j = i - nRight; y = P.spH * SP + j / P.W; x = j % P.W; }
    const FrameDev F = P.frames[slot];
    const size_t p = (size_t)y * P.W + x;
    P.tex[(size_t)slot * P.pxStride + p] = make_uint2(__float_as_uint(F.depthG()[(size_t)y * P.dstride + x]), P.index[(size_t)slot * P.pxStride + p]);
}

// =============================================================================================
// Map stage (per keyframe, sequential on the map stream)
// =============================================================================================
// "Last workgroup continues" hand-off (cdna_hip_programming.md G16): every workgroup publishes its global stores with an
// agent-scope release, then takes a ticket; the one that draws the last ticket acquires and carries on with the next
// stage inside the same launch, saving a dependent kernel boundary (~5 us each on this latency-critical chain).
__device__ __forceinline__ bool last_workgroup(unsigned *ticket, unsigned *s_flag) {
    // Everything the continuing workgroup reads from this launch is stored write-through (agent-scope atomic stores /
    // RMW atomics) and read back with agent-scope loads, so no L2 write-back fence is needed -- a release fence per
    // workgroup would flush megabytes of freshly dirtied surfel lines 1000 times per launch.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = atomicAdd(ticket, 1u);
        *s_flag = (t == gridDim.x - 1) ? 1u : 0u;
        if (*s_flag) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // reset for the next launch
    }
    __syncthreads();
    return *s_flag != 0;
}
__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// k_fuse (:167-283): ONE WAVE per sub-block of 256 consecutive surfels, no LDS and no workgroup barrier, so a wave starts wherever a
// SIMD has a free slot and registers -- next to the LDS-heavy frame-batched kernels the former 2 KB workgroups waited for LDS
// (kb_seed_plane leaves 4 KB of a CU's 160 KB free) and k_fuse took 20 us in the timed region against 16 us alone.
//   Phase A (streaming): lane l owns the surfels l, 64 + l, 128 + l, 192 + l of the sub-block (four 20-byte hot records; a load instruction
//     covers 64 consecutive records).  Stale / deleted / out of range / out of image surfels finish here; the in-view ones need ONE 8-byte
//     gather each ({depth, superpixel index} texel written by kb_seed_plane) for the occlusion test.  The four gathers of a lane leave
//     together (branch-free, clamped addresses).
//   Hand-over inside the wave: survivor number s (rank by (k, lane) = array order) goes to lane s % 64, round s / 64, with one
//     ds_permute_b32 per k -- a push through the LDS crossbar that allocates no LDS.  Non-survivors push an empty word to the remaining
//     lanes, so every k is a permutation of the 64 lanes and no two lanes ever target the same destination.
//   Phase B (gathers): per round one survivor per lane, neighbouring lanes = neighbouring surfels; its hot record (just streamed: cache
//     hit), 32-byte cold record and 48-byte FuseRec are requested together, so <= 64 survivors per sub-block cost one round trip.
// Deleted slots are handed to k_compact in delU (one atomic per wave that deleted something -- a handful per keyframe); per-sub-block
// deleted / updated counts go to blockSums / blockUpd with plain stores.
// The sub-block -> wave mapping uses the HOST's upper bound of the live count (nSubGrid), so the first loads do not wait for ctr[0].
// int(projectU + 0.5) of :204-205 (a double addition, truncation towards zero) without double arithmetic: for u >= 1/2 it equals
// floor(u) + (u - floor(u) >= 1/2) -- floor and the difference are exact in float --, and for smaller u (or NaN) both expressions are
// <= 0, which the image test (pUInt < 1) rejects whatever the exact value is; the clamp keeps the conversion defined for huge / infinite u.
__device__ __forceinline__ int round_half_up_pixel(float u) {
    const float c = fminf(fmaxf(u, -4.0f), 1.0e6f);   // NaN -> -4
    const float f = floorf(c);
    return (int)f + ((c - f) >= 0.5f ? 1 : 0);
}
__device__ __forceinline__ unsigned lane_rank(unsigned long long m) {   // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Agent-scope (L2-bypassing) forms of the record loads: the dependent waves of a merged launch read sub-blocks that workgroup 0 has just written
// (write-through) from another XCD, and must not be served a line their own L2 fetched before.
__device__ __forceinline__ uint4 ld_agent16(const uint4 *p) {
    const unsigned *q = reinterpret_cast<const unsigned *>(p);
    return make_uint4(ld_agent(q), ld_agent(q + 1), ld_agent(q + 2), ld_agent(q + 3));
}
__device__ __forceinline__ HotRec ld_hot(const MapSoA &M, long long i, bool coh) {
    if (!coh) return M.hot[i];
    const unsigned *q = reinterpret_cast<const unsigned *>(M.hot + i);
    HotRec h;
    h.px = __uint_as_float(ld_agent(q)); h.py = __uint_as_float(ld_agent(q + 1)); h.pz = __uint_as_float(ld_agent(q + 2));
    h.updateTimes = (int)ld_agent(q + 3); h.lastUpdate = (int)ld_agent(q + 4);
    return h;
}
__device__ __forceinline__ ColdRec ld_cold(const MapSoA &M, long long i, bool coh) {
    if (!coh) return M.cold[i];
    const uint4 a = ld_agent16(reinterpret_cast<const uint4 *>(M.cold + i)), b = ld_agent16(reinterpret_cast<const uint4 *>(M.cold + i) + 1);
    ColdRec c;
    c.nx = __uint_as_float(a.x); c.ny = __uint_as_float(a.y); c.nz = __uint_as_float(a.z); c.size = __uint_as_float(a.w);
    c.color = __uint_as_float(b.x); c.weight = __uint_as_float(b.y); c.rgbf = b.z; c._spare = b.w;
    return c;
}

// One fuse wave.  waveIdx / G: this wave's number and the number of fuse waves of the launch (a plain k_fuse launch: blockIdx.x / gridDim.x; a merged
// launch: one less each, workgroup 0 being the compaction).
template <bool MERGED>   // MERGED: the fuse waves of k_fuse_merged (fuseMode 3); the plain kernel does not carry the waiting / past-the-L2 paths
__device__ __forceinline__ void fuse_body(const SfDev &P, int slot, const FrameDev &F, int nSubHint, unsigned waveIdx, int G) {
    const MapSoA &M = P.map;
    const unsigned lane = threadIdx.x;
    const uint2 *tex = P.tex + (size_t)slot * P.pxStride;
    const float4 *fuseRec = P.fuseRec + (size_t)slot * P.nseeds * 3;
    uint8_t *fused = P.fused + (size_t)slot * P.nseeds;
    const int ref = F.ref;
    const float cameraF = (float)(((double)fabsf(P.fx) + (double)fabsf(P.fy)) / 2.0);
    const float halfF = 0.5f * cameraF;   // BASELINE * cameraF (:220), exact
    // Wave g owns sub-block G - 1 - g (the newest surfels -- nearly all in view: most phase-B work -- are dispatched first) and, should the
    // map have outgrown the grid, G - 1 - g + G, ... (grid-stride; normally one iteration).  The grid covers the host's last KNOWN live count
    // plus a margin, not its upper bound (which runs up to 1.5 x ahead between count snapshots).  Sub-blocks below nSubHint load at once; above
    // it the wave reads the live count first and leaves if there is nothing for it.  Capacity is a multiple of 4096 and every sub-block that
    // loads speculatively lies below it, so the 16-byte loads stay in bounds.
    // Modes 1 / 2 (compaction j - 1 runs beside this launch, run_batch): a sub-block is SAFE when compaction j - 1 neither reads nor writes it --
    // it held no deleted slot after keyframe j - 1 (no new surfel or tail element lands in it) and it ends below n - D (the tail moves take
    // their sources from [n - (D - K), n), new surfels are appended from n on).  Mode 1 fuses the safe sub-blocks with the live count as it was
    // before that compaction (all of them lie below it whatever the compaction does); mode 2, launched behind the compaction, the others.
    const int mode = MERGED ? 3 : P.fuseMode;
    long long nBefore = 0, safeEnd = 0;
    if (mode) {
        nBefore = *P.nPubPrev;
        safeEnd = nBefore - (long long)*P.prevDelUCount;
    }
    if (mode == 0 && waveIdx == 0 && lane == 0) *P.nPubOut = P.ctr[0];   // what the next keyframe's mode-1 launch takes as its live count
    bool cohVar = false;   // mode 3: this wave waited for the compaction and reads its sub-block past the L2
#define coh (MERGED && cohVar)
    auto wait_compaction = [&]() -> bool {
        unsigned spins = 0;
        while (ld_agent(P.doneFlag) != P.epoch) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > (1u << 20)) {   // ~0.5 s: workgroup 0 is gone; report instead of hanging the queue
                if (lane == 0) __hip_atomic_store(&P.ctr[5], 31ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
        return true;
    };
    // Workgroups are dispatched round-robin over the 8 XCDs: give each XCD runs of FUSE_CHUNK consecutive sub-blocks (neighbouring surfels
    // project to neighbouring pixels, so an XCD's L2 fetches a part of the texel map instead of all of it; small enough runs keep the XCDs
    // balanced -- whole eighths of the map were 2 x slower).  Dense map: k_fuse 17.4 -> 17.0 us.
    constexpr unsigned FUSE_CHUNK = 16;
    long long lin = waveIdx;
    {
        constexpr unsigned T = 8u * FUSE_CHUNK;
        const unsigned full = ((unsigned)G / T) * T;
        if (waveIdx < full) { const unsigned grp = waveIdx / T, r = waveIdx % T; lin = (long long)grp * T + (r & 7u) * FUSE_CHUNK + (r >> 3); }
    }
    for (long long sb = (long long)G - 1 - lin;; sb += G) {
#ifdef MSL_FUSE_STAMPS   // instrumented experiment builds (tools/fuse_stamps.py): 100 MHz device-clock stamps of every wave in srcOf[]
        const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
#endif
        if constexpr (MERGED) {
            // merged launch: workgroup 0 compacts keyframe j - 1 meanwhile.  A wave whose sub-block that compaction can touch waits for it, then reads
            // the sub-block past its L2 (the compaction's stores are write-through; this XCD must not serve an older copy of the line)
            if (sb * SUB_ITEMS >= nBefore + P.nseeds) return;   // beyond anything the compaction can append
            if (!coh && (sb + 1) * SUB_ITEMS > safeEnd) {       // the end of the array: tail moves, appended surfels
                if (!wait_compaction()) return;
                cohVar = true;
            }
            if (coh && sb * SUB_ITEMS >= __hip_atomic_load(&P.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        } else if (mode == 1) {
            if ((sb + 1) * SUB_ITEMS > safeEnd) return;   // this sub-block and the ones the wave would visit next (+G) belong to the mode-2 launch
        } else if (sb >= nSubHint && sb * SUB_ITEMS >= __hip_atomic_load(&P.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        unsigned prevDeleted = 0;
        if (MERGED && !coh) prevDeleted = P.prevBlockSums[sb];   // (requested together with the hot records)
        if (mode == 2) {   // nearly every wave of this launch leaves here: decide before loading anything
            const bool unsafe = (sb + 1) * SUB_ITEMS > safeEnd || P.prevBlockSums[sb] != 0;
            if (!unsafe) {
                if ((sb + G) * SUB_ITEMS >= P.ctr[0]) return;
                continue;
            }
        } else if (mode == 1) {
            prevDeleted = P.prevBlockSums[sb];   // requested together with the hot records (the few unsafe sub-blocks waste their loads)
        }
        const long long c0 = sb * SUB_ITEMS;
        // lane l owns records l, 64 + l, 128 + l, 192 + l of the sub-block: the survivors' rank order (k, lane) is then the array order, so
        // neighbouring lanes of phase B work on neighbouring records and their gathers and stores share cache lines (round 4: with four
        // CONSECUTIVE records per lane -- five aligned 16-byte loads -- rank neighbours were 4 records apart and every lane of phase B
        // touched lines of its own: 8 L2 requests per fused surfel; k_fuse 22.5 -> 17.7 us on the dense map)
#define REC_LOCAL(k) (64u * (unsigned)(k) + lane)
        HotRec hq[4];
#pragma unroll
        for (int k = 0; k < 4; k++) hq[k] = ld_hot(M, c0 + REC_LOCAL(k), coh);
        if (MERGED && prevDeleted) {   // a sub-block with a hole: the compaction may put a surfel there -- wait for it and load again, past the L2
            if (!wait_compaction()) return;
            cohVar = true;
#pragma unroll
            for (int k = 0; k < 4; k++) hq[k] = ld_hot(M, c0 + REC_LOCAL(k), true);
            prevDeleted = 0;
        }
        if (prevDeleted) continue;   // mode 1: compaction j - 1 puts a surfel into this sub-block; the mode-2 launch fuses it
        const long long n = (mode == 1 || (MERGED && !coh)) ? nBefore : (coh ? __hip_atomic_load(&P.ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P.ctr[0]);
        unsigned w[20];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            w[5 * k] = __float_as_uint(hq[k].px); w[5 * k + 1] = __float_as_uint(hq[k].py); w[5 * k + 2] = __float_as_uint(hq[k].pz);
            w[5 * k + 3] = (unsigned)hq[k].updateTimes; w[5 * k + 4] = (unsigned)hq[k].lastUpdate;
        }
        int state[4];      // 0: nothing to do, 1: stale -> delete, 2: already deleted, 3: in view
        float pzv[4];
        unsigned offT[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const long long i = c0 + REC_LOCAL(k);
            const float x = __uint_as_float(w[5 * k]), y = __uint_as_float(w[5 * k + 1]), z = __uint_as_float(w[5 * k + 2]);
            const int ut = (int)w[5 * k + 3], lu = (int)w[5 * k + 4];
            float pc[4];
            mul4(F.invPose, x, y, z, 1.0f, pc);
            const bool inRange = !(pc[2] < P.fuseNear || pc[2] > P.fuseFar);
            const float zq = inRange ? pc[2] : 1.0f;   // keeps the (unused) quotients of skipped surfels finite
            const float projectU = pc[0] * P.fx / zq + P.cx, projectV = pc[1] * P.fy / zq + P.cy;  // :75-78
            const int pUInt = round_half_up_pixel(projectU), pVInt = round_half_up_pixel(projectV);   // int(projectU + 0.5) wherever it matters
            const bool inImage = !(pUInt < 1 || pUInt > P.W - 2 || pVInt < 1 || pVInt > P.H - 2);
            int st = 0;
            if (i < n) st = (ref - lu > 5 && ut < 5) ? (ut != 0 ? 1 : 2) : (ut == 0 ? 2 : ((inRange && inImage) ? 3 : 0));
            state[k] = st; pzv[k] = pc[2];
            const int pUc = min(max(pUInt, 0), P.W - 1), pVc = min(max(pVInt, 0), P.H - 1);   // always a valid address
            offT[k] = (unsigned)(pVc * P.W + pUc);
        }
        uint2 tx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) tx[k] = tex[offT[k]];
        // a common use of all four results: keeps the compiler from sinking each load into its (conditional) consumer, which would turn one
        // round trip back into up to four dependent ones
        asm volatile("" ::"v"(tx[0].x), "v"(tx[1].x), "v"(tx[2].x), "v"(tx[3].x), "v"(tx[0].y), "v"(tx[1].y), "v"(tx[2].y), "v"(tx[3].y));
#ifdef MSL_FUSE_STAMPS
        const unsigned long long stamp1 = __builtin_amdgcn_s_memrealtime();
#endif
        // ---- classification: deletions of phase A, survivors ----
        bool del[4], surv[4];
        unsigned long long mdel[4];
        unsigned cntDel = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool occluded = state[k] == 3 && (double)pzv[k] < (double)__uint_as_float(tx[k].x) - 1.0;
            if (state[k] == 1 || occluded) M.hot[c0 + REC_LOCAL(k)].updateTimes = 0;
            del[k] = state[k] == 1 || state[k] == 2 || occluded;
            surv[k] = state[k] == 3 && !occluded;
            mdel[k] = __ballot(del[k]);
            cntDel += (unsigned)__popcll(mdel[k]);
        }
        auto hand_over = [&](bool d, unsigned long long m, unsigned base, long long i) {   // append this lane's deleted slot to delU
            if (d) { const unsigned j = base + lane_rank(m); if (j < LIST_D) P.delU[j] = (unsigned)i; }
        };
        if (cntDel) {   // rare: a handful of slots per keyframe
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(P.delUCount, cntDel);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int k = 0; k < 4; k++) { hand_over(del[k], mdel[k], base, c0 + REC_LOCAL(k)); base += (unsigned)__popcll(mdel[k]); }
        }
        // ---- survivors -> (round, lane): one push per k.  word = local index (4 lane + k), valid bit, superpixel << 16 ----
        unsigned rcv[4], bk[4];
        unsigned total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned long long m = __ballot(surv[k]);
            const unsigned c = (unsigned)__popcll(m), rs = lane_rank(m);
            const unsigned dest = (surv[k] ? total + rs : total + c + (lane - rs)) & 63u;
            const unsigned payload = surv[k] ? (REC_LOCAL(k) | 0x100u | (tx[k].y << 16)) : 0u;
            rcv[k] = (unsigned)__builtin_amdgcn_ds_permute((int)(dest * 4u), (int)payload);
            bk[k] = total;
            total += c;
        }
        const unsigned rounds = (total + 63u) >> 6;
        unsigned nupd = 0, cntDelB = 0;
        for (unsigned r = 0; r < rounds; r++) {   // one round for <= 64 survivors (a few per cent of the map are in view; the median sub-block has 14)
            unsigned item = 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned rk = (bk[k] + ((lane - bk[k]) & 63u)) >> 6;   // round of the survivor this lane received from k (if any)
                if ((rcv[k] & 0x100u) && rk == r) item = rcv[k];
            }
            // branch-free loads: a lane without a survivor in this round reads record c0 / seed 0 (valid addresses, one line for all such
            // lanes) -- conditional loads made the compiler sink the first uses into the load block and wait there
            const long long i = c0 + (item & 0xFFu);
            const unsigned sp = item >> 16;
            const HotRec h = ld_hot(M, i, coh);
            ColdRec c = ld_cold(M, i, coh);
            const float4 f0 = fuseRec[3 * sp], f1 = fuseRec[3 * sp + 1], f2 = fuseRec[3 * sp + 2];
            // common use of one field per load instruction: all records are in flight together
            asm volatile("" ::"v"(h.px), "v"(h.lastUpdate), "v"(c.nx), "v"(c.color), "v"(f0.x), "v"(f1.x), "v"(f2.x));
            bool upd = false, delB = false;
            if (item && __float_as_uint(f2.w) != 0u) {   // seed tests of :214-219 (norm != 0, viewCos >= MAX_ANGLE_COS)
                const float seedDepth = f0.w;
                const float pz = ((F.invPose[2] * h.px + F.invPose[6] * h.py) + F.invPose[10] * h.pz) + F.invPose[14] * 1.0f;   // row 2 of mul4: as in phase A
                // :220-221 is (float)((double)(pz pz) / (0.5 (double)cameraF) * 4.0).  Both operands of the division are float values (0.5 cameraF
                // exactly), the multiplication by 4 is exact, and rounding a correctly rounded binary64 quotient of two binary32 numbers to
                // binary32 gives the correctly rounded binary32 quotient (53 >= 2 * 24 + 2: double rounding is innocuous for division), so one
                // IEEE float division yields the same bits as the double expression at a third of the instructions.
                float tolerateDiff = (pz * pz) / halfF * 4.0f;
                tolerateDiff = tolerateDiff < MIN_TOLERATE_DIFF ? (float)MIN_TOLERATE_DIFF : tolerateDiff;
                if (!(pz < seedDepth - tolerateDiff) && !(pz > seedDepth + tolerateDiff)) {
                    float nc[3];
                    mul3(F.invPose, c.nx, c.ny, c.nz, nc);
                    const float normDiffCos = nc[0] * f0.x + nc[1] * f0.y + nc[2] * f0.z;
                    if (normDiffCos < MAX_ANGLE_COS) {
                        M.hot[i].updateTimes = 0;
                        delB = true;
                    } else {
                        const float oldWeight = c.weight;
                        const float newWeight = f1.w;                      // getWeight(seed.meanDepth)
                        const float sumWeight = oldWeight + newWeight;
                        const float fusedPx = (h.px * oldWeight + newWeight * f1.x) / sumWeight;   // f1.xyz = pose * seed.pos
                        const float fusedPy = (h.py * oldWeight + newWeight * f1.y) / sumWeight;
                        const float fusedPz = (h.pz * oldWeight + newWeight * f1.z) / sumWeight;
                        float fusedNx = nc[0] * oldWeight + newWeight * f0.x;
                        float fusedNy = nc[1] * oldWeight + newWeight * f0.y;
                        float fusedNz = nc[2] * oldWeight + newWeight * f0.z;
                        // :254-257: newNormLength is a double that holds a float (std::sqrt(float)); float /= double is a binary64 division
                        // of two float values rounded to float = the IEEE float division (same argument as above)
                        const float newNormLength = sqrtf(fusedNx * fusedNx + fusedNy * fusedNy + fusedNz * fusedNz);
                        fusedNx = fusedNx / newNormLength; fusedNy = fusedNy / newNormLength; fusedNz = fusedNz / newNormLength;
                        float newNormW[3];
                        mul3(F.pose, fusedNx, fusedNy, fusedNz, newNormW);
                        HotRec Hn;
                        Hn.px = fusedPx; Hn.py = fusedPy; Hn.pz = fusedPz; Hn.updateTimes = h.updateTimes + 1; Hn.lastUpdate = ref;
                        c.rgbf = __float_as_uint(f2.z);                    // r, g, b of the seed (bytes: never COLD_WIDE)
                        c.nx = newNormW[0]; c.ny = newNormW[1]; c.nz = newNormW[2];
                        c.weight = sumWeight;
                        c.color = f2.y;                                    // seed.meanIntensity
                        const float newSize = f2.x;                        // seed.size * fabs(meanDepth / (cameraF * viewCos))
                        if (newSize < c.size) c.size = newSize;
                        M.hot[i] = Hn;
                        M.cold[i] = c;
                        fused[sp] = 1;
                        upd = true;
                    }
                }
            }
            nupd += (unsigned)__popcll(__ballot(upd));
            const unsigned long long mb = __ballot(delB);
            if (mb) {   // rare
                const unsigned cb = (unsigned)__popcll(mb);
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(P.delUCount, cb);
                base = __builtin_amdgcn_readfirstlane(base);
                hand_over(delB, mb, base, i);
                cntDelB += cb;
            }
        }
        if (lane == 0) {
            P.blockSums[sb] = cntDel + cntDelB; P.blockUpd[sb] = nupd;
            if (P.updCtr && nupd) atomicAdd(&P.updCtr[sb & 63], nupd);   // merged batches: the compaction wave adds up 64 words instead of one per sub-block
        }
#ifdef MSL_FUSE_STAMPS
        if (lane == 0 && ref == MSL_FUSE_STAMPS) {   // the keyframe with this number only: one in the middle of a batch, co-running kernels and all
            unsigned hwid;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            const unsigned long long stamp2 = __builtin_amdgcn_s_memrealtime();
            uint4 *o = reinterpret_cast<uint4 *>(P.srcOf) + 2 * sb;
            o[0] = make_uint4((unsigned)stamp0, (unsigned)stamp1, (unsigned)stamp2, total);
            o[1] = make_uint4(hwid, xcc, nupd, 0u);
        }
#endif
        // (normally) nothing beyond the grid.  A merged launch's wave that has not waited knows the live count BEFORE the compaction only: what the
        // compaction appends may lie in its next sub-block (the head of the loop decides)
        if ((sb + G) * SUB_ITEMS >= ((MERGED && !coh) ? nBefore + (long long)P.nseeds : n)) return;
    }
#undef coh
}

__device__ __forceinline__ void store_surfel(const MapSoA &M, long long i, const msl_surfel &e) {
    HotRec h; h.px = e.px; h.py = e.py; h.pz = e.pz; h.updateTimes = e.updateTimes; h.lastUpdate = e.lastUpdate;
    ColdRec c; c.nx = e.nx; c.ny = e.ny; c.nz = e.nz; c.size = e.size; c.color = e.color; c.weight = e.weight; c._spare = 0;
    if (rgb_fits(e.r, e.g, e.b)) c.rgbf = rgb_pack(e.r, e.g, e.b);
    else { c.rgbf = COLD_WIDE; *M.wideFlag = 1; M.rgbWide[3 * i] = e.r; M.rgbWide[3 * i + 1] = e.g; M.rgbWide[3 * i + 2] = e.b; }
    M.hot[i] = h; M.cold[i] = c;
}
__device__ __forceinline__ void load_surfel(const MapSoA &M, long long i, const HotRec &h, msl_surfel &e) {
    const ColdRec c = M.cold[i];
    e.px = h.px; e.py = h.py; e.pz = h.pz; e.nx = c.nx; e.ny = c.ny; e.nz = c.nz; e.size = c.size; e.color = c.color;
    if (c.rgbf & COLD_WIDE) { e.r = M.rgbWide[3 * i]; e.g = M.rgbWide[3 * i + 1]; e.b = M.rgbWide[3 * i + 2]; }
    else { e.r = (int)(c.rgbf & 255u); e.g = (int)((c.rgbf >> 8) & 255u); e.b = (int)((c.rgbf >> 16) & 255u); }
    e.weight = c.weight; e.updateTimes = h.updateTimes; e.lastUpdate = h.lastUpdate;
}
__device__ __forceinline__ void move_surfel(const MapSoA &M, long long dst, long long src) {
    const ColdRec c = M.cold[src];
    M.hot[dst] = M.hot[src]; M.cold[dst] = c;
    if (c.rgbf & COLD_WIDE) { M.rgbWide[3 * dst] = M.rgbWide[3 * src]; M.rgbWide[3 * dst + 1] = M.rgbWide[3 * src + 1]; M.rgbWide[3 * dst + 2] = M.rgbWide[3 * src + 2]; }
}

constexpr int TAIL_MAX_HOPS = 64;   // relay hops resolved per hole before the literal loop takes over (k_compact, compact_wave)

// Write-through (agent-scope) forms for the compaction wave of a merged launch: other XCDs' waves read these records in the same launch.
__device__ __forceinline__ void st_agent_f(float *p, float v) { __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void store_hot_wt(const MapSoA &M, long long i, const HotRec &h) {
    unsigned *q = reinterpret_cast<unsigned *>(M.hot + i);
    st_agent(q, __float_as_uint(h.px)); st_agent(q + 1, __float_as_uint(h.py)); st_agent(q + 2, __float_as_uint(h.pz));
    st_agent(q + 3, (unsigned)h.updateTimes); st_agent(q + 4, (unsigned)h.lastUpdate);
}
__device__ __forceinline__ void store_cold_wt(const MapSoA &M, long long i, const ColdRec &c) {
    unsigned *q = reinterpret_cast<unsigned *>(M.cold + i);
    st_agent(q, __float_as_uint(c.nx)); st_agent(q + 1, __float_as_uint(c.ny)); st_agent(q + 2, __float_as_uint(c.nz)); st_agent(q + 3, __float_as_uint(c.size));
    st_agent(q + 4, __float_as_uint(c.color)); st_agent(q + 5, __float_as_uint(c.weight)); st_agent(q + 6, c.rgbf); st_agent(q + 7, c._spare);
}
__device__ __forceinline__ void store_surfel_wt(const MapSoA &M, long long i, const msl_surfel &e) {
    HotRec h; h.px = e.px; h.py = e.py; h.pz = e.pz; h.updateTimes = e.updateTimes; h.lastUpdate = e.lastUpdate;
    ColdRec c; c.nx = e.nx; c.ny = e.ny; c.nz = e.nz; c.size = e.size; c.color = e.color; c.weight = e.weight; c._spare = 0;
    if (rgb_fits(e.r, e.g, e.b)) c.rgbf = rgb_pack(e.r, e.g, e.b);
    else {
        c.rgbf = COLD_WIDE; __hip_atomic_store(M.wideFlag, 1ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_agent(reinterpret_cast<unsigned *>(M.rgbWide + 3 * i), (unsigned)e.r); st_agent(reinterpret_cast<unsigned *>(M.rgbWide + 3 * i + 1), (unsigned)e.g);
        st_agent(reinterpret_cast<unsigned *>(M.rgbWide + 3 * i + 2), (unsigned)e.b);
    }
    store_hot_wt(M, i, h); store_cold_wt(M, i, c);
}
__device__ __forceinline__ void move_surfel_wt(const MapSoA &M, long long dst, long long src, bool cohLoad) {
    const ColdRec c = ld_cold(M, src, cohLoad);
    store_hot_wt(M, dst, ld_hot(M, src, cohLoad)); store_cold_wt(M, dst, c);
    if (c.rgbf & COLD_WIDE)
        for (int q = 0; q < 3; q++) st_agent(reinterpret_cast<unsigned *>(M.rgbWide + 3 * dst + q), ld_agent(reinterpret_cast<const unsigned *>(M.rgbWide + 3 * src + q)));
}

// compact_wave: the whole of k_compact (mode 0) by ONE wave without LDS, for workgroup 0 of a merged launch (k_fuse_merged): it compacts keyframe
// j - 1 while the other workgroups fuse keyframe j.  No LDS and <= 64 VGPRs on purpose: registers and LDS are allocated per kernel, and the fuse
// waves of the same launch must keep fitting the holes the frame-batched kernels leave.  Everything other XCDs read in this launch (the records
// it places or moves, the live count, the published count, the re-armed hand-over counter) is stored write-through, then the flag.
//   D <= LIST_D (the steady state): the hand-over list is rank-sorted through the deleted-slot scratch and kept in registers (4 per lane,
//   ascending; lookups by ds_bpermute).  Larger D (a mispredicted launch; the host uses the two-kernel chain when it expects many deletions):
//   this wave lists the deleted slots of every sub-block that reports any, in order -- correct, slow.
__device__ void compact_wave(const SfDev &P) {
    const unsigned lane = threadIdx.x;
#ifdef MSL_FUSE_STAMPS
    unsigned long long cwst[10]; int cwn = 0;
#define CW_STAMP() cwst[cwn++] = __builtin_amdgcn_s_memrealtime()
#else
#define CW_STAMP()
#endif
    CW_STAMP();
    const MapSoA &M = P.map;
    const int slot = P.prevSlot;
    const uint8_t *candOk = P.candOk + (size_t)slot * P.nseeds, *fused = P.fused + (size_t)slot * P.nseeds;
    const msl_surfel *cand = P.cand + (size_t)slot * P.nseeds;
    // ---- everything that depends on nothing is requested first: counters, the hand-over list, the updated counts, the first flag words ----
    const unsigned dHand = *P.prevDelUCount;
    const long long n = P.ctr[0];
    const bool bad = P.ctr[5] == 20;
    unsigned upd = P.prevUpdCtr[lane];   // updated surfels of keyframe j - 1: 64 hashed partial sums (its fuse waves added them up)
    const long long tot8 = P.ctr[8], tot9 = P.ctr[9], tot10 = P.ctr[10], tot11 = P.ctr[11], tot12 = P.ctr[12];   // running totals (only this wave writes them)
    unsigned du[4];
#pragma unroll
    for (int t = 0; t < 4; t++) du[t] = P.prevDelU[lane + 64 * t];   // (LIST_D entries exist whatever dHand is)
    // flag bytes, 16 consecutive seeds per lane and load (1024 per round), five rounds (5120 seeds: the whole 640 x 480 lattice) requested at once;
    // branch-free (clamped addresses; bytes beyond the lattice are masked when they are looked at)
    constexpr int FR = 5;
    uint4 cwv[FR], fwv[FR];
    const int nRounds16 = (P.nseeds + 1023) >> 10;
    const int lastQuad = ((P.nseeds - 1) >> 4) * 16;   // (nseeds is a multiple of 16 for every lattice the host merges launches for; the tail is masked)
    auto load_flags = [&](int r0) {
#pragma unroll
        for (int q = 0; q < FR; q++) {
            const int sI = min(1024 * (r0 + q) + 16 * (int)lane, lastQuad);
            cwv[q] = *reinterpret_cast<const uint4 *>(candOk + sI);
            fwv[q] = *reinterpret_cast<const uint4 *>(fused + sI);
        }
    };
    auto spawn_bits = [&](const uint4 &c, const uint4 &f, int sI) -> unsigned {
        const unsigned cw[4] = {c.x, c.y, c.z, c.w}, fw[4] = {f.x, f.y, f.z, f.w};
        unsigned bits = 0;
#pragma unroll
        for (int w = 0; w < 4; w++)
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (((cw[w] >> (8 * j)) & 0xFF) && !((fw[w] >> (8 * j)) & 0xFF)) bits |= 1u << (4 * w + j);
        const int left = P.nseeds - sI;   // seeds of this lane inside the lattice
        return left >= 16 ? bits : (left <= 0 ? 0u : bits & ((1u << left) - 1u));
    };
    load_flags(0);
    P.prevUpdCtr[lane] = 0;              // cleared for the slot's next use
    upd = wave_incl_scan(upd);
    upd = (unsigned)__builtin_amdgcn_readlane((int)upd, 63);
    unsigned spawn[FR];   // one bit per seed that spawns a surfel; the 40 flag registers die here, before the sort below needs its own
#pragma unroll
    for (int q = 0; q < FR; q++) spawn[q] = q < nRounds16 ? spawn_bits(cwv[q], fwv[q], 1024 * q + 16 * (int)lane) : 0u;
    asm volatile("" ::"v"(upd), "v"(du[0]), "v"(spawn[0]), "v"(spawn[1]), "v"(spawn[2]), "v"(spawn[3]), "v"(spawn[4]));
    CW_STAMP();   // 1: first loads
    // ---- deleted slots, ascending: rank r lives in lane r & 63, register r >> 6 ----
    long long D = 0;
    unsigned sd[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    const bool regList = dHand <= LIST_D;
    if (dHand <= 64) {
        // the steady state: one entry per lane, sorted inside the wave -- rank = number of smaller entries; a lane without an entry takes a rank
        // behind them, so that the ranks are a permutation and one ds_permute puts every entry in its place (no memory round trip)
        D = dHand;
        const bool has = lane < dHand;
        const unsigned v = has ? du[0] : 0xFFFFFFFFu;
        unsigned rk = 0;
        for (unsigned j = 0; j < dHand; j++) {
            const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)v, (int)j);
            rk += x < v ? 1u : 0u;
        }
        if (!has) rk = lane;   // lanes dHand .. 63 keep their own places (the entries occupy ranks 0 .. dHand - 1)
        sd[0] = (unsigned)__builtin_amdgcn_ds_permute((int)(rk * 4u), (int)v);
    } else if (regList) {
        D = dHand;
        unsigned rk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; t++) if (lane + 64 * t >= dHand) du[t] = 0xFFFFFFFFu;
        for (unsigned j = 0; j < dHand; j++) {   // rank = number of smaller entries (the slots are distinct)
            const unsigned pick = (j >> 6) == 0 ? du[0] : ((j >> 6) == 1 ? du[1] : ((j >> 6) == 2 ? du[2] : du[3]));
            const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)pick, (int)(j & 63));
#pragma unroll
            for (int t = 0; t < 4; t++) rk[t] += x < du[t] ? 1u : 0u;
        }
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (lane + 64 * t < dHand) st_agent(&P.delList[rk[t]], du[t]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (lane + 64 * t < dHand) sd[t] = ld_agent(&P.delList[lane + 64 * t]);
    } else {
        const long long nblk = (n + SUB_ITEMS - 1) / SUB_ITEMS;
        unsigned base = 0;
        for (long long b0 = 0; b0 < nblk; b0 += 64) {
            const long long bb = b0 + lane;
            const unsigned cntb = bb < nblk ? P.prevBlockSums[bb] : 0u;
            unsigned long long nz = __ballot(cntb != 0);
            while (nz) {
                const int src = __builtin_ctzll(nz);
                nz &= nz - 1;
                const long long sb = b0 + src;
                for (int k = 0; k < SUB_ITEMS / 64; k++) {
                    const long long i = sb * SUB_ITEMS + 64 * k + lane;
                    const bool del = i < n && M.hot[i].updateTimes == 0;
                    const unsigned long long m = __ballot(del);
                    if (del) st_agent(&P.delList[base + lane_rank(m)], (unsigned)i);
                    base += (unsigned)__popcll(m);
                }
            }
        }
        D = base;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    CW_STAMP();   // 2: sorted list
    const bool oneReg = D <= 64;
    auto DL = [&](long long j) -> unsigned {   // j-th smallest deleted slot (per-lane j; EVERY lane must call it: a cross-lane read)
        if (!regList) return ld_agent(&P.delList[j]);
        const int srcLane = (int)(j & 63) * 4;
        const unsigned v0 = (unsigned)__builtin_amdgcn_ds_bpermute(srcLane, (int)sd[0]);
        if (oneReg) return v0;
        const unsigned v1 = (unsigned)__builtin_amdgcn_ds_bpermute(srcLane, (int)sd[1]);
        const unsigned v2 = (unsigned)__builtin_amdgcn_ds_bpermute(srcLane, (int)sd[2]), v3 = (unsigned)__builtin_amdgcn_ds_bpermute(srcLane, (int)sd[3]);
        return (j >> 6) == 0 ? v0 : ((j >> 6) == 1 ? v1 : ((j >> 6) == 2 ? v2 : v3));
    };
    // ---- initializeSurfels (:285-331), first half: which seeds spawn a surfel, in seed order.  The flag words of eight rounds travel together and the
    // next eight are requested before these are looked at; the seed index of new surfel k goes to a list (srcOf[k]) so that the second half
    // can work on 64 of them at a time instead of one round after the other.
    long long pos = 0;
    unsigned mySeed = 0;     // seed of new surfel number `lane` while regSeeds (the first 64 travel through registers; the others through srcOf[])
    bool regSeeds = true;
    for (int r0 = 0; r0 < nRounds16; r0 += FR) {
        if (r0) {   // (larger lattices: one more trip per 5120 seeds)
            load_flags(r0);
#pragma unroll
            for (int q = 0; q < FR; q++) spawn[q] = r0 + q < nRounds16 ? spawn_bits(cwv[q], fwv[q], 1024 * (r0 + q) + 16 * (int)lane) : 0u;
        }
#pragma unroll
        for (int q = 0; q < FR; q++) {
            const int sI = 1024 * (r0 + q) + 16 * (int)lane;
            const unsigned bits = spawn[q];
            unsigned long long any = __ballot(bits != 0);
            if (!any) continue;   // most rounds spawn nothing
            if (__popcll(any) <= 12) {
                // a few spawning lanes (the steady state): walked by the whole wave in lockstep, no scan
                while (any) {
                    const int src = __builtin_ctzll(any);
                    any &= any - 1;
                    unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bits, src);
                    while (b) {
                        const unsigned seed = (unsigned)(1024 * (r0 + q) + 16 * src + __builtin_ctz(b));
                        b &= b - 1;
                        if (regSeeds && pos < 64) { if ((long long)lane == pos) mySeed = seed; }
                        else if (lane == 0) st_agent(&P.srcOf[pos], seed);
                        pos++;
                    }
                }
            } else {
                // many (a young map): scanned, every lane lists its own through srcOf[]; what the registers held so far goes there too
                if (regSeeds) { if ((long long)lane < pos) st_agent(&P.srcOf[lane], mySeed); regSeeds = false; }
                const unsigned c = (unsigned)__builtin_popcount(bits);
                const unsigned incl = wave_incl_scan(c);
                long long k = pos + (long long)(incl - c);
                for (unsigned m = bits; m; m &= m - 1) st_agent(&P.srcOf[k++], (unsigned)(sI + __builtin_ctz(m)));
                pos += (long long)(unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
    }
    CW_STAMP();   // 3: flags
    const long long K = pos;
    const long long nAfter = D >= K ? n - (D - K) : n + (K - D);
    // second half: new surfel k -> the k-th largest deleted slot while any remain, else appended (SurfelMapping.cpp:372-384); 64 at a time
    // (the host keeps room for nseeds more surfels before it enqueues a keyframe, so nAfter <= cap; checked all the same)
    const bool place = !bad && (unsigned long long)nAfter <= P.cap;
    if (K > 0) {
        if (K > 64 || !regSeeds) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the list entries kept in memory have arrived)
        for (long long kb = 0; kb < K; kb += 64) {
            const long long k = kb + lane;
            const bool on = k < K;
            const unsigned seed = (kb == 0 && regSeeds) ? mySeed : (on ? ld_agent(&P.srcOf[k]) : 0u);
            const unsigned hole = DL(on && k < D ? D - 1 - k : 0);
            if (on) {
                const msl_surfel e = cand[seed];
                P.newSurfels[k] = e;                    // host-vector mode and debugging read this list
                if (place) store_surfel_wt(M, k < D ? (long long)hole : n + (k - D), e);
            }
        }
    }
    CW_STAMP();   // 4: emission
    // ---- leftover holes: the back-to-front loop of SurfelMapping.cpp:386-390 resolved per hole (k_compact's formulation) ----
    if (place && D > K) {
        const long long R = D - K, nFinal = n - R;
        auto lower = [&](long long x) -> long long {   // first index in the R smallest deleted slots with value >= x (per-lane x)
            long long lo = 0, hi = R;
            while (__ballot(lo < hi)) {   // all lanes step together (DL() is a cross-lane read); a lane that has finished probes a dummy
                const long long mid = lo < hi ? (lo + hi) >> 1 : 0;
                const long long v = (long long)DL(mid);
                if (lo < hi) { if (v < x) lo = mid + 1; else hi = mid; }
            }
            return lo;
        };
        // (lower() and DL() use cross-lane reads: every lane runs the same number of search steps on its own argument, inactive lanes on a dummy)
        const long long cntLow = lower(nFinal);   // uniform argument -> uniform result
        bool overflow = false;
        for (long long a0 = 0; a0 < cntLow && !overflow; a0 += 64) {
            const long long a = a0 + lane;
            long long p = nFinal + (a < cntLow ? a : 0);
            bool chain = a < cntLow;
            for (int hop = 0; hop < TAIL_MAX_HOPS; hop++) {   // uniform trip count; a lane whose chain has ended keeps p
                const long long lb = lower(p);
                const long long held = (long long)DL(lb < R ? lb : 0);   // (every lane: a cross-lane read)
                const bool relay = chain && lb < R && held == p;
                if (relay) p = n - (R - lb); else chain = false;
                if (!__ballot(chain)) break;
            }
            if (__ballot(chain)) { overflow = true; break; }   // pathological chain: literal loop below
            const unsigned dst = DL(a < cntLow ? a : 0);
            if (a < cntLow) move_surfel_wt(M, (long long)dst, p, false);
        }
        if (overflow) {
            // literal back-to-front loop, pathological delete patterns only.  Moves already made above are repeated identically (same source
            // content: a source is never a destination of this formulation), so starting over is safe.
            for (long long i = 1; i <= R; i++) {
                const unsigned hole = DL(R - i);   // (uniform argument)
                const long long src = n - i;
                if (lane == 0 && src != (long long)hole) move_surfel_wt(M, (long long)hole, src, true);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
    }
    CW_STAMP();   // 5: tail
    if (lane == 0) {
        P.ctr[1] = K; P.ctr[2] = D; P.ctr[3] = upd; P.ctr[4] = n; P.ctr[6] = nAfter;
        P.ctr[8] = tot8 + K; P.ctr[9] = tot9 + D; P.ctr[10] = tot10 + upd; P.ctr[11] = tot11 + 1; P.ctr[12] = tot12 + n;
        if (!place && !bad) P.ctr[5] = 20;   // capacity exceeded
        const long long nOut = place ? nAfter : n;
        __hip_atomic_store(&P.ctr[0], nOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(P.nPubCompact, nOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        st_agent(P.resetDelUCountCompact, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every lane's write-through stores have been acknowledged
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) st_agent(P.doneFlag, P.epoch);
#ifdef MSL_FUSE_STAMPS
    CW_STAMP();   // 6: publish
    if (lane == 0 && (P.epoch & 31) == 20) { for (int q = 0; q < cwn; q++) P.delList[4096 + q] = (unsigned)cwst[q]; P.delList[4096 + 15] = (unsigned)K; P.delList[4096 + 14] = (unsigned)D; }
#endif
}

__global__ __launch_bounds__(64) void k_fuse(SfDev P, int slot, FrameDev F, int nSubHint) {   // F by value: kernarg -> SGPRs
    __builtin_amdgcn_s_setprio(3);   // the map chain is sequential per keyframe: issue ahead of the batched kernels' waves
    fuse_body<false>(P, slot, F, nSubHint, blockIdx.x, (int)gridDim.x);
}
// Merged launch: workgroup 0 compacts keyframe j - 1 (compact_wave), workgroups 1 .. G fuse keyframe j (fuseMode 3).
__global__ __launch_bounds__(64) void k_fuse_merged(SfDev P, int slot, FrameDev F, int nSubHint) {
    __builtin_amdgcn_s_setprio(3);
    if (blockIdx.x == 0) { compact_wave(P); return; }
    fuse_body<true>(P, slot, F, nSubHint, blockIdx.x - 1, (int)gridDim.x - 1);
}

// Resident-map compaction (SurfelMapping.cpp:366-391) with prefix sums.  Deleted slots ascending d_0..d_{D-1};
// new surfel k -> d_{D-1-k} while any remain, else appended.  If D > K the literal `while` loop (:386-390) moves,
// at step i = 1..R (R = D-K), the element at position n-i into the i-th largest leftover hole; a hole inside the
// tail [nFinal, n) only relays what lands in it.  So the a-th smallest leftover hole (< nFinal) finally receives
// resolve(nFinal + a), resolve(p) = p if p is live, else resolve(n - rank_desc(p)): a short upward chain.

// k_compact: everything after k_fuse in ONE launch.
//   every workgroup : exclusive scan of the per-chunk deleted counts (each workgroup scans the <= cap/1024 partials itself,
//                     so there is no inter-workgroup dependency), then lists the deleted slots of its own chunks in
//                     ascending order (write-through stores);
//   last workgroup  : initializeSurfels (:285-331) = ordered emission of the seed candidates the fuse step did not consume,
//                     counters, new surfel k -> k-th largest deleted slot else appended, tail sources resolved and moved.
// mode 1 (host-vector drop-in, one workgroup): emission and counters only; the caller compacts (SurfelMapping.cpp:366-391).
constexpr int SMALL_D = 512, SMALL_CHUNKS = 48;   // single-workgroup path: few deletions in few chunks

// LDS is kept to ~3.5 KB: on a GPU saturated by the LDS-heavy batched kernels a larger workgroup waits for a CU to drain.
__global__ __launch_bounds__(256) void k_compact(SfDev P, int slot, int mode) {
    constexpr int NT = 256, TILE = 4 * NT;
    __shared__ unsigned s_wave[33];
    __shared__ unsigned s_dl[SMALL_D];          // single-workgroup paths: the ascending deleted-slot list stays in LDS
    __shared__ unsigned s_raw[LIST_D];          // fastest path: k_fuse's unordered hand-over list
    __shared__ unsigned s_last, s_upd, s_nzChunks, s_base, s_cntChunk;
    __shared__ unsigned s_nzIdx[SMALL_CHUNKS], s_nzCnt[SMALL_CHUNKS], s_nzSortIdx[SMALL_CHUNKS], s_nzSortCnt[SMALL_CHUNKS];   // sub-blocks with deletions
    __shared__ int s_fallback;
    __builtin_amdgcn_s_setprio(3);   // latency-critical serial chain next to the throughput-oriented batched kernels
#ifdef MSL_FUSE_STAMPS
    unsigned long long cst[6];
    cst[0] = __builtin_amdgcn_s_memrealtime();
#endif
    // Steady state (k_fuse handed over <= LIST_D deleted slots): workgroup 0 does everything alone; the others leave after one load
    // instead of fetching the partials and flags as well.
    if (mode == 0 && blockIdx.x != 0 && *P.delUCount <= LIST_D) return;
    // Loads that do not depend on anything are issued first; in particular every workgroup already fetches the seed flags
    // the continuation needs, so the continuing workgroup does not start its dependent chain with a cold memory round trip.
    const uint4 bs0 = *reinterpret_cast<const uint4 *>(P.blockSums + 4 * threadIdx.x);   // first tile of chunk partials
    uint4 bu[4];   // the first 4096 per-workgroup updated counts (arrays are padded by >= 4096 zeroed entries)
#pragma unroll
    for (int q = 0; q < 4; q++) bu[q] = *reinterpret_cast<const uint4 *>(P.blockUpd + TILE * q + 4 * threadIdx.x);
    static_assert(LIST_D == NT, "one hand-over entry per thread");
    const unsigned du = P.delU[threadIdx.x];
    const unsigned dHand = *P.delUCount;   // k_fuse's running total of deleted slots = D of this keyframe
    const long long n = P.ctr[0];
    const bool bad = P.ctr[5] == 20;
    const uint8_t *candOk = P.candOk + (size_t)slot * P.nseeds, *fused = P.fused + (size_t)slot * P.nseeds;
    const int per = (((P.nseeds + NT - 1) / NT) + 3) & ~3;      // seeds per thread, multiple of 4: aligned 32-bit flag loads
    const int s0 = threadIdx.x * per, s1 = min(s0 + per, P.nseeds);
    unsigned cnt = 0;
    unsigned long long emit = 0, emitHi = 0;   // bit j: seed s0 + j spawns a surfel (emit: j < 64; emitHi: 64 <= j < 128)
    const msl_surfel *cand = P.cand + (size_t)slot * P.nseeds;
    const bool pf = blockIdx.x == 0 || mode == 1;   // the workgroup that will emit (steady state / host-vector mode)
    const bool aligned4 = (P.nseeds & 3) == 0 && ((reinterpret_cast<size_t>(candOk) | reinterpret_cast<size_t>(fused)) & 3) == 0;
    // all flag words of the thread in ONE round trip: 8 words each for <= 32 seeds per thread (640 x 480: 19), 24 words for <= 96 (1280 x 960: 76 --
    // round 3 walked the seeds beyond the 64th one by one, two dependent byte loads each, and the kernel took 30 us at that size)
    auto flags_in_one_trip = [&](auto nqTag) {
        constexpr int NQ = decltype(nqTag)::value;
        unsigned cw[NQ], fw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int i = s0 + 4 * q;
            const bool in = 4 * q < per && i < s1;
            cw[q] = in ? *reinterpret_cast<const unsigned *>(candOk + i) : 0u;
            fw[q] = in ? *reinterpret_cast<const unsigned *>(fused + i) : 0u;
        }
#pragma unroll
        for (int q = 0; q < NQ; q += 8)   // (a common use per group of loads keeps them from being sunk into their consumers)
            asm volatile("" ::"v"(cw[q]), "v"(cw[q + 1]), "v"(cw[q + 2]), "v"(cw[q + 3]), "v"(cw[q + 4]), "v"(cw[q + 5]), "v"(cw[q + 6]), "v"(cw[q + 7]),
                         "v"(fw[q]), "v"(fw[q + 1]), "v"(fw[q + 2]), "v"(fw[q + 3]), "v"(fw[q + 4]), "v"(fw[q + 5]), "v"(fw[q + 6]), "v"(fw[q + 7]));
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned e = (s0 + 4 * q + j < s1 && ((cw[q] >> (8 * j)) & 0xFF) && !((fw[q] >> (8 * j)) & 0xFF)) ? 1u : 0u;
                cnt += e;
                if (4 * q + j < 64) emit |= (unsigned long long)e << ((4 * q + j) & 63);
                else emitHi |= (unsigned long long)e << ((4 * q + j - 64) & 63);
            }
    };
    if (per <= 32 && aligned4) {
        flags_in_one_trip(std::integral_constant<int, 8>{});
    } else if (per <= 96 && aligned4) {
        flags_in_one_trip(std::integral_constant<int, 24>{});
    } else {
        for (int i = s0; i < s1; i += 4) {
            unsigned c4, f4;
            if (i + 4 <= P.nseeds && ((reinterpret_cast<size_t>(candOk + i) | reinterpret_cast<size_t>(fused + i)) & 3) == 0) {
                c4 = *reinterpret_cast<const unsigned *>(candOk + i); f4 = *reinterpret_cast<const unsigned *>(fused + i);
            } else {
                c4 = f4 = 0;
                for (int j = 0; j < 4 && i + j < P.nseeds; j++) { c4 |= (unsigned)candOk[i + j] << (8 * j); f4 |= (unsigned)fused[i + j] << (8 * j); }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned e = (i + j < s1 && ((c4 >> (8 * j)) & 0xFF) && !((f4 >> (8 * j)) & 0xFF)) ? 1u : 0u;
                cnt += e;
                if (i + j - s0 < 64) emit |= (unsigned long long)e << (i + j - s0);
                else if (i + j - s0 < 128) emitHi |= (unsigned long long)e << (i + j - s0 - 64);
            }
        }
    }
    // The continuing workgroup of the steady-state path is workgroup 0: it fetches its first two candidate surfels now, so
    // that this round trip overlaps the scans below instead of following them.
    msl_surfel e0, e1;
    memset(&e0, 0, sizeof(e0)); memset(&e1, 0, sizeof(e1));
    if (pf && emit) {
        e0 = cand[s0 + __builtin_ctzll(emit)];
        const unsigned long long m1 = emit & (emit - 1);
        if (m1) e1 = cand[s0 + __builtin_ctzll(m1)];
    }
    const long long nblk = (n + SUB_ITEMS - 1) / SUB_ITEMS;   // sub-block partials written by k_fuse
    const long long nWg = nblk;   // k_fuse waves (blockUpd entries): one per sub-block
    s_raw[threadIdx.x] = du;
    if (threadIdx.x == 0) { s_upd = 0; s_fallback = 0; s_nzChunks = 0; }
    __syncthreads();
#ifdef MSL_FUSE_STAMPS
    cst[1] = __builtin_amdgcn_s_memrealtime();
#endif
    // k_fuse already counted the deleted slots; when they all fit its hand-over list (the steady state) the per-sub-block
    // counts are not needed at all.  Otherwise one pass over them (4 consecutive per thread and tile) lists the sub-blocks
    // that contain deletions.
    const bool fastest = mode == 0 && dHand <= LIST_D;
    unsigned vsum = 0;
    if (!fastest)
        for (long long t0 = 0; t0 < nblk; t0 += TILE) {
            const long long c = t0 + 4 * threadIdx.x;
            const uint4 v4 = t0 == 0 ? bs0 : *reinterpret_cast<const uint4 *>(P.blockSums + c);
            const unsigned x[4] = {c < nblk ? v4.x : 0u, c + 1 < nblk ? v4.y : 0u, c + 2 < nblk ? v4.z : 0u, c + 3 < nblk ? v4.w : 0u};
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (x[j] > 0) {
                    vsum += x[j];
                    const unsigned q = atomicAdd(&s_nzChunks, 1u);
                    if (q < SMALL_CHUNKS) { s_nzIdx[q] = (unsigned)(c + j); s_nzCnt[q] = x[j]; }
                }
        }
    unsigned Dtot, Ku, exUnused, pos;
    block_excl_scan_pair(vsum, cnt, s_wave, &Dtot, &Ku, exUnused, pos);   // total deletions + emission scan
#ifdef MSL_FUSE_STAMPS
    cst[2] = __builtin_amdgcn_s_memrealtime();
#endif
    const long long D = fastest ? (long long)dHand : (long long)Dtot;
    // single-workgroup paths: workgroup 0 does everything alone -- no ticket, no write-through list
    const bool small = mode == 0 && !fastest && D <= SMALL_D && s_nzChunks <= SMALL_CHUNKS;
    const bool single = fastest || small;
    if (single && blockIdx.x != 0) return;
    if (mode == 0 && !bad) {
        if (fastest) {
            if (threadIdx.x < D) {   // rank-sort in LDS
                unsigned r = 0;
                for (unsigned j = 0; j < (unsigned)D; j++) r += s_raw[j] < du ? 1u : 0u;
                s_dl[r] = du;
            }
        } else if (small) {
            // few sub-blocks hold all deletions: order them by index (rank sort); a sub-block's offset in the ascending
            // list is the sum of the counts before it -- no scan over the (thousands of) empty sub-blocks
            const unsigned nz = s_nzChunks;
            if (threadIdx.x < nz) {
                const unsigned me = s_nzIdx[threadIdx.x];
                unsigned r = 0;
                for (unsigned j = 0; j < nz; j++) r += s_nzIdx[j] < me ? 1u : 0u;
                s_nzSortIdx[r] = me; s_nzSortCnt[r] = s_nzCnt[threadIdx.x];
            }
            __syncthreads();
            unsigned base = 0;
            for (unsigned it = 0; it < nz; it++) {
                const long long i0 = (long long)s_nzSortIdx[it] * SUB_ITEMS + threadIdx.x;   // one slot per thread: ascending
                const unsigned f = (i0 < n && P.map.hot[i0].updateTimes == 0) ? 1u : 0u;
                unsigned tt;
                const unsigned w = base + block_excl_scan(f, s_wave, &tt);
                if (f) s_dl[w] = (unsigned)i0;
                base += s_nzSortCnt[it];
            }
        } else {
            // every workgroup lists the deleted slots of its own sub-blocks in ascending order; a sub-block's base offset
            // lives in the registers of the thread that scanned it and is broadcast through one LDS word
            unsigned carry = 0;
            for (long long t0 = 0; t0 < nblk; t0 += TILE) {
                const long long c = t0 + 4 * threadIdx.x;
                const uint4 v4 = t0 == 0 ? bs0 : *reinterpret_cast<const uint4 *>(P.blockSums + c);
                const unsigned v[4] = {c < nblk ? v4.x : 0u, c + 1 < nblk ? v4.y : 0u, c + 2 < nblk ? v4.z : 0u, c + 3 < nblk ? v4.w : 0u};
                unsigned tot;
                const unsigned ex = carry + block_excl_scan(v[0] + v[1] + v[2] + v[3], s_wave, &tot);
                const long long nIter = (min(t0 + TILE, nblk) - t0 - blockIdx.x + gridDim.x - 1) / gridDim.x;
                for (long long it = 0; it < nIter; it++) {
                    const long long b = t0 + blockIdx.x + it * gridDim.x;
                    const int q = (int)(b - t0);
                    if ((int)threadIdx.x == (q >> 2)) {
                        const int comp = q & 3;
                        s_base = ex + (comp > 0 ? v[0] : 0u) + (comp > 1 ? v[1] : 0u) + (comp > 2 ? v[2] : 0u);
                        s_cntChunk = v[comp];
                    }
                    __syncthreads();
                    const unsigned base = s_base, cntChunk = s_cntChunk;
                    if (cntChunk == 0) { __syncthreads(); continue; }   // nothing deleted in this sub-block
                    const long long i0 = b * SUB_ITEMS + threadIdx.x;       // one slot per thread keeps the list ascending
                    const unsigned f = (i0 < n && P.map.hot[i0].updateTimes == 0) ? 1u : 0u;
                    unsigned tt;
                    const unsigned w = base + block_excl_scan(f, s_wave, &tt);   // (its barriers also protect s_base)
                    if (f) st_agent(&P.delList[w], (unsigned)i0);
                }
                carry += tot;
                __syncthreads();
            }
        }
    }
    __syncthreads();
    if (mode == 0 && !single && !last_workgroup(&P.tickets[1], &s_last)) return;
    // ================= continuation: one workgroup =================
    // updated count
    {
        unsigned u = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const long long c = TILE * q + 4 * threadIdx.x;
            u += (c < nWg ? bu[q].x : 0u) + (c + 1 < nWg ? bu[q].y : 0u) + (c + 2 < nWg ? bu[q].z : 0u) + (c + 3 < nWg ? bu[q].w : 0u);
        }
        for (long long c2 = 4 * TILE + threadIdx.x; c2 < nWg; c2 += blockDim.x) u += P.blockUpd[c2];
        u = wave_incl_scan(u);                                   // one LDS atomic per wave instead of 256 on one address
        if ((threadIdx.x & 63) == 63 && u) atomicAdd(&s_upd, u);
    }
    // initializeSurfels (:285-331): thread t owns the contiguous seeds [t*per, (t+1)*per); emission order = seed index order
    const long long K = Ku;
    const long long nAfter = mode == 1 ? n : (D >= K ? n - (D - K) : n + (K - D));
    const bool place = mode == 0 && !bad && (unsigned long long)nAfter <= P.cap;
    auto DL = [&](long long j) -> unsigned { return single ? s_dl[j] : ld_agent(&P.delList[j]); };
    if (cnt) {
        auto emit_one = [&](const msl_surfel &e) {
            const long long k = pos++;
            P.newSurfels[k] = e;                    // host-vector mode and debugging read this list
            if (place)                              // new surfel k -> k-th largest deleted slot while any remain, else appended
                store_surfel(P.map, k < D ? (long long)DL(D - 1 - k) : n + (k - D), e);   // (SurfelMapping.cpp:372-384)
        };
        unsigned long long m = emit;
        for (int j = 0; m; j++, m &= m - 1) {
            const int i = s0 + __builtin_ctzll(m);
            if (pf && j == 0) emit_one(e0);
            else if (pf && j == 1) emit_one(e1);
            else emit_one(cand[i]);
        }
        for (unsigned long long mh = emitHi; mh; mh &= mh - 1) emit_one(cand[s0 + 64 + __builtin_ctzll(mh)]);
        for (int i = s0 + 128; i < s1; i++)
            if (candOk[i] && !fused[i]) emit_one(cand[i]);
    }
    __syncthreads();   // s_upd complete; new-surfel stores ordered before the tail moves below (same workgroup)
#ifdef MSL_FUSE_STAMPS
    cst[3] = __builtin_amdgcn_s_memrealtime();
#endif
    if (P.updCtr && threadIdx.x < 64) P.updCtr[threadIdx.x] = 0;   // merged batches: this keyframe's hashed updated counts (used by compact_wave only) start over
    if (threadIdx.x == 0) {
        P.ctr[1] = K; P.ctr[2] = D; P.ctr[3] = s_upd; P.ctr[4] = n; P.ctr[6] = nAfter;
        // running totals over all keyframes of this handle (one writer per launch, launches are ordered): bench.py derives the
        // per-keyframe averages of a timed region from their differences
        P.ctr[8] += K; P.ctr[9] += D; P.ctr[10] += s_upd; P.ctr[11] += 1; P.ctr[12] += n;
        if ((unsigned long long)nAfter > P.cap) P.ctr[5] = 20;  // capacity exceeded
    }
    if (!place) { if (threadIdx.x == 0) { *P.nPubOut = mode == 1 ? n : nAfter; *P.resetDelUCount = 0; } return; }
    const long long t0 = threadIdx.x, stride = blockDim.x;
    if (D > K) {
        const long long R = D - K, nFinal = n - R;
        auto lower = [&](long long x) -> long long {   // first index in delList[0..R) with value >= x
            long long lo = 0, hi = R;
            while (lo < hi) { const long long mid = (lo + hi) >> 1; if ((long long)DL(mid) < x) lo = mid + 1; else hi = mid; }
            return lo;
        };
        const long long cntLow = lower(nFinal);
        for (long long a = t0; a < cntLow; a += stride) {
            long long p = nFinal + a;
            int hop = 0;
            for (; hop < TAIL_MAX_HOPS; hop++) {
                const long long lb = lower(p);
                if (lb < R && (long long)DL(lb) == p) p = n - (R - lb);   // relay hole: follow to where its content came from
                else break;
            }
            if (hop == TAIL_MAX_HOPS) s_fallback = 1;   // pathological chain: fall back to the literal loop
            P.srcOf[a] = (unsigned)p;
        }
        __syncthreads();   // also orders the new-surfel stores above before the moves below (same workgroup)
        if (s_fallback) {
            if (threadIdx.x == 0)   // literal back-to-front loop (SurfelMapping.cpp:386-390), pathological delete patterns only
                for (long long i = 1; i <= R; i++) {
                    const long long hole = DL(R - i), src = n - i;
                    if (src != hole) move_surfel(P.map, hole, src);
                }
        } else {
            for (long long a = t0; a < cntLow; a += stride) move_surfel(P.map, (long long)DL(a), (long long)P.srcOf[a]);
        }
    }
    if (threadIdx.x == 0) { P.ctr[0] = nAfter; *P.nPubOut = nAfter; *P.resetDelUCount = 0; }   // publish the new live count, re-arm the hand-over list keyframe j + 2 will use
#ifdef MSL_FUSE_STAMPS
    if (threadIdx.x == 0 && (P.ctr[11] & 255) == MSL_FUSE_STAMPS + 1) {
        cst[4] = __builtin_amdgcn_s_memrealtime();
        for (int q = 0; q < 5; q++) P.delList[q] = (unsigned)cst[q];
        P.delList[5] = (unsigned)K; P.delList[6] = (unsigned)D;
    }
#endif
}

// ---- map maintenance: ordered selection of surfels by a predicate -------------------------------
// mode 0: updateTimes > 0 && lastUpdate == arg (moveAddSurfels, src/SurfelMapping.cpp:213)   mode 1: updateTimes >= arg (Stop, :68)
__device__ __forceinline__ bool select_pred(const HotRec &h, int mode, int arg) {
    return mode == 0 ? (h.updateTimes > 0 && h.lastUpdate == arg) : (h.updateTimes >= arg);
}
__global__ __launch_bounds__(256) void k_select_count(SfDev P, int mode, int arg) {
    __shared__ unsigned s_c;
    const long long n = P.ctr[0];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        if (threadIdx.x == 0) s_c = 0;
        __syncthreads();
        unsigned c = 0;
        for (int k = 0; k < SCAN_ITEMS / 256; k++) {
            const long long i = b * SCAN_ITEMS + k * 256 + threadIdx.x;
            if (i < n && select_pred(P.map.hot[i], mode, arg)) c++;
        }
        if (c) atomicAdd(&s_c, c);
        __syncthreads();
        if (threadIdx.x == 0) P.blockSums[b] = s_c;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_select_scan(SfDev P) {   // one workgroup: exclusive scan of the chunk counts, total -> ctr[7]
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
    if (threadIdx.x == 0) P.ctr[7] = carry;
}
__global__ __launch_bounds__(256) void k_select_write(SfDev P, int mode, int arg, msl_surfel *out, int markDeleted) {
    __shared__ unsigned s_wave[17];
    const long long n = P.ctr[0];
    const long long nblk = (n + SCAN_ITEMS - 1) / SCAN_ITEMS;
    for (long long b = blockIdx.x; b < nblk; b += gridDim.x) {
        unsigned base = P.blockSums[b];
        for (int k = 0; k < SCAN_ITEMS / 256; k++) {           // 256 consecutive surfels per round keep the map order
            const long long i = b * SCAN_ITEMS + k * 256 + threadIdx.x;
            HotRec h{};
            const bool sel = i < n && select_pred(h = P.map.hot[i], mode, arg);
            unsigned tot;
            const unsigned pos = base + block_excl_scan(sel ? 1u : 0u, s_wave, &tot);
            if (sel) {
                msl_surfel e;
                load_surfel(P.map, i, h, e);
                out[pos] = e;
                if (markDeleted) P.map.hot[i].updateTimes = 0;   // "Delete the surfel from the local point" (:224)
            }
            base += tot;
        }
    }
}
__global__ void k_add_ctr(long long *ctr, long long add) {
    if (threadIdx.x == 0) { ctr[0] += add; ctr[4] = ctr[0]; ctr[6] = ctr[0]; }
}
__global__ __launch_bounds__(256) void k_aos_to_soa_at(MapSoA M, const msl_surfel *src, long long n, const long long *ctr) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_surfel(M, ctr[0] + i, src[i]);
}

// AoS <-> SoA conversion for upload / download / host-vector mode
__global__ __launch_bounds__(256) void k_aos_to_soa(MapSoA M, const msl_surfel *src, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) store_surfel(M, i, src[i]);
}
__global__ __launch_bounds__(256) void k_soa_to_aos(MapSoA M, msl_surfel *dst, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const HotRec h = M.hot[i];
    msl_surfel e;
    load_surfel(M, i, h, e);
    dst[i] = e;
}
// wide: -1 = leave the wide-rgb flag ctr[13] alone (upload: k_aos_to_soa has just set it if needed), 0 / 1 = the restored snapshot's flag
__global__ void k_set_ctr(long long *ctr, long long n, unsigned *delUCount, int wide) {   // delUCount: the three rotating hand-over counters; ctr[16..18]: the published live counts
    if (threadIdx.x == 0) {
        delUCount[0] = 0; delUCount[1] = 0; delUCount[2] = 0; ctr[16] = n; ctr[17] = n; ctr[18] = n;
        ctr[0] = n; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0; ctr[4] = n; ctr[6] = n; ctr[7] = 0;
        if (wide >= 0) ctr[13] = wide;
    }
}

// Host-vector mode, sparse case: the records this keyframe touched (updated: lastUpdate == ref; deleted: updateTimes == 0) of the sub-blocks that
// report any, as a compact list {index, reference-layout record}.  One wave per sub-block; slots by one atomic per wave.
__global__ __launch_bounds__(64) void k_collect_changed(SfDev P, int ref, long long n, unsigned *count, unsigned *idxOut, msl_surfel *recOut, unsigned capOut) {
    const long long sb = blockIdx.x;
    if (!(P.blockSums[sb] | P.blockUpd[sb])) return;
    const unsigned lane = threadIdx.x;
    for (int k = 0; k < SUB_ITEMS / 64; k++) {
        const long long i = sb * SUB_ITEMS + k * 64 + lane;
        HotRec h; h.updateTimes = 1; h.lastUpdate = ref - 1;
        if (i < n) h = P.map.hot[i];
        const bool ch = i < n && (h.updateTimes == 0 || h.lastUpdate == ref);
        const unsigned long long m = __ballot(ch);
        if (!m) continue;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd(count, (unsigned)__popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        if (ch) {
            const unsigned j = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (j < capOut) { msl_surfel e; load_surfel(P.map, i, h, e); recOut[j] = e; idxOut[j] = (unsigned)i; }
        }
    }
}

__global__ void k_empty(int grid_dummy) { (void)grid_dummy; }
__global__ void k_debug_div100(const float *x, double *out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = div100_exact((double)(x[i] * x[i]));
}

enum { SK_SEED_INIT = 0, SK_ASSIGN, SK_PROP, SK_COMMIT_PX, SK_UPDATE_SEEDS, SK_COMMIT_SEEDS, SK_SEED_PLANE, SK_FUSE, SK_NEW, SK_COMPACT,
       SK_CONVERT, SK_COPY };
const char *kSfNames[MSL_SF_NKERNELS] = {"kb_seed_init", "kb_assign", "kb_prop", "kb_commit_px", "kb_update_seeds", "kb_commit_seeds",
                                         "kb_seed_plane", "k_fuse", "k_empty", "k_compact", "k_convert", "copy"};

}  // namespace

struct msl_sf {
    int device = 0;
    SfDev dev{};
    int maxBatch = 1;              // keyframes per batch; slots = 2 * maxBatch (double-buffered sets)
    hipStream_t preStream = nullptr, mapStream = nullptr; bool ownStreams = true;
    // overlap of compaction j with fusion j + 1: the compaction chain runs on its own stream; kfSerial numbers the keyframes of the handle
    hipStream_t cmpStream = nullptr; std::vector<hipEvent_t> evFuse, evTail; hipEvent_t evCmp = nullptr;
    unsigned long long kfSerial = 0; size_t blkStride = 0; int lastPar = 0;
    hipStream_t copyStream = nullptr;   // host-image mode: the H2D copies of slot set i + 1 run beside the superpixel kernels of set i
    hipEvent_t evH2D[2] = {nullptr, nullptr};
    hipEvent_t evPre[2] = {nullptr, nullptr}, evMap[2] = {nullptr, nullptr}, evCopy[2] = {nullptr, nullptr};
    bool evMapValid[2] = {false, false}, evCopyValid[2] = {false, false};
    unsigned long long batchNo = 0;
    int lastSlot = 0;
    // per-slot device buffers
    FrameDev *d_frames = nullptr; FrameDev *h_frames = nullptr;  // pinned host staging [slots]
    msl_seed *d_seeds = nullptr, *d_seedsTmp = nullptr; msl_surfel *d_cand = nullptr; uint8_t *d_candOk = nullptr, *d_fused = nullptr; uint2 *d_tex = nullptr; float4 *d_fuseRec = nullptr;
    unsigned short *d_index = nullptr, *d_amap = nullptr; unsigned *d_tmin = nullptr; int *d_chunkAbort = nullptr, *d_changed = nullptr;
    AssignRec *d_arec = nullptr; unsigned *d_wl = nullptr, *d_wlCount = nullptr;
    float *d_pxInv = nullptr;
    // staged images (host input mode), per slot
    uint8_t *d_gray = nullptr; float *d_depth = nullptr; int32_t *d_member = nullptr;
    size_t grayCap = 0, depthCap = 0, memberCap = 0;  // bytes per slot
    long long *d_ctr = nullptr; long long *h_ctr = nullptr;
    unsigned *d_tickets = nullptr, *d_delU = nullptr, *d_updCtr = nullptr;
    float *d_projTab = nullptr;
    bool propLds = false;        // t(s) of one keyframe fits the LDS: single-launch relaxation
    msl_surfel *d_new = nullptr;
    float *d_mapStore = nullptr; size_t mapCap = 0;
    size_t liveBound = 0;        // host-side upper bound of the live count: last known count + nseeds per keyframe enqueued since
    size_t liveKnown = 0;        // the most recent live count the host has seen (exact at that time; only a hint for k_fuse's speculative loads)
    unsigned long long liveKnownKf = 0;   // ... and the number of keyframes that had been enqueued when it was exact: an older snapshot never replaces a newer one
    // asynchronous refresh of that bound: after every batch the live count is copied to pinned memory behind an event; a later call picks
    // up whatever has arrived, so the bound follows the real count a couple of batches late instead of forcing a pipeline drain
    // every capacity / nseeds keyframes
    static constexpr int NSNAP = 4;
    long long *h_snap = nullptr; hipEvent_t snapEv[NSNAP] = {}; unsigned long long snapKf[NSNAP] = {}; bool snapBusy[NSNAP] = {};
    unsigned long long kfEnq = 0; int snapNext = 0;
    unsigned *d_blockSums = nullptr, *d_blockUpd = nullptr, *d_delList = nullptr, *d_srcOf = nullptr;
    msl_surfel *d_aos = nullptr; size_t aosCap = 0;
    float *d_snapStore = nullptr; size_t snapCap = 0, snapN = 0; bool snapValid = false, snapWide = false;   // msl_sf_map_snapshot / _restore
    // host-vector mode (msl_sf_fuse_ex): the device map equals the caller's vector as the last call left it
    bool mirrorValid = false; size_t mirrorN = 0;
    unsigned *h_blk = nullptr; size_t blkCap = 0;   // pinned: per-sub-block deleted / updated counts of the call's k_fuse launch
    uint8_t *h_list = nullptr; size_t listCap = 0;  // pinned: {count | indices | records} of the sparse download
    KernelProfiler prof;
};

namespace {

void set_map_ptrs(msl_sf *h) {
    const size_t c = h->mapCap;
    MapSoA &M = h->dev.map;
    M.hot = reinterpret_cast<HotRec *>(h->d_mapStore);                  // [cap] 20-byte records
    M.cold = reinterpret_cast<ColdRec *>(h->d_mapStore + 5 * c);        // [cap] 32-byte records (cap is a multiple of 4096: 32-byte aligned)
    M.rgbWide = reinterpret_cast<int *>(h->d_mapStore + 13 * c);        // [cap][3] exact ints of the COLD_WIDE records (untouched otherwise)
    M.wideFlag = h->d_ctr + 13;
    h->dev.cap = c;
    h->dev.blockSums = h->d_blockSums; h->dev.blockUpd = h->d_blockUpd; h->dev.delList = h->d_delList; h->dev.srcOf = h->d_srcOf;
}

// The asynchronous live-count snapshots only ever LOWER liveBound; whenever the map is replaced from outside the keyframe chain
// (upload, restore) the ones still pending describe the old map and must be ignored.
void drop_live_snapshots(msl_sf *h) {
    for (int i = 0; i < msl_sf::NSNAP; i++) h->snapBusy[i] = false;
}

int sync_all(msl_sf *h) {
    if (h->ownStreams && h->copyStream) MSL_HIP_TRY(hipStreamSynchronize(h->copyStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->preStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    if (h->cmpStream) MSL_HIP_TRY(hipStreamSynchronize(h->cmpStream));   // (every batch ends with the map stream waiting for it: normally idle already)
    return MSL_OK;
}

// (Re)allocate the resident map for `cap` surfels, preserving the first `keep` entries.
int map_realloc(msl_sf *h, size_t cap, size_t keep) {
    cap = (cap + 4095) & ~(size_t)4095;
    float *nstore = nullptr; unsigned *nbs = nullptr, *nbu = nullptr, *ndl = nullptr, *nso = nullptr;
    auto attempt = [&]() -> int {
        MSL_HIP_TRY(hipMalloc(&nstore, sizeof(float) * 16 * cap));
        const size_t bst = cap / SUB_ITEMS + 4100;   // per slot; >= 1024 / 4096 padding entries: the compaction reads its first tiles unconditionally
        MSL_HIP_TRY(hipMalloc(&nbs, sizeof(unsigned) * 3 * bst));   // three rotating slots each (keyframe j % 3)
        MSL_HIP_TRY(hipMalloc(&nbu, sizeof(unsigned) * 3 * bst));
        MSL_HIP_TRY(hipMemset(nbs, 0, sizeof(unsigned) * 3 * bst));
        MSL_HIP_TRY(hipMemset(nbu, 0, sizeof(unsigned) * 3 * bst));
        MSL_HIP_TRY(hipMalloc(&ndl, sizeof(unsigned) * cap));
        MSL_HIP_TRY(hipMemset(ndl, 0, sizeof(unsigned) * 256));   // (instrumented builds accumulate section counters in its first words)
        MSL_HIP_TRY(hipMalloc(&nso, sizeof(unsigned) * cap));
        if (keep && h->d_mapStore) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            MSL_HIP_TRY(hipMemcpy(nstore, h->d_mapStore, sizeof(HotRec) * keep, hipMemcpyDeviceToDevice));
            MSL_HIP_TRY(hipMemcpy(nstore + 5 * cap, h->d_mapStore + 5 * h->mapCap, sizeof(ColdRec) * keep, hipMemcpyDeviceToDevice));
            MSL_HIP_TRY(hipMemcpy(nstore + 13 * cap, h->d_mapStore + 13 * h->mapCap, sizeof(int) * 3 * keep, hipMemcpyDeviceToDevice));   // (rare path: no need to know whether any record is wide)
        }
        return MSL_OK;
    };
    const int arc = attempt();
    if (arc != MSL_OK) {   // nothing of a failed attempt stays allocated; the old map is untouched
        if (nstore) (void)hipFree(nstore);
        if (nbs) (void)hipFree(nbs);
        if (nbu) (void)hipFree(nbu);
        if (ndl) (void)hipFree(ndl);
        if (nso) (void)hipFree(nso);
        return arc;
    }
    if (h->d_mapStore) {
        (void)hipFree(h->d_mapStore); (void)hipFree(h->d_blockSums); (void)hipFree(h->d_blockUpd); (void)hipFree(h->d_delList); (void)hipFree(h->d_srcOf);
    }
    h->d_mapStore = nstore; h->d_blockSums = nbs; h->d_blockUpd = nbu; h->d_delList = ndl; h->d_srcOf = nso; h->mapCap = cap;
    h->blkStride = cap / SUB_ITEMS + 4100;
    set_map_ptrs(h);
    return MSL_OK;
}

void free_slots(msl_sf *h) {
    auto F = [](auto *&p) { if (p) { (void)hipFree(p); p = nullptr; } };
    F(h->d_frames); F(h->d_seeds); F(h->d_seedsTmp); F(h->d_cand); F(h->d_candOk); F(h->d_fused); F(h->d_tex); F(h->d_fuseRec); F(h->d_index); F(h->d_amap); F(h->d_tmin);
    F(h->d_chunkAbort); F(h->d_changed); F(h->d_arec); F(h->d_pxInv); F(h->d_wl); F(h->d_wlCount); F(h->d_gray); F(h->d_depth); F(h->d_member);
    if (h->h_frames) { (void)hipHostFree(h->h_frames); h->h_frames = nullptr; }
    h->grayCap = h->depthCap = h->memberCap = 0;
}

int alloc_slots(msl_sf *h, int maxBatch) {
    free_slots(h);
    SfDev &D = h->dev;
    const size_t slots = 2 * (size_t)maxBatch, ns = D.nseeds, npx = D.pxStride;
    MSL_HIP_TRY(hipMalloc(&h->d_frames, sizeof(FrameDev) * slots));
    MSL_HIP_TRY(hipHostMalloc(&h->h_frames, sizeof(FrameDev) * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_seeds, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_seedsTmp, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_cand, sizeof(msl_surfel) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_candOk, ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_fused, ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_tex, sizeof(uint2) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_fuseRec, sizeof(float4) * 3 * ns * slots));
    MSL_HIP_TRY(hipMemset(h->d_tex, 0, sizeof(uint2) * npx * slots));
    MSL_HIP_TRY(hipMemset(h->d_fuseRec, 0, sizeof(float4) * 3 * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_index, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_amap, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_tmin, sizeof(unsigned) * ns * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_arec, sizeof(AssignRec) * (ns * slots + 2)));   // + a record either side: kb_assign loads row pairs that may start one before / end one after
    MSL_HIP_TRY(hipMalloc(&h->d_pxInv, sizeof(float) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_wl, sizeof(unsigned) * npx * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_wlCount, sizeof(unsigned) * slots));
    MSL_HIP_TRY(hipMemset(h->d_wlCount, 0, sizeof(unsigned) * slots));
    MSL_HIP_TRY(hipMemset(h->d_arec, 0, sizeof(AssignRec) * (ns * slots + 2)));
    MSL_HIP_TRY(hipMalloc(&h->d_chunkAbort, sizeof(int) * 32 * slots));
    MSL_HIP_TRY(hipMalloc(&h->d_changed, sizeof(int) * 8 * slots));
    MSL_HIP_TRY(hipMemset(h->d_seeds, 0, sizeof(msl_seed) * ns * slots));
    MSL_HIP_TRY(hipMemset(h->d_index, 0, sizeof(unsigned short) * npx * slots));
    MSL_HIP_TRY(hipMemset(h->d_fused, 0, ns * slots));
    MSL_HIP_TRY(hipMemset(h->d_candOk, 0, ns * slots));
    D.frames = h->d_frames; D.seeds = h->d_seeds; D.seedsTmp = h->d_seedsTmp; D.cand = h->d_cand; D.candOk = h->d_candOk; D.fused = h->d_fused; D.tex = h->d_tex; D.fuseRec = h->d_fuseRec;
    D.index = h->d_index; D.amap = h->d_amap; D.tmin = h->d_tmin; D.chunkAbort = h->d_chunkAbort; D.changed = h->d_changed;
    D.arec = h->d_arec + 1; D.pxInv = h->d_pxInv; D.wl = h->d_wl; D.wlCount = h->d_wlCount;
    h->maxBatch = maxBatch;
    h->lastSlot = 0;            // the debug accessors must never index beyond the reallocated slot buffers
    h->evMapValid[0] = h->evMapValid[1] = false;
    h->evCopyValid[0] = h->evCopyValid[1] = false;
    return MSL_OK;
}

int read_ctr(msl_sf *h) {
    if (h->ownStreams && h->copyStream) MSL_HIP_TRY(hipStreamSynchronize(h->copyStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->preStream));
    MSL_HIP_TRY(hipMemcpyAsync(h->h_ctr, h->d_ctr, sizeof(long long) * 16, hipMemcpyDeviceToHost, h->mapStream));
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    h->prof.drain();
    h->liveBound = (size_t)h->h_ctr[0];   // both streams are idle: the count is exact
    h->liveKnown = h->liveBound; h->liveKnownKf = h->kfEnq;
    for (int i = 0; i < msl_sf::NSNAP; i++) h->snapBusy[i] = false;   // (their events have fired: the stream is idle)
    return MSL_OK;
}

int check_err(msl_sf *h) {
    if (h->h_ctr[5]) {
        const long long e = h->h_ctr[5];
        (void)hipMemsetAsync(h->d_ctr + 5, 0, sizeof(long long), h->mapStream);
        if (e == 20) set_error("resident surfel map capacity exceeded (reserve more with msl_sf_map_reserve)");
        else set_error("surfel pipeline device-side bound exceeded (code %lld)", e);
        return MSL_ERR_OVERFLOW;
    }
    return MSL_OK;
}

// When kernel `kid` is being timed its dispatch carries its own start/stop events (hipExtLaunchKernelGGL), so the
// measurement adds no extra packets to the stream.  (Cross-checked once against in-kernel 100 MHz device-clock stamps:
// 64.3 us by events vs 62.1 us by stamps for the same launches.)
#define LAUNCH_LDS(kid, st, kern, grid, block, lds, ...)                                               \
    do {                                                                                               \
        hipEvent_t _ea, _eb;                                                                           \
        if (h->prof.kernel_pair(kid, &_ea, &_eb))                                                      \
            hipExtLaunchKernelGGL(kern, grid, block, lds, st, _ea, _eb, 0, __VA_ARGS__);               \
        else                                                                                           \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                               \
    } while (0)
#define LAUNCH(kid, st, kern, grid, block, ...) LAUNCH_LDS(kid, st, kern, grid, block, 0, __VA_ARGS__)

// Superpixel stage for slots [slot0, slot0+n) on the pre stream, then the map stage per keyframe on the map stream.
int run_batch(msl_sf *h, int n, const int32_t *refs, const uint8_t *gray, size_t gs, size_t gfs, const float *depth, size_t ds, size_t dfs,
              const int32_t *member, size_t ms, size_t mfs, msl_mem mem, const float *poses, bool compact) {
    SfDev &D = h->dev;
    const int W = D.W, H = D.H;
    if (n < 1 || n > h->maxBatch) { set_error("msl_sf: batch of %d keyframes exceeds the batch capacity %d", n, h->maxBatch); return MSL_ERR_INVALID; }
    if (!gray || !depth || !member || !poses || !refs || gs < (size_t)W || ds < (size_t)W * 4 || ms < (size_t)((W + 1) / 2) * 4 || (ds & 3) || (ms & 3)) {
        set_error("msl_sf: bad image pointers or strides");
        return MSL_ERR_INVALID;
    }
    if (compact) {
        h->mirrorValid = false;   // the resident map moves on without the host-vector caller
        // The reference's mvLocalSurfels is an unbounded std::vector (include/Map.h:130): grow the resident map before a batch could
        // overflow it.  Every keyframe adds at most nseeds surfels, so the host only needs an upper bound of the live count; the
        // exact count is read back (one sync) only when that bound reaches the capacity.
        const size_t need = (size_t)n * (size_t)D.nseeds;
        for (int i = 0; i < msl_sf::NSNAP; i++)
            if (h->snapBusy[i] && hipEventQuery(h->snapEv[i]) == hipSuccess) {
                h->snapBusy[i] = false;
                const size_t cand = (size_t)h->h_snap[i] + (size_t)(h->kfEnq - h->snapKf[i]) * (size_t)D.nseeds;   // count then + what was enqueued since
                if (cand < h->liveBound) h->liveBound = cand;
                if (h->snapKf[i] >= h->liveKnownKf) { h->liveKnown = (size_t)h->h_snap[i]; h->liveKnownKf = h->snapKf[i]; }   // completed snapshots are visited in array order, not age order
            }
        if (h->liveBound + need > h->mapCap) {
            int rc = read_ctr(h);
            if (rc != MSL_OK) return rc;
            rc = check_err(h);
            if (rc != MSL_OK) return rc;
            if (h->liveBound + need > h->mapCap) {
                rc = map_realloc(h, 2 * h->liveBound + 2 * need + 65536, h->liveBound);
                if (rc != MSL_OK) return rc;
            }
        }
        h->liveBound += need;
        h->kfEnq += (unsigned long long)n;
    }
    const int set = (int)(h->batchNo & 1), slot0 = set * h->maxBatch;
    hipStream_t sp = h->preStream, sm = h->mapStream;
    if (h->evMapValid[set] && sp != sm) MSL_HIP_TRY(hipStreamWaitEvent(sp, h->evMap[set], 0));   // the set's previous user is done
    D.gstride = gs; D.gbytes = gs * (size_t)(H - 1) + W; D.dstride = ds / 4; D.mstride = ms / 4;
    // bytes actually present in the caller's buffers: the last row carries no stride padding
    const size_t gb = gs * (size_t)(H - 1) + W, db = ds * (size_t)(H - 1) + (size_t)W * 4, mb = ms * (size_t)((H + 1) / 2 - 1) + (size_t)((W + 1) / 2) * 4;   // the membership image is ceil(H / 2) x ceil(W / 2) (PlaneDetection's cloud size)
    if (mem == MSL_MEM_HOST) {
        const size_t slots = 2 * (size_t)h->maxBatch;
        if (gb > h->grayCap || db > h->depthCap || mb > h->memberCap) {
            int rc = sync_all(h);
            if (rc != MSL_OK) return rc;
            if (h->d_gray) (void)hipFree(h->d_gray);
            if (h->d_depth) (void)hipFree(h->d_depth);
            if (h->d_member) (void)hipFree(h->d_member);
            h->d_gray = nullptr; h->d_depth = nullptr; h->d_member = nullptr;
            MSL_HIP_TRY(hipMalloc(&h->d_gray, gb * slots)); MSL_HIP_TRY(hipMalloc(&h->d_depth, db * slots)); MSL_HIP_TRY(hipMalloc(&h->d_member, mb * slots));
            h->grayCap = gb; h->depthCap = db; h->memberCap = mb;
        }
        // The images travel on their own stream so that they overlap the superpixel kernels of the previous call (the other slot set);
        // with caller-provided streams (msl_sf_set_stream) everything stays on that one stream.
        hipStream_t sc = (h->ownStreams && h->copyStream) ? h->copyStream : sp;
        if (sc != sp && h->evMapValid[set]) MSL_HIP_TRY(hipStreamWaitEvent(sc, h->evMap[set], 0));   // the set's previous user is done
        h->prof.begin(SK_COPY, sc);
        // Tightly packed frame arrays (the streaming case) travel as ONE copy per image kind instead of one per frame; a membership image
        // shared by all keyframes of the call (member_frame_stride == 0) is staged once.
        const bool packedG = n > 1 && gfs == gb && h->grayCap == gb, packedD = n > 1 && dfs == db && h->depthCap == db;
        const bool packedM = n > 1 && mfs == mb && h->memberCap == mb;
        if (packedG) MSL_HIP_TRY(hipMemcpyAsync(h->d_gray + (size_t)slot0 * h->grayCap, gray, gb * (size_t)n, hipMemcpyHostToDevice, sc));
        if (packedD) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_depth + (size_t)slot0 * h->depthCap, depth, db * (size_t)n, hipMemcpyHostToDevice, sc));
        if (packedM) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + (size_t)slot0 * h->memberCap, member, mb * (size_t)n, hipMemcpyHostToDevice, sc));
        if (mfs == 0) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + (size_t)slot0 * h->memberCap, member, mb, hipMemcpyHostToDevice, sc));
        for (int f = 0; f < n; f++) {
            const size_t s = slot0 + f;
            if (!packedG) MSL_HIP_TRY(hipMemcpyAsync(h->d_gray + s * h->grayCap, gray + f * gfs, gb, hipMemcpyHostToDevice, sc));
            if (!packedD) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_depth + s * h->depthCap, (const uint8_t *)depth + f * dfs, db, hipMemcpyHostToDevice, sc));
            if (!packedM && mfs != 0) MSL_HIP_TRY(hipMemcpyAsync((uint8_t *)h->d_member + s * h->memberCap, (const uint8_t *)member + f * mfs, mb, hipMemcpyHostToDevice, sc));
        }
        h->prof.end(sc);
        if (sc != sp) {
            MSL_HIP_TRY(hipEventRecord(h->evH2D[set], sc));
            MSL_HIP_TRY(hipStreamWaitEvent(sp, h->evH2D[set], 0));
        }
    }
    if (h->evCopyValid[set]) MSL_HIP_TRY(hipEventSynchronize(h->evCopy[set]));   // pinned staging of this set is free again
    for (int f = 0; f < n; f++) {
        FrameDev &F = h->h_frames[slot0 + f];
        if (mem == MSL_MEM_HOST) {
            const size_t s = slot0 + f;
            F.gray = h->d_gray + s * h->grayCap; F.depth = (const float *)((uint8_t *)h->d_depth + s * h->depthCap);
            F.member = (const int32_t *)((uint8_t *)h->d_member + (mfs == 0 ? (size_t)slot0 : s) * h->memberCap);
        } else {
            F.gray = gray + f * gfs; F.depth = (const float *)((const uint8_t *)depth + f * dfs); F.member = (const int32_t *)((const uint8_t *)member + f * mfs);
        }
        memcpy(F.pose, poses + 16 * f, sizeof(float) * 16);
        inverse4<float>(F.pose, F.invPose);   // pose.inverse() (:59), adjugate/determinant in float
        F.ref = refs[f]; F._pad = 0;
    }
    MSL_HIP_TRY(hipMemcpyAsync(h->d_frames + slot0, h->h_frames + slot0, sizeof(FrameDev) * n, hipMemcpyHostToDevice, sp));
    MSL_HIP_TRY(hipEventRecord(h->evCopy[set], sp));
    h->evCopyValid[set] = true;

    SfDev P = D;
    // shift every per-slot base so that blockIdx.y/z == 0 addresses slot0
    P.frames = D.frames + slot0; P.seeds = D.seeds + (size_t)slot0 * D.nseeds; P.seedsTmp = D.seedsTmp + (size_t)slot0 * D.nseeds;
    P.cand = D.cand + (size_t)slot0 * D.nseeds; P.candOk = D.candOk + (size_t)slot0 * D.nseeds; P.fused = D.fused + (size_t)slot0 * D.nseeds;
    P.tex = D.tex + (size_t)slot0 * D.pxStride; P.fuseRec = D.fuseRec + (size_t)slot0 * D.nseeds * 3;
    P.index = D.index + (size_t)slot0 * D.pxStride; P.amap = D.amap + (size_t)slot0 * D.pxStride; P.tmin = D.tmin + (size_t)slot0 * D.nseeds;
    P.arec = D.arec + (size_t)slot0 * D.nseeds; P.pxInv = D.pxInv + (size_t)slot0 * D.pxStride; P.wl = D.wl + (size_t)slot0 * D.pxStride; P.wlCount = D.wlCount + slot0;
    P.chunkAbort = D.chunkAbort + slot0 * 32; P.changed = D.changed + slot0 * 8;
    const unsigned un = (unsigned)n;
    const dim3 seedGrid((D.nseeds + 255) / 256, un);
    const int nbx = ((W - 5) >> 3) + 2, nby = ((H - 5) >> 3) + 2;   // dual cells [8 b + 4, 8 b + 12), b from -1, that meet the image
    const dim3 pxGrid(xcd_grid(((nbx + 3) / 4) * ((nby + ASSIGN_NY - 1) / ASSIGN_NY), n)), flatPx(xcd_grid(((D.npx + 7) / 8 + 255) / 256, n));
    LAUNCH(SK_SEED_INIT, sp, kb_seed_init, seedGrid, dim3(256), P);
    for (int it = 0; it < 3; it++) {
        LAUNCH(SK_ASSIGN, sp, kb_assign, pxGrid, dim3(256), P, it, n, nbx, nby);
        if (it > 0) {
            h->prof.begin(SK_PROP, sp);
            if (h->propLds) {
                hipLaunchKernelGGL(kb_prop_lds, dim3(un), dim3(256), sizeof(unsigned) * D.nseeds, sp, P);
            } else {
                for (int r = 0; r < PROP_ROUNDS; r++) hipLaunchKernelGGL(kb_prop, dim3(xcd_grid(PROP_BLOCKS, n)), dim3(256), 0, sp, P, r, n);
                hipLaunchKernelGGL(kb_prop_finish, dim3(un), dim3(1024), 0, sp, P);
            }
            h->prof.end(sp);
            LAUNCH(SK_COMMIT_PX, sp, kb_commit_px, flatPx, dim3(256), P, n);
        }
        if ((W % SP) >= 1 && (W % SP) <= 3) LAUNCH(SK_UPDATE_SEEDS, sp, kb_update_seeds<true>, dim3(xcd_grid((D.nseeds + 15) / 16, n)), dim3(256), P, it, n);
        else LAUNCH(SK_UPDATE_SEEDS, sp, kb_update_seeds<false>, dim3(xcd_grid((D.nseeds + 15) / 16, n)), dim3(256), P, it, n);
        LAUNCH(SK_COMMIT_SEEDS, sp, kb_commit_seeds, seedGrid, dim3(256), P, it);
    }
    // 4 KB of (unused) dynamic LDS cap the kernel at 8 waves per CU (it could run 11).  Measured on the whole front end (round 3, same box,
    // alternating runs): 11 waves 19.6 k frames/s, 10 waves 20.6-20.9 k, 9 waves 21.0-21.1 k, 8 waves 21.3-21.5 k, 7 waves 20.3-20.9 k -- the
    // kernel alone is no slower with fewer waves (its waves are VALU-latency bound), and the wave slots, registers and LDS it leaves go to the
    // ORB kernels and to the map stage's k_fuse / k_compact (3.3 KB LDS) that run beside it.
    constexpr unsigned planePad = 4096;
    if ((W % SP) >= 1 && (W % SP) <= 3) LAUNCH_LDS(SK_SEED_PLANE, sp, kb_seed_plane<true>, dim3(xcd_grid(((D.spW + 1) / 2) * ((D.spH + 1) / 2), n)), dim3(64), planePad, P, n);
    else LAUNCH_LDS(SK_SEED_PLANE, sp, kb_seed_plane<false>, dim3(xcd_grid(((D.spW + 1) / 2) * ((D.spH + 1) / 2), n)), dim3(64), planePad, P, n);
    if ((W % SP) || (H % SP)) {   // pixels outside the whole cells (sizes that are not multiples of 8)
        const int nStrip = (W - D.spW * SP) * D.spH * SP + W * (H - D.spH * SP);
        hipLaunchKernelGGL(kb_tex_strips, dim3((unsigned)((nStrip + 255) / 256), un), dim3(256), 0, sp, P);
    }
    if (sp != sm) {
        MSL_HIP_TRY(hipEventRecord(h->evPre[set], sp));
        MSL_HIP_TRY(hipStreamWaitEvent(sm, h->evPre[set], 0));
    }
    const size_t boundLive = compact ? h->liveBound : h->mapCap;
    // grid: the last known live count plus a margin (k_fuse is grid-stride, so a map that outgrew it is still covered), never beyond the upper
    // bound; hint: the sub-blocks that were full at the last known count load without waiting for the live count
    const size_t known = std::min(h->liveKnown, boundLive);
    const int nSubGrid = (int)std::max<size_t>(1, (std::min(known + 2 * (size_t)D.nseeds, boundLive) + SUB_ITEMS - 1) / SUB_ITEMS);
    const int nSubHint = (int)(known / SUB_ITEMS);
    // Optional overlap (MSL_SF_OVERLAP=1; resident mode, own streams): keyframe f's fusion does not wait for keyframe f - 1's compaction.  The
    // compaction only touches sub-blocks that held a deleted slot and the end of the array, so
    //   map stream : fuse_0 (all) | fuse_1 (safe sub-blocks) | [tail_1 done] fuse_2 (safe) | [tail_2 done] fuse_3 (safe) ...
    //   cmp stream : [fuse_0 done] compact_0, tail_1 (the other sub-blocks of keyframe 1) | [fuse_1 done] compact_1, tail_2 | ...
    // and on paper a keyframe costs max(fusion, compaction + tail) instead of their sum.  What fuse_f needs of keyframe f - 1 (deleted-slot counts
    // per sub-block, D, the live count before compaction f - 1) rotates through three slots, so that nothing it reads is written while it runs.
    // All resident-map parity tests pass in this mode (identical maps, counters and new-surfel lists).  It is OFF by default because it measured
    // SLOWER on MI355X / ROCm 7.2: SurfelFusion alone 19.2 k keyframes/s against 21.3 k (dense map), 23.2 k against 26.2 k (sparse map) -- the two
    // cross-stream event dependencies per keyframe cost more than the two in-stream dependent launches they replace (k_fuse itself is unchanged,
    // 26.8 us per event pair in both modes; GPU_MAX_HW_QUEUES = 8 / 16, the compaction stream's priority and device-scope release events made no
    // difference).  Kept, tested and documented as a measured dead end for the stream-level form of the idea.
    static const bool overlapOn = getenv("MSL_SF_OVERLAP") && !strcmp(getenv("MSL_SF_OVERLAP"), "1");
    const bool ov = overlapOn && compact && h->ownStreams && sp != sm && n > 1;
    if (ov && !h->cmpStream) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        const char *pe = getenv("MSL_SF_CMP_PRIO");   // experiment hook: 0 = default priority, 1 = highest (default), 2 = lowest
        const int pr = pe && pe[0] == '0' ? 0 : (pe && pe[0] == '2' ? lo : hi);
        MSL_HIP_TRY(hipStreamCreateWithPriority(&h->cmpStream, hipStreamNonBlocking, pr));
        MSL_HIP_TRY(hipEventCreateWithFlags(&h->evCmp, hipEventDisableTiming | hipEventReleaseToDevice));
    }
    if (ov)
        while ((int)h->evFuse.size() < n) {
            hipEvent_t a, b;
            // device-scope release: the events order kernels of two streams of this GPU, nothing the host reads
            MSL_HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming | hipEventReleaseToDevice)); MSL_HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming | hipEventReleaseToDevice));
            h->evFuse.push_back(a); h->evTail.push_back(b);
        }
    // Merged launches (MSL_SF_MERGED=1): from the third keyframe of a call on, ONE launch per keyframe -- workgroup 0 of k_fuse_merged compacts
    // keyframe f - 1 (compact_wave) while the other workgroups fuse keyframe f; the few waves whose sub-block that compaction can touch poll a
    // flag and then read past their L2.  No second stream, no events: the chain loses one dependent launch per keyframe and the compaction
    // disappears behind the fusion.  fuse_0, compact_0, fuse_1 | K_2 = {compact_1, fuse_2} | K_3 | ... | compact_{n-1}.
    static const bool mergedOn = getenv("MSL_SF_MERGED") && !strcmp(getenv("MSL_SF_MERGED"), "1");
    const bool mg = mergedOn && !ov && compact && n >= 3 && (D.nseeds & 15) == 0;
    hipStream_t sc = ov ? h->cmpStream : sm;
    for (int f = 0; f < n; f++) {
        // slot rotation: this keyframe's hand-over data in slot j % 3, what keyframe j - 1 left in (j - 1) % 3, live counts published in ctr[16 + slot]
        const unsigned long long j = h->kfSerial + (unsigned long long)f;
        const int par = (int)(j % 3), prev = (int)((j + 2) % 3), next = (int)((j + 1) % 3);
        P.blockSums = D.blockSums + (size_t)par * h->blkStride; P.blockUpd = D.blockUpd + (size_t)par * h->blkStride;
        P.delU = D.delU + (size_t)par * LIST_D; P.delUCount = h->d_tickets + 4 + par;
        P.prevBlockSums = D.blockSums + (size_t)prev * h->blkStride; P.prevDelUCount = h->d_tickets + 4 + prev;
        P.nPubPrev = h->d_ctr + 16 + next;             // n before compaction j - 1 = after compaction j - 2, slot (j - 2) % 3 = (j + 1) % 3
        P.resetDelUCount = h->d_tickets + 4 + prev;    // slot (j + 2) % 3 = (j - 1) % 3: keyframe j + 2's; its last readers (fusion j) are done when compaction j runs
        h->lastPar = par;
        P.updCtr = mg ? h->d_updCtr + 64 * par : nullptr;
        if (mg && f >= 2) {
            // K_f: compaction of keyframe f - 1 (its hand-over data in slot prev, its superpixel data in slot f - 1) + fusion of keyframe f
            P.fuseMode = 3;
            P.prevDelU = D.delU + (size_t)prev * LIST_D; P.prevUpdCtr = h->d_updCtr + 64 * prev; P.prevSlot = f - 1;
            P.nPubCompact = h->d_ctr + 16 + prev;               // compaction j - 1 publishes in its own slot
            P.resetDelUCountCompact = h->d_tickets + 4 + next;  // (j - 1 + 2) % 3: keyframe j + 1's counter
            P.doneFlag = h->d_tickets + 7; P.epoch = (unsigned)j;
            P.nPubOut = h->d_ctr + 16 + par;
            LAUNCH(SK_FUSE, sm, k_fuse_merged, dim3((unsigned)nSubGrid + 1u), dim3(64), P, f, h->h_frames[slot0 + f], nSubHint);
            if (f == n - 1) {   // the call's last keyframe: its compaction as a launch of its own
                P.fuseMode = 0;
                LAUNCH(SK_COMPACT, sm, k_compact, dim3(128), dim3(256), P, f, 0);
            }
            if (f == n / 2) {
                hipEvent_t ea, eb;
                if (h->prof.kernel_pair(SK_NEW, &ea, &eb)) hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sm, ea, eb, 0, 0);
            }
            continue;
        }
        if (ov && f > 0) {
            P.fuseMode = 2; P.nPubOut = h->d_ctr + 16 + par;
            hipLaunchKernelGGL(k_fuse, dim3((unsigned)nSubGrid), dim3(64), 0, sc, P, f, h->h_frames[slot0 + f], nSubHint);   // tail_f: behind compaction f - 1
            MSL_HIP_TRY(hipEventRecord(h->evTail[f], sc));
            if (f > 1) MSL_HIP_TRY(hipStreamWaitEvent(sm, h->evTail[f - 1], 0));   // fuse_f reads what tail_{f-1} wrote (deleted counts of keyframe f - 1)
            P.fuseMode = 1;
        } else {
            P.fuseMode = 0;
        }
        P.nPubOut = P.fuseMode == 0 ? h->d_ctr + 16 + prev : h->d_ctr + 16 + par;   // mode 0 publishes the count it found where keyframe j + 1's mode 1 looks ((j + 1 - 2) % 3)
        LAUNCH(SK_FUSE, sm, k_fuse, dim3((unsigned)nSubGrid), dim3(64), P, f, h->h_frames[slot0 + f], nSubHint);
        if (ov) {
            MSL_HIP_TRY(hipEventRecord(h->evFuse[f], sm));
            MSL_HIP_TRY(hipStreamWaitEvent(sc, h->evFuse[f], 0));
        }
        P.nPubOut = h->d_ctr + 16 + par;
        if (!(mg && f == 1))   // (merged: keyframe 1's compaction is workgroup 0 of K_2)
            LAUNCH(SK_COMPACT, sc, k_compact, dim3(compact ? 128 : 1), dim3(256), P, f, compact ? 0 : 1);   // scan + new surfels + refill + tail compaction
        if (ov && f == n - 1) { MSL_HIP_TRY(hipEventRecord(h->evCmp, sc)); MSL_HIP_TRY(hipStreamWaitEvent(sm, h->evCmp, 0)); }
        if (f == n / 2) {   // only when its profiler slot is enabled: what an event pair reports for an EMPTY dispatch at this place of the chain
            hipEvent_t ea, eb;   // (the pair's first event completes with the previous command, so every event time contains the dependent-launch gap)
            if (h->prof.kernel_pair(SK_NEW, &ea, &eb)) hipExtLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sm, ea, eb, 0, 0);
        }
    }
    h->kfSerial += (unsigned long long)n;
    if (sp != sm) { MSL_HIP_TRY(hipEventRecord(h->evMap[set], sm)); h->evMapValid[set] = true; }
    if (compact && h->h_snap) {   // snapshot of the live count after this batch (picked up by a later call, never waited for)
        const int i = h->snapNext;
        if (!h->snapBusy[i]) {
            MSL_HIP_TRY(hipMemcpyAsync(&h->h_snap[i], h->d_ctr, sizeof(long long), hipMemcpyDeviceToHost, sm));
            MSL_HIP_TRY(hipEventRecord(h->snapEv[i], sm));
            h->snapKf[i] = h->kfEnq; h->snapBusy[i] = true; h->snapNext = (i + 1) % msl_sf::NSNAP;
        }
    }
    MSL_HIP_TRY(hipGetLastError());
    h->lastSlot = slot0 + n - 1;
    h->batchNo++;
    return MSL_OK;
}

}  // namespace

extern "C" {

msl_sf *msl_sf_create(int width, int height, float fx, float fy, float cx, float cy, float fuseFar, float fuseNear, int device) {
    if (width < 16 || height < 16 || fx == 0 || fy == 0 || (width / SP) * (height / SP) >= IDX_PLANE || (long long)width * height >= (1ll << 31)) {
        set_error("msl_sf_create: width/height must be >= 16 with fewer than 65534 superpixels, fx and fy non-zero");
        return nullptr;
    }
    if (bind_device(device) != MSL_OK) return nullptr;
    msl_sf *h = new msl_sf;
    h->device = device;
    SfDev &D = h->dev;
    D.W = width; D.H = height; D.spW = width / SP; D.spH = height / SP; D.nseeds = D.spW * D.spH; D.npx = width * height;   // spWidth = width / SP_SIZE: truncation (:29-38)
    D.pxStride = (D.npx + 63) & ~63;
    D.fx = fx; D.fy = fy; D.cx = cx; D.cy = cy; D.fuseFar = fuseFar; D.fuseNear = fuseNear;
    bool ok = true;
    {   // the per-keyframe map stage is the latency-critical chain: highest priority for its stream, lowest for the
        // throughput-oriented frame-batched superpixel stage
        int lo = 0, hi = 0;
        ok = ok && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&h->preStream, hipStreamNonBlocking, lo) == hipSuccess;
        ok = ok && hipStreamCreateWithPriority(&h->mapStream, hipStreamNonBlocking, hi) == hipSuccess;
        ok = ok && hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking) == hipSuccess;
    }
    for (int i = 0; i < 2 && ok; i++)
        ok = hipEventCreateWithFlags(&h->evPre[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->evMap[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->evCopy[i], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&h->evH2D[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&h->d_ctr, sizeof(long long) * 32) == hipSuccess;   // 16 counters (read_ctr) + [16..18] the published live counts
    ok = ok && hipMemset(h->d_ctr, 0, sizeof(long long) * 32) == hipSuccess;
    ok = ok && hipHostMalloc(&h->h_ctr, sizeof(long long) * 16) == hipSuccess;
    ok = ok && hipHostMalloc(&h->h_snap, sizeof(long long) * msl_sf::NSNAP) == hipSuccess;
    for (int i = 0; i < msl_sf::NSNAP && ok; i++) ok = hipEventCreateWithFlags(&h->snapEv[i], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc(&h->d_new, sizeof(msl_surfel) * D.nseeds) == hipSuccess;
    ok = ok && hipMalloc(&h->d_tickets, sizeof(unsigned) * 8) == hipSuccess && hipMemset(h->d_tickets, 0, sizeof(unsigned) * 8) == hipSuccess;   // [0..1] tickets, [3] change-list length, [4..6] the rotating hand-over counts
    ok = ok && hipMalloc(&h->d_delU, sizeof(unsigned) * LIST_D * 3) == hipSuccess;
    ok = ok && hipMalloc(&h->d_updCtr, sizeof(unsigned) * 64 * 3) == hipSuccess && hipMemset(h->d_updCtr, 0, sizeof(unsigned) * 64 * 3) == hipSuccess;
    {   // (u - cx) / fx and (v - cy) / fy of every integer pixel coordinate: the float expression of back_project
        // (src/SurfelFusion.cpp:80-85) evaluated once here instead of six divisions per pixel in kb_seed_plane
        std::vector<float> tab((size_t)width + 1 + height + 1);
        for (int u = 0; u <= width; u++) tab[u] = ((float)u - cx) / fx;
        for (int v = 0; v <= height; v++) tab[(size_t)width + 1 + v] = ((float)v - cy) / fy;
        ok = ok && hipMalloc(&h->d_projTab, sizeof(float) * tab.size()) == hipSuccess;
        ok = ok && hipMemcpy(h->d_projTab, tab.data(), sizeof(float) * tab.size(), hipMemcpyHostToDevice) == hipSuccess;
        D.colX = h->d_projTab; D.rowY = h->d_projTab + width + 1;
    }
    if (ok && D.nseeds <= PROP_LDS_MAX_SEEDS)   // the attribute belongs to the function, not to this handle: always ask for the largest size any handle may use
        h->propLds = hipFuncSetAttribute((const void *)kb_prop_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(unsigned) * PROP_LDS_MAX_SEEDS)) == hipSuccess;
    if (!ok) { set_error("msl_sf_create: HIP allocation failed"); msl_sf_destroy(h); return nullptr; }
    memset(h->h_ctr, 0, sizeof(long long) * 16);
    D.ctr = h->d_ctr; D.newSurfels = h->d_new; D.tickets = h->d_tickets; D.delU = h->d_delU; D.delUCount = h->d_tickets + 4;
    D.prevDelU = h->d_delU; D.prevBlockUpd = nullptr; D.nPubCompact = h->d_ctr + 16; D.resetDelUCountCompact = h->d_tickets + 4; D.doneFlag = h->d_tickets + 7; D.epoch = 0; D.prevSlot = 0;
    D.updCtr = nullptr; D.prevUpdCtr = h->d_updCtr;
    D.fuseMode = 0; D.prevBlockSums = nullptr; D.prevDelUCount = h->d_tickets + 4; D.nPubPrev = h->d_ctr + 16; D.nPubOut = h->d_ctr + 16; D.resetDelUCount = h->d_tickets + 4;
    h->prof.nk = MSL_SF_NKERNELS;
    if (alloc_slots(h, 1) != MSL_OK || map_realloc(h, 1 << 16, 0) != MSL_OK) { msl_sf_destroy(h); return nullptr; }
    return h;
}

void msl_sf_destroy(msl_sf *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->copyStream) (void)hipStreamSynchronize(h->copyStream);
    if (h->preStream) (void)hipStreamSynchronize(h->preStream);
    if (h->mapStream) (void)hipStreamSynchronize(h->mapStream);
    h->prof.destroy();
    free_slots(h);
    auto F = [](auto *p) { if (p) (void)hipFree(p); };
    F(h->d_ctr); F(h->d_tickets); F(h->d_delU); F(h->d_updCtr); F(h->d_projTab); F(h->d_new); F(h->d_mapStore); F(h->d_blockSums); F(h->d_blockUpd); F(h->d_delList); F(h->d_srcOf); F(h->d_aos); F(h->d_snapStore);
    if (h->h_ctr) (void)hipHostFree(h->h_ctr);
    if (h->h_snap) (void)hipHostFree(h->h_snap);
    if (h->h_blk) (void)hipHostFree(h->h_blk);
    if (h->h_list) (void)hipHostFree(h->h_list);
    for (int i = 0; i < msl_sf::NSNAP; i++) if (h->snapEv[i]) (void)hipEventDestroy(h->snapEv[i]);
    for (int i = 0; i < 2; i++) { if (h->evPre[i]) (void)hipEventDestroy(h->evPre[i]); if (h->evMap[i]) (void)hipEventDestroy(h->evMap[i]); if (h->evCopy[i]) (void)hipEventDestroy(h->evCopy[i]); if (h->evH2D[i]) (void)hipEventDestroy(h->evH2D[i]); }
    if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
    if (h->ownStreams) { if (h->preStream) (void)hipStreamDestroy(h->preStream); if (h->mapStream) (void)hipStreamDestroy(h->mapStream); }
    if (h->cmpStream) { (void)hipStreamSynchronize(h->cmpStream); (void)hipStreamDestroy(h->cmpStream); }
    for (hipEvent_t e : h->evFuse) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->evTail) (void)hipEventDestroy(e);
    if (h->evCmp) (void)hipEventDestroy(h->evCmp);
    delete h;
}

int msl_sf_set_stream(msl_sf *h, void *hip_stream) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (h->ownStreams) { (void)hipStreamDestroy(h->preStream); (void)hipStreamDestroy(h->mapStream); }
    h->preStream = h->mapStream = (hipStream_t)hip_stream; h->ownStreams = false;
    return MSL_OK;
}

int msl_sf_set_batch_capacity(msl_sf *h, int max_frames) {
    if (!h || max_frames < 1 || max_frames > 4096) { set_error("msl_sf_set_batch_capacity: invalid argument"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (max_frames == h->maxBatch) return MSL_OK;
    return alloc_slots(h, max_frames);
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
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    if (n + (size_t)h->dev.nseeds > h->mapCap) {
        rc = map_realloc(h, n + n / 4 + 4 * (size_t)h->dev.nseeds, 0);
        if (rc != MSL_OK) return rc;
    }
    hipStream_t s = h->mapStream;
    MSL_HIP_TRY(hipMemsetAsync(h->d_ctr + 13, 0, sizeof(long long), s));   // a fresh map: no wide r, g, b records yet
    if (n) {
        rc = ensure_aos(h, n);
        if (rc != MSL_OK) return rc;
        MSL_HIP_TRY(hipMemcpyAsync(h->d_aos, host, sizeof(msl_surfel) * n, hipMemcpyHostToDevice, s));
        LAUNCH(SK_CONVERT, s, k_aos_to_soa, dim3((unsigned)((n + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n);
    }
    hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(64), 0, s, h->d_ctr, (long long)n, h->d_tickets + 4, -1);
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->liveBound = n; h->liveKnown = n; h->liveKnownKf = h->kfEnq;
    drop_live_snapshots(h);   // a count recorded before the upload would otherwise lower the bound below n
    return MSL_OK;
}

int msl_sf_map_snapshot(msl_sf *h) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    if (n > h->snapCap) {
        if (h->d_snapStore) (void)hipFree(h->d_snapStore);
        h->d_snapStore = nullptr; h->snapCap = 0; h->snapValid = false;
        const size_t c = (n + 4095) & ~(size_t)4095;
        MSL_HIP_TRY(hipMalloc(&h->d_snapStore, sizeof(float) * 16 * c));
        h->snapCap = c;
    }
    if (n) {
        MSL_HIP_TRY(hipMemcpy(h->d_snapStore, h->dev.map.hot, sizeof(HotRec) * n, hipMemcpyDeviceToDevice));
        MSL_HIP_TRY(hipMemcpy(h->d_snapStore + 5 * h->snapCap, h->dev.map.cold, sizeof(ColdRec) * n, hipMemcpyDeviceToDevice));
        if (h->h_ctr[13]) MSL_HIP_TRY(hipMemcpy(h->d_snapStore + 13 * h->snapCap, h->dev.map.rgbWide, sizeof(int) * 3 * n, hipMemcpyDeviceToDevice));
    }
    h->snapN = n; h->snapValid = true; h->snapWide = h->h_ctr[13] != 0;
    return MSL_OK;
}

int msl_sf_map_restore(msl_sf *h) {
    if (!h || !h->snapValid) { set_error("msl_sf_map_restore: no snapshot"); return MSL_ERR_INVALID; }
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    const size_t n = h->snapN;
    if (n + (size_t)h->dev.nseeds > h->mapCap) {   // the map was reallocated smaller than the snapshot (upload of a small map): grow again
        int rc = read_ctr(h);
        if (rc != MSL_OK) return rc;
        rc = map_realloc(h, n + n / 4 + 4 * (size_t)h->dev.nseeds, 0);
        if (rc != MSL_OK) return rc;
    }
    hipStream_t s = h->mapStream;   // ordered after every keyframe enqueued so far; the superpixel stream never touches the map
    if (n) {
        MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.hot, h->d_snapStore, sizeof(HotRec) * n, hipMemcpyDeviceToDevice, s));
        MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.cold, h->d_snapStore + 5 * h->snapCap, sizeof(ColdRec) * n, hipMemcpyDeviceToDevice, s));
        if (h->snapWide) {
            MSL_HIP_TRY(hipMemcpyAsync(h->dev.map.rgbWide, h->d_snapStore + 13 * h->snapCap, sizeof(int) * 3 * n, hipMemcpyDeviceToDevice, s));
        }
    }
    // the restored map has exactly the snapshot's wide-rgb state: without the flag a later snapshot would skip rgbWide and a restore of THAT
    // one would bring COLD_WIDE records back without their exact ints
    hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(64), 0, s, h->d_ctr, (long long)n, h->d_tickets + 4, h->snapWide ? 1 : 0);
    MSL_HIP_TRY(hipGetLastError());
    h->liveBound = n; h->liveKnown = n; h->liveKnownKf = h->kfEnq;
    drop_live_snapshots(h);
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
        hipStream_t s = h->mapStream;
        LAUNCH(SK_CONVERT, s, k_soa_to_aos, dim3((unsigned)((n + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n);
        MSL_HIP_TRY(hipMemcpyAsync(host, h->d_aos, sizeof(msl_surfel) * n, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipStreamSynchronize(s));
    }
    return check_err(h);
}

static int map_select(msl_sf *h, int mode, int arg, bool mark, msl_surfel *out, size_t cap, size_t *n_out, const char *what) {
    if (!h || !n_out) { set_error("%s: invalid argument", what); return MSL_ERR_INVALID; }
    if (mark) h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);                       // waits for both streams
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t n = (size_t)h->h_ctr[0];
    *n_out = 0;
    if (n == 0) return MSL_OK;
    hipStream_t s = h->mapStream;
    const SfDev P = h->dev;
    hipLaunchKernelGGL(k_select_count, dim3(512), dim3(256), 0, s, P, mode, arg);
    hipLaunchKernelGGL(k_select_scan, dim3(1), dim3(1024), 0, s, P);
    rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    const size_t m = (size_t)h->h_ctr[7];
    *n_out = m;
    (void)hipMemsetAsync(h->d_ctr + 7, 0, sizeof(long long), s);
    if (m > cap || (m && !out)) { set_error("%s: %zu surfels selected, capacity %zu", what, m, cap); return MSL_ERR_CAPACITY; }
    if (m == 0) return MSL_OK;
    rc = ensure_aos(h, m);
    if (rc != MSL_OK) return rc;
    hipLaunchKernelGGL(k_select_write, dim3(512), dim3(256), 0, s, P, mode, arg, h->d_aos, mark ? 1 : 0);
    MSL_HIP_TRY(hipMemcpyAsync(out, h->d_aos, sizeof(msl_surfel) * m, hipMemcpyDeviceToHost, s));
    MSL_HIP_TRY(hipStreamSynchronize(s));
    return MSL_OK;
}

int msl_sf_map_detach(msl_sf *h, int pose_index, msl_surfel *out, size_t cap, size_t *n_out) {
    return map_select(h, 0, pose_index, true, out, cap, n_out, "msl_sf_map_detach");
}
int msl_sf_map_export(msl_sf *h, int min_update_times, msl_surfel *out, size_t cap, size_t *n_out) {
    return map_select(h, 1, min_update_times, false, out, cap, n_out, "msl_sf_map_export");
}
// System::saveSurfels (src/System.cc:296-382) for the cloud SurfelMapping::Stop builds (src/SurfelMapping.cpp:62-104): the local surfels
// seen at least min_update_times times (filtered on the device, map order), then the caller's inactive surfels.  ASCII PLY with the
// element / property layout the reference hands to tinyply; NaN positions are skipped (:311-312); alpha = 1, quality = weight,
// radius = size * 1000 (SurfelMapping.cpp:80).  Number formatting is that of a default std::ostream (tinyply itself is a third party).
int msl_sf_export_ply(msl_sf *h, int min_update_times, const msl_surfel *inactive, size_t n_inactive, const char *path) {
    if (!h || !path || (n_inactive && !inactive)) { set_error("msl_sf_export_ply: invalid argument"); return MSL_ERR_INVALID; }
    size_t n = 0;
    int rc = msl_sf_map_export(h, min_update_times, nullptr, 0, &n);
    if (rc != MSL_OK && rc != MSL_ERR_CAPACITY) return rc;
    std::vector<msl_surfel> pts(n + n_inactive);
    if (n) { rc = msl_sf_map_export(h, min_update_times, pts.data(), n, &n); if (rc != MSL_OK) return rc; }
    for (size_t i = 0; i < n_inactive; i++) pts[n + i] = inactive[i];
    size_t count = 0;
    for (const msl_surfel &e : pts) count += std::isnan(e.px) ? 0 : 1;
    std::ofstream os(path, std::ios::out);
    if (os.fail()) { set_error("msl_sf_export_ply: cannot open %s", path); return MSL_ERR_INVALID; }
    os << "ply\nformat ascii 1.0\nelement vertex " << count << "\n";
    for (const char *p : {"x", "y", "z", "nx", "ny", "nz"}) os << "property float " << p << "\n";
    for (const char *p : {"red", "green", "blue", "alpha"}) os << "property uchar " << p << "\n";
    for (const char *p : {"quality", "radius"}) os << "property float " << p << "\n";
    os << "element camera 1\n";
    for (const char *p : {"view_px", "view_py", "view_pz", "x_axisx", "x_axisy", "x_axisz", "y_axisx", "y_axisy", "y_axisz", "z_axisx", "z_axisy", "z_axisz",
                          "focal", "scalex", "scaley", "centerx", "centery"})
        os << "property float " << p << "\n";
    os << "property int viewportx\nproperty int viewporty\nproperty float k1\nproperty float k2\nend_header\n";
    for (const msl_surfel &e : pts) {
        if (std::isnan(e.px)) continue;
        os << e.px << " " << e.py << " " << e.pz << " " << e.nx << " " << e.ny << " " << e.nz << " " << (unsigned)(uint8_t)e.r << " " << (unsigned)(uint8_t)e.g << " "
           << (unsigned)(uint8_t)e.b << " 1 " << e.weight << " " << e.size * 1000 << "\n";
    }
    os << "0 0 0 1 0 0 0 1 0 0 0 1 0 0 0 0 0 " << (int)count << " 1 0 0\n";
    return os.fail() ? MSL_ERR_INVALID : MSL_OK;
}

int msl_sf_map_append(msl_sf *h, const msl_surfel *surfels, size_t n) {
    if (!h || (n && !surfels)) { set_error("msl_sf_map_append: invalid argument"); return MSL_ERR_INVALID; }
    h->mirrorValid = false;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    if (n == 0) return MSL_OK;
    const size_t cur = (size_t)h->h_ctr[0];
    if (cur + n + (size_t)h->dev.nseeds > h->mapCap) {
        rc = map_realloc(h, cur + n + (cur + n) / 4 + 4 * (size_t)h->dev.nseeds, cur);
        if (rc != MSL_OK) return rc;
    }
    rc = ensure_aos(h, n);
    if (rc != MSL_OK) return rc;
    hipStream_t s = h->mapStream;
    MSL_HIP_TRY(hipMemcpyAsync(h->d_aos, surfels, sizeof(msl_surfel) * n, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_aos_to_soa_at, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, h->dev.map, h->d_aos, (long long)n, h->d_ctr);
    hipLaunchKernelGGL(k_add_ctr, dim3(1), dim3(64), 0, s, h->d_ctr, (long long)n);
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->liveBound = cur + n; h->liveKnown = cur + n; h->liveKnownKf = h->kfEnq;
    return MSL_OK;
}

int msl_sf_fuse_resident_batch(msl_sf *h, int n_frames, const int32_t *refs, const uint8_t *gray, size_t gray_stride,
                               size_t gray_frame_stride, const float *depth, size_t depth_stride, size_t depth_frame_stride,
                               const int32_t *member, size_t member_stride, size_t member_frame_stride, msl_mem img_mem,
                               const float *poses_colmajor) {
    if (!h) { set_error("msl_sf_fuse_resident_batch: NULL handle"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    return run_batch(h, n_frames, refs, gray, gray_stride, gray_frame_stride, depth, depth_stride, depth_frame_stride, member, member_stride,
                     member_frame_stride, img_mem, poses_colmajor, true);
}

int msl_sf_fuse_resident(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth,
                         size_t depth_stride, const int32_t *member, size_t member_stride, msl_mem img_mem,
                         const float pose_colmajor[16]) {
    if (!h) { set_error("msl_sf_fuse_resident: NULL handle"); return MSL_ERR_INVALID; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    const int32_t ref = referenceFrameIndex;
    return run_batch(h, 1, &ref, gray, gray_stride, 0, depth, depth_stride, 0, member, member_stride, 0, img_mem, pose_colmajor, true);
}

int msl_sf_last_counters(msl_sf *h, int64_t counters[5]) {
    if (!h || !counters) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    counters[0] = h->h_ctr[4]; counters[1] = h->h_ctr[1]; counters[2] = h->h_ctr[2]; counters[3] = h->h_ctr[3]; counters[4] = h->h_ctr[0];
    return check_err(h);
}

// Host-vector mode.  The caller's vector is the map for this call; what travels is kept to what has to:
//   in : the whole vector (56 B per surfel) -- unless MSL_SF_LOCAL_UNCHANGED says it still is what the previous call on this handle left there, in
//        which case the device copy of that call is used as it stands (checked: same length, no other map operation on the handle in between);
//   out: only the stretches of the vector that hold surfels this keyframe touched.  k_fuse leaves a deleted and an updated count per 256-surfel
//        sub-block; sub-blocks with neither are byte-identical to the caller's copy and are not sent back (runs of touched sub-blocks travel as
//        one copy each, small gaps bridged; more than 64 runs collapse into fewer by bridging larger gaps).
int msl_sf_fuse_ex(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth, size_t depth_stride,
                   const int32_t *member, size_t member_stride, const float pose_colmajor[16], msl_surfel *local, size_t n_local,
                   msl_surfel *new_out, size_t new_cap, size_t *n_new, unsigned flags) {
    if (!h || !pose_colmajor || !n_new || (n_local && !local)) { set_error("msl_sf_fuse: invalid argument"); return MSL_ERR_INVALID; }
    if (new_cap < (size_t)h->dev.nseeds || !new_out) { set_error("msl_sf_fuse: new_cap must be >= (w/8)*(h/8) = %d", h->dev.nseeds); return MSL_ERR_CAPACITY; }
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc;
    const bool reuse = (flags & MSL_SF_LOCAL_UNCHANGED) && h->mirrorValid && h->mirrorN == n_local;
    if (reuse) {
        // the device map is the caller's vector already: only the per-call counters start over
        hipLaunchKernelGGL(k_set_ctr, dim3(1), dim3(64), 0, h->mapStream, h->d_ctr, (long long)n_local, h->d_tickets + 4, -1);
        h->liveBound = n_local; h->liveKnown = n_local; h->liveKnownKf = h->kfEnq;
    } else {
        rc = msl_sf_map_upload(h, local, n_local);
        if (rc != MSL_OK) return rc;
    }
    h->mirrorValid = false;   // (until this call has completed)
    const int32_t ref = referenceFrameIndex;
    rc = run_batch(h, 1, &ref, gray, gray_stride, 0, depth, depth_stride, 0, member, member_stride, 0, MSL_MEM_HOST, pose_colmajor, false);
    if (rc != MSL_OK) return rc;
    hipStream_t s = h->mapStream;
    const size_t nblk = (n_local + SUB_ITEMS - 1) / SUB_ITEMS;
    if (nblk > h->blkCap) {
        if (h->h_blk) (void)hipHostFree(h->h_blk);
    if (h->h_list) (void)hipHostFree(h->h_list);
        h->h_blk = nullptr; h->blkCap = 0;
        MSL_HIP_TRY(hipHostMalloc(&h->h_blk, sizeof(unsigned) * 2 * (nblk + 1024)));
        h->blkCap = nblk + 1024;
    }
    if (nblk) {
        // (the keyframe's per-sub-block counts sit in the rotating slot run_batch used: lastPar)
        MSL_HIP_TRY(hipMemcpyAsync(h->h_blk, h->dev.blockSums + (size_t)h->lastPar * h->blkStride, sizeof(unsigned) * nblk, hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipMemcpyAsync(h->h_blk + h->blkCap, h->dev.blockUpd + (size_t)h->lastPar * h->blkStride, sizeof(unsigned) * nblk, hipMemcpyDeviceToHost, s));
    }
    rc = read_ctr(h);   // the call's first synchronisation: counters and the per-sub-block counts are on the host
    if (rc != MSL_OK) return rc;
    rc = check_err(h);
    if (rc != MSL_OK) return rc;
    const size_t K = (size_t)h->h_ctr[1];
    *n_new = K;
    // what was touched, and where
    struct Run { size_t b0, b1; };
    std::vector<Run> runs;
    size_t touched = 0, runSurfels = 0;
    for (size_t b = 0; b < nblk; b++) touched += (size_t)h->h_blk[b] + h->h_blk[h->blkCap + b];
    for (size_t gapMax = 4; ; gapMax *= 4) {
        runs.clear();
        for (size_t b = 0; b < nblk; b++) {
            if (!(h->h_blk[b] | h->h_blk[h->blkCap + b])) continue;
            if (!runs.empty() && b - runs.back().b1 <= gapMax) runs.back().b1 = b + 1;
            else runs.push_back({b, b + 1});
        }
        if (runs.size() <= 64) break;
    }
    for (const Run &r : runs) runSurfels += std::min(r.b1 * SUB_ITEMS, n_local) - r.b0 * SUB_ITEMS;
    // Two ways back.  Runs of touched sub-blocks copied straight into the caller's vector (~45 GB/s), or -- when few surfels in many sub-blocks
    // changed (a map in no particular order) -- a compact {index, record} list scattered by the CPU (~6 ns per record on top of its 60 bytes).
    const size_t listLimit = n_local / 8;
    const double costRuns = 56.0 * (double)runSurfels / 45e9, costList = (double)touched * (60.0 / 45e9 + 6e-9);
    if (touched && touched <= listLimit && costList < costRuns) {
        const size_t need = 256 + (sizeof(unsigned) + sizeof(msl_surfel)) * listLimit;
        if (need > h->listCap) {
            if (h->h_list) (void)hipHostFree(h->h_list);
            h->h_list = nullptr; h->listCap = 0;
            MSL_HIP_TRY(hipHostMalloc(&h->h_list, need));
            h->listCap = need;
        }
        // device side: the count sits in tickets[3], indices in delList, records in the AoS buffer (both >= n_local entries)
        unsigned *d_count = h->d_tickets + 3;
        MSL_HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned), s));
        SfDev Pc = h->dev;
        Pc.blockSums = h->dev.blockSums + (size_t)h->lastPar * h->blkStride; Pc.blockUpd = h->dev.blockUpd + (size_t)h->lastPar * h->blkStride;
        hipLaunchKernelGGL(k_collect_changed, dim3((unsigned)nblk), dim3(64), 0, s, Pc, (int)ref, (long long)n_local, d_count, h->dev.delList, h->d_aos, (unsigned)listLimit);
        unsigned *hc = reinterpret_cast<unsigned *>(h->h_list);
        unsigned *hi = reinterpret_cast<unsigned *>(h->h_list + 256);
        msl_surfel *hr = reinterpret_cast<msl_surfel *>(h->h_list + 256 + sizeof(unsigned) * listLimit);
        MSL_HIP_TRY(hipMemcpyAsync(hc, d_count, sizeof(unsigned), hipMemcpyDeviceToHost, s));
        MSL_HIP_TRY(hipStreamSynchronize(s));
        // (the list may be longer than `touched`: a surfel that carried lastUpdate == ref before the call is listed as well -- harmless, its
        // record is unchanged -- so the length is read first; a list beyond the staging size falls back to the runs)
        const size_t cnt = *hc;
        if (cnt <= listLimit) {
            MSL_HIP_TRY(hipMemcpyAsync(hi, h->dev.delList, sizeof(unsigned) * cnt, hipMemcpyDeviceToHost, s));
            MSL_HIP_TRY(hipMemcpyAsync(hr, h->d_aos, sizeof(msl_surfel) * cnt, hipMemcpyDeviceToHost, s));
            if (K) MSL_HIP_TRY(hipMemcpyAsync(new_out, h->d_new, sizeof(msl_surfel) * K, hipMemcpyDeviceToHost, s));
            MSL_HIP_TRY(hipStreamSynchronize(s));
            for (size_t j = 0; j < cnt; j++) local[hi[j]] = hr[j];
            h->mirrorValid = true; h->mirrorN = n_local;
            return MSL_OK;
        }
    }
    if (nblk && !runs.empty())
        LAUNCH(SK_CONVERT, s, k_soa_to_aos, dim3((unsigned)((n_local + 255) / 256)), dim3(256), h->dev.map, h->d_aos, (long long)n_local);
    for (const Run &r : runs) {
        const size_t i0 = r.b0 * SUB_ITEMS, i1 = std::min(r.b1 * SUB_ITEMS, n_local);
        MSL_HIP_TRY(hipMemcpyAsync(local + i0, h->d_aos + i0, sizeof(msl_surfel) * (i1 - i0), hipMemcpyDeviceToHost, s));
    }
    if (K) MSL_HIP_TRY(hipMemcpyAsync(new_out, h->d_new, sizeof(msl_surfel) * K, hipMemcpyDeviceToHost, s));
    MSL_HIP_TRY(hipStreamSynchronize(s));
    h->mirrorValid = true; h->mirrorN = n_local;
    return MSL_OK;
}

int msl_sf_fuse(msl_sf *h, int referenceFrameIndex, const uint8_t *gray, size_t gray_stride, const float *depth, size_t depth_stride,
                const int32_t *member, size_t member_stride, const float pose_colmajor[16], msl_surfel *local, size_t n_local,
                msl_surfel *new_out, size_t new_cap, size_t *n_new) {
    return msl_sf_fuse_ex(h, referenceFrameIndex, gray, gray_stride, depth, depth_stride, member, member_stride, pose_colmajor, local, n_local, new_out,
                          new_cap, n_new, 0u);
}

int msl_sf_debug_seeds(msl_sf *h, msl_seed *out) {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    const size_t ns = h->dev.nseeds;
    MSL_HIP_TRY(hipMemcpy(out, h->d_seeds + ns * h->lastSlot, sizeof(msl_seed) * ns, hipMemcpyDeviceToHost));
    std::vector<uint8_t> fused(ns);
    MSL_HIP_TRY(hipMemcpy(fused.data(), h->d_fused + ns * h->lastSlot, ns, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < ns; i++) out[i].fused = fused[i];
    return MSL_OK;
}
int msl_sf_debug_ctr(msl_sf *h, int64_t out[16]) {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctr(h);
    if (rc != MSL_OK) return rc;
    for (int i = 0; i < 16; i++) out[i] = h->h_ctr[i];
    return MSL_OK;
}
int msl_sf_debug_scratch(msl_sf *h, int which, size_t offset_words, uint32_t *out, size_t n_words) {
    if (!h || !out || which < 0 || which > 1 || offset_words + n_words > h->mapCap) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    MSL_HIP_TRY(hipMemcpy(out, (which == 0 ? h->d_srcOf : h->d_delList) + offset_words, sizeof(uint32_t) * n_words, hipMemcpyDeviceToHost));
    return MSL_OK;
}
// What an event pair carried by a dispatch (hipExtLaunchKernelGGL) reports for a kernel that does nothing: n launches of an empty kernel with
// `grid` single-wave workgroups on the map stream.  bench.py quotes it next to the roofline kernel's event time: rocprofv3's kernel duration
// (first wave start to last wave end) is shorter than the event time by about this much.
int msl_sf_debug_event_overhead(msl_sf *h, int grid, int n, float *mean_us) {
    if (!h || !mean_us || n < 1 || grid < 1) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    std::vector<hipEvent_t> ev(2 * (size_t)n);
    for (auto &e : ev) MSL_HIP_TRY(hipEventCreate(&e));
    for (int i = 0; i < n; i++) hipExtLaunchKernelGGL(k_empty, dim3((unsigned)grid), dim3(64), 0, h->mapStream, ev[2 * i], ev[2 * i + 1], 0, grid);
    MSL_HIP_TRY(hipStreamSynchronize(h->mapStream));
    double tot = 0;
    for (int i = 0; i < n; i++) { float ms = 0; MSL_HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
    for (auto &e : ev) (void)hipEventDestroy(e);
    *mean_us = (float)(tot * 1e3 / n);
    return MSL_OK;
}
int msl_sf_debug_index(msl_sf *h, int32_t *out) {
    if (!h || !out) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    const size_t npx = h->dev.npx;
    std::vector<unsigned short> tmp(npx);
    MSL_HIP_TRY(hipMemcpy(tmp.data(), h->d_index + (size_t)h->dev.pxStride * h->lastSlot, sizeof(unsigned short) * npx, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < npx; i++) out[i] = tmp[i];
    return MSL_OK;
}

int msl_sf_profile_enable(msl_sf *h, int on) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    h->prof.drain();
    h->prof.set_mode(on);
    return MSL_OK;
}
int msl_sf_profile_stride(msl_sf *h, int stride) {
    if (!h || stride < 1) return MSL_ERR_INVALID;
    h->prof.stride = stride;
    return MSL_OK;
}
int msl_sf_profile_read(msl_sf *h, float *ms, int32_t *launches) {
    if (!h) return MSL_ERR_INVALID;
    MSL_HIP_TRY(hipSetDevice(h->device));
    int rc = sync_all(h);
    if (rc != MSL_OK) return rc;
    h->prof.drain();
    for (int i = 0; i < MSL_SF_NKERNELS; i++) { if (ms) ms[i] = h->prof.ms[i]; if (launches) launches[i] = h->prof.launches[i]; }
    return MSL_OK;
}
int msl_debug_div100(const float *x_host, double *out_host, size_t n) {
    if (n == 0) return MSL_OK;
    if (!x_host || !out_host) return MSL_ERR_INVALID;
    float *dx = nullptr; double *dout = nullptr;
    MSL_HIP_TRY(hipMalloc(&dx, sizeof(float) * n)); MSL_HIP_TRY(hipMalloc(&dout, sizeof(double) * n));
    MSL_HIP_TRY(hipMemcpy(dx, x_host, sizeof(float) * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_debug_div100, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, dx, dout, (long long)n);
    MSL_HIP_TRY(hipMemcpy(out_host, dout, sizeof(double) * n, hipMemcpyDeviceToHost));
    (void)hipFree(dx); (void)hipFree(dout);
    return MSL_OK;
}
const char *msl_sf_kernel_name(int k) { return (k >= 0 && k < MSL_SF_NKERNELS) ? kSfNames[k] : ""; }

}  // extern "C"
